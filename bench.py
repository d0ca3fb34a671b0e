#!/usr/bin/env python3
"""Benchmark of the north-star path: images/sec of the migan-512 generator forward, batch 32 per GPU, fp32, on MI355X --
BASELINE.json's metric/config.

    python bench.py --gpus N --steps K --warmup W          # N > 1: spawns its N ranks itself (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          # or is launched as N ranks (RANK / WORLD_SIZE in the env)

A "step" is one Generator.forward over one synthetic batch already resident in HBM (input tensor on device -> output
tensor on device).  For N > 1 every rank runs its own batch of 32 (weak scaling: BASELINE configs[3] = 256 images over 8
GPUs) and each step ends with the RCCL all-gather of the output shards, inside the timed region.  Rank 0 prints ONE JSON
line; it exits non-zero if the number of ranks that took part differs from --gpus.

Other workloads behind the same protocol / JSON schema:
    --resolution 256 --dtype bf16      BASELINE configs[1] (16-bit activation storage)
    --model comodgan-512 --batch 16    BASELINE configs[4]
The default N = 1 run appends them as "secondary" (and the exact-fp32-MFMA variant of the primary workload as
"value_exact_f32"), so that one driver run times everything; --no-secondary skips that.

Extra objects on the line:
  roofline     dominant kernel (largest share of GPU time), measured live with hipEvent pairs around every launch on the
               launch stream (one stream, whole-batch launches): achieved = algorithmic flops (or bytes) of its launches /
               their summed duration, against the MI355X peak of the binding resource.
  cpu_baseline the torch-CPU port of the reference module (oracle/*_torch_cpu.py / comodgan_oracle.py; the reference itself
               is Python and is not present on the GPU box) timed on the host cores on a small sample of the same inputs --
               reported, not the target.
"""
import argparse
import importlib
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
T_PROCESS_START = time.perf_counter()   # (after the imports above: a fresh box spends 1-2 minutes paging torch in before this)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 peak
PEAK_BF16_MFMA_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak
# Split GEMM variants: every fp32 product costs six bf16 MFMA products ("bf16x3") or three fp16 MFMA
# products ("f16x2"), so the matrix-core ceiling in ALGORITHMIC flops is 2500 / 6 resp. 2500 / 3
MFMA_PRODUCTS = {"f32": None, "bf16x3": 6.0, "f16x2": 3.0, "f16": 1.0}
GEMM_TEXT = {"f32": "1x1 convs on exact fp32 MFMA",
             "bf16x3": "1x1 convs on bf16x3-split MFMA (6 bf16 products per fp32 product, fp32 accumulate)",
             "f16x2": "1x1 convs on f16x2-split MFMA (3 fp16 products per fp32 product on scaled operands, fp32 accumulate)",
             "f16": "1x1 convs on fp16 MFMA (operands rounded to fp16 after power-of-two scaling, one product each, fp32 accumulate)"}
BASELINE_CONFIG = {("migan", 512, "f32"): "BASELINE configs[2]", ("migan", 256, "bf16"): "BASELINE configs[1]",
                   ("comodgan", 512, "f32"): "BASELINE configs[4]"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", type=str, default="migan-512", help="migan-<R> | comodgan-<R>")
    ap.add_argument("--resolution", type=int, default=0, help="overrides the resolution in --model")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default 32, comodgan 16)")
    ap.add_argument("--dtype", type=str, default="f32", choices=["f32", "bf16", "f16"],
                    help="activation storage between layers (f32 = the reference's precision; bf16 = BASELINE configs[1])")
    ap.add_argument("--gemm", type=str, default="f16x2", choices=["f16x2", "bf16x3", "f32"])
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2, 3, 4], help="sub-batches / HIP streams per forward (migan)")
    ap.add_argument("--nan-policy", type=str, default="propagate", choices=["propagate", "clamp"],
                    help="migan: what lrelu_agc's clamp does with a NaN activation.  propagate (default library): it stays a NaN, as Tensor.clamp in the "
                         "reference module; clamp (libmigan_hip_nanclamp.so, -DMIGAN_NAN_CLAMP): -256, as the reference's CUDA plugin -- no NaN test "
                         "in the kernels, ~2 %% faster; finite inputs give the same bits")
    ap.add_argument("--io", type=str, default="f32", choices=["f32", "u8"],
                    help="migan: u8 = uint8 image + mask in, composed uint8 image out (demo.py's pre/post-processing inside the first / "
                         "last kernels, migan_forward_u8); the all-gather then moves uint8 shards (4x fewer bytes)")
    ap.add_argument("--cpu-images", type=int, default=4, help="sample size of the CPU baseline (0 = skip)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the output all-gather")
    ap.add_argument("--gather-dtype", type=str, default="auto", choices=["auto", "f32", "f16", "u8"],
                    help="N>1: what the ranks exchange.  auto = the forward's own output (fp32 [n,3,R,R], 100.7 MB per rank and step at "
                         "migan-512 x 32; uint8 [n,R,R,3], 25.2 MB, with --io u8); f16 = the fp32 output rounded to fp16 before the gather "
                         "(half the bytes; |y| ~ 30, so ~1e-2 absolute: not for parity runs); u8 requires --io u8")
    ap.add_argument("--gather-mode", type=str, default="inplace", choices=["inplace", "parts", "copy"],
                    help="N>1 (migan, fp32 output): inplace (default) = the forward writes its images into this rank's slice of the collective's receive "
                         "buffer, ONE in-place all_gather_into_tensor per step; parts = the same per sub-batch (migan_forward_parts: the collective of "
                         "sub-batch 0's shard is enqueued behind sub-batch 0 while sub-batch 1 computes); copy = the step's output tensor is copied "
                         "into the receive buffer first (producers that cannot write in place always do this).  Measured at world size 1 "
                         "(profiles/r06_experiments.md): exposed per step 0.0 / 0.29 / 0.05 ms")
    ap.add_argument("--reserve-cus", type=int, default=0, metavar="K",
                    help="N>1: run the forward on a HIP stream whose CU mask leaves K CUs (a multiple of 8: K/8 per XCD) to the RCCL "
                         "kernels of the overlapped all-gather, instead of letting them queue behind 256-CU-wide layers "
                         "(hipExtStreamCreateWithCUMask; implies --streams 1, the library's own side streams are not masked).  An "
                         "experiment knob, measured counterproductive (profiles/r05_contention.md): use multiples of 32 -- one CU per "
                         "shader engine of every XCD -- or the workgroup dealing becomes unbalanced")
    ap.add_argument("--occupy", type=str, default="", metavar="WGS,US[,THREADS[,LDS]]",
                    help="contention probe for the N>1 question on a one-GPU box: after every step a stand-in for the collective's kernel "
                         "(scripts/ubench/occupy.hip: WGS workgroups that hold their CUs for US microseconds) is launched on its own stream, "
                         "ordered behind the step like the output gather, so that it overlaps the next step")
    ap.add_argument("--force-pg", action="store_true",
                    help="N=1: also create a one-rank NCCL (= RCCL) process group and time the same steps through the pipelined output "
                         "gather (rccl_world1 on the line): the collective path on the hardware without a second GPU")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency measurement (latency_b1_ms)")
    ap.add_argument("--no-secondary", action="store_true", help="N=1 default run: skip the secondary workloads")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the CPU port on every logical core (on the 256-thread host of the GPU box: 0.015 images/s, ~70 s per "
                         "image -- oversubscription; off by default so that the default run stays within minutes)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads of the CPU baseline (0 = all logical cores); 16 is the fastest setting measured for this\n"
                         "graph of small oneDNN convs on the 256-thread GPU-box host (8: 1.04, 16: 0.94, 32: 1.07, 64: 1.72, 128: 4.4 s/img)")
    ap.add_argument("--dump-layers", type=str, default="", help="write per-launch hipEvent durations to this JSON file")
    ap.add_argument("--pmc-pass", type=int, default=0, metavar="N",
                    help="counter-collection pass (run under rocprofv3 --pmc): ONLY N + 2 per-launch-timed forwards -- the launches the roofline "
                         "table times, one per layer on the whole batch, in the kernel forms of the throughput run -- then exit; "
                         "scripts/pmc_traffic.py turns the two passes into bytes per forward and per launch of every kernel symbol")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="experiments: migan_set_tuning(KEY, VALUE) before the model is planned (the line records it in config.tuning)")
    ap.add_argument("--backend", type=str, default="nccl", choices=["nccl", "gloo"], help="gloo: only with --dry (CPU box)")
    ap.add_argument("--dry", action="store_true",
                    help="exercise rank spawning, the process group, the output gather and the JSON line without a GPU: the forward\n"
                         "is replaced by a copy (tests only; the line says so and carries no throughput claim)")
    a = ap.parse_args(argv)
    if a.reserve_cus:
        if a.reserve_cus % 8 or not (0 < a.reserve_cus < 256):
            ap.error("--reserve-cus must be a multiple of 8 between 8 and 248")
        a.streams = 1                                        # the library's own side streams carry no CU mask
    return a


def split_model(args):
    name, _, r = args.model.partition("-")
    if name not in ("migan", "comodgan") or not r.isdigit():
        raise SystemExit(f"--model must be migan-<R> or comodgan-<R>, got {args.model}")
    res = args.resolution or int(r)
    batch = args.batch or (16 if name == "comodgan" else 32)
    return name, res, batch


def roofline_from_launches(launches, ms_rounds, batch, gemm="f32"):
    """Group the per-launch hipEvent durations by kernel symbol and describe the dominant one."""
    ms = np.median(np.asarray(ms_rounds, dtype=np.float64), axis=0)
    groups = {}
    for L, t in zip(launches, ms):
        g = groups.setdefault(L["kernel"], dict(ms=0.0, flops=0.0, mfma=0.0, bytes=0.0, n=0, layers=[]))
        g["ms"] += t
        g["flops"] += L["flops"] * batch
        g["mfma"] += L["mfma_flops"] * batch
        g["bytes"] += L["bytes"] * batch
        # the first half of a fused down=2 layer is a plan entry that launches nothing: its bytes and flops belong to the fused kernel's launch
        g["n"] += 0 if (L["layer"].endswith(".dwfir") and "pipedown" in L["kernel"]) else 1
        g["layers"].append(L["layer"])
    peak_mfma = PEAK_BF16_MFMA_TFLOPS / MFMA_PRODUCTS[gemm] if MFMA_PRODUCTS.get(gemm) else PEAK_F32_MFMA_TFLOPS
    name, g = max(groups.items(), key=lambda kv: kv[1]["ms"])
    sec = g["ms"] * 1e-3
    t_mfma = g["mfma"] / (peak_mfma * 1e12)
    t_hbm = g["bytes"] / (PEAK_HBM_GBS * 1e9)
    if t_mfma >= t_hbm:
        bound, achieved, peak, unit = "mfma", g["mfma"] / sec / 1e12, round(peak_mfma, 1), "TFLOP/s"
    else:
        bound, achieved, peak, unit = "hbm", g["bytes"] / sec / 1e9, PEAK_HBM_GBS, "GB/s"
    total_ms = float(ms.sum())
    tot_mfma = sum(L["mfma_flops"] for L in launches) * batch
    tot_bytes = sum(L["bytes"] for L in launches) * batch
    roof = {
        "bound": bound, "achieved": round(achieved, 3), "peak": peak, "unit": unit,
        "frac": round(achieved / peak, 4), "traffic": None,
        "gemm": gemm, "dominant_mfma_tflops": round(g["mfma"] / sec / 1e12, 2),
        "frac_vs_fp32_mfma_peak": round(g["mfma"] / sec / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
        "dominant_hbm_gbs": round(g["bytes"] / sec / 1e9, 1),
        "kernel": name, "launches": g["n"], "avg_launch_ms": round(g["ms"] / g["n"], 4),
        "share_of_gpu_time": round(g["ms"] / total_ms, 4),
        "alg_per_launch": {"mfma_flop": g["mfma"] / g["n"], "bytes": g["bytes"] / g["n"]},
        "measured": "hipEvent pair around every launch, one stream, whole-batch launches (per-launch durations are not defined "
                    "while two sub-batches share the GPU)",
        "whole_forward": {
            "sum_kernel_ms": round(total_ms, 4),
            "mfma_tflops": round(tot_mfma / (total_ms * 1e-3) / 1e12, 3),
            "mfma_frac": round(tot_mfma / (total_ms * 1e-3) / 1e12 / peak_mfma, 4),          # of this GEMM variant's ceiling (algorithmic flops)
            "mfma_peak_tflops": round(peak_mfma, 1),
            "frac_vs_fp32_mfma_peak": round(tot_mfma / (total_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "hbm_gbs": round(tot_bytes / (total_ms * 1e-3) / 1e9, 1),
            "hbm_frac": round(tot_bytes / (total_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "alg_bytes": tot_bytes, "alg_mfma_flop": tot_mfma,
        },
        "per_kernel": {k: {"ms": round(v["ms"], 4), "launches": v["n"],
                           "mfma_tflops": round(v["mfma"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0.0,
                           "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0.0,
                           "alg_bytes_per_launch": round(v["bytes"] / v["n"]),
                           "hbm_frac": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if v["ms"] > 0 else 0.0,
                           "mfma_frac": round(v["mfma"] / (v["ms"] * 1e-3) / 1e12 / peak_mfma, 4) if v["ms"] > 0 else 0.0}
                       for k, v in groups.items()},
    }
    return roof


ACHIEVABLE_HBM_GBS = 6300.0    # MI355X_MICROARCH.md: "8 TB/s peak (spec); ~6.3 TB/s achievable" (6.29 TB/s measured with a float4 copy)


def forward_roofline(dom, ms_per_step):
    """The `roofline` object of the JSON line: the WHOLE forward -- sum of the algorithmic bytes (or matrix
    flops) of every launch / sum of the hipEvent durations of every launch -- against the roof that binds the forward; the dominant
    kernel symbol (what roofline_from_launches describes) rides along as `dominant` with its share of the GPU time, so that a change of
    which symbol is the largest cannot be read as a speed-up of the forward."""
    wf = dict(dom["whole_forward"])
    sec = wf["sum_kernel_ms"] * 1e-3
    t_mfma = wf["alg_mfma_flop"] / (wf["mfma_peak_tflops"] * 1e12)
    t_hbm = wf["alg_bytes"] / (PEAK_HBM_GBS * 1e9)
    wf["ms_per_step"] = round(ms_per_step, 4)
    wf["hbm_frac_of_timed_step"] = round(wf["alg_bytes"] / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
    wf["mfma_frac_of_timed_step"] = round(wf["alg_mfma_flop"] / (ms_per_step * 1e-3) / 1e12 / wf["mfma_peak_tflops"], 4)
    if t_mfma >= t_hbm:
        head = {"bound": "mfma", "achieved": wf["mfma_tflops"], "peak": wf["mfma_peak_tflops"], "unit": "TFLOP/s", "frac": wf["mfma_frac"],
                "frac_of_timed_step": wf["mfma_frac_of_timed_step"]}
    else:
        head = {"bound": "hbm", "achieved": wf["hbm_gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": wf["hbm_frac"],
                "frac_of_achievable": round(wf["hbm_gbs"] / ACHIEVABLE_HBM_GBS, 4), "achievable_peak": ACHIEVABLE_HBM_GBS,
                "frac_of_timed_step": wf["hbm_frac_of_timed_step"]}
    dominant = {k: dom[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "share_of_gpu_time",
                                    "alg_per_launch", "dominant_mfma_tflops", "dominant_hbm_gbs", "frac_vs_fp32_mfma_peak") if k in dom}
    out = dict(head)
    out["traffic"] = None                                    # whole-forward PMC bytes: attach_traffic_forward()
    out["scope"] = ("whole forward: sum over ALL launches of the algorithmic " + ("matrix flops" if head["bound"] == "mfma" else "bytes") +
                    " / sum of their hipEvent durations (one stream, whole-batch launches, kernel forms of the throughput run); "
                    "`dominant` = the kernel symbol with the largest share of that time, priced on its own")
    out["time_weighted_hbm_frac"] = wf["hbm_frac"]
    out["gemm"] = dom["gemm"]
    out["measured"] = dom["measured"]
    out["dominant"] = dominant
    out["whole_forward"] = wf
    out["per_kernel"] = dom["per_kernel"]
    return out


def finish_traffic(roof, dom):
    """after attach_traffic(dom, ...): carry the PMC figures over to the whole-forward object"""
    for k in ("traffic_stale", "traffic_measured_on", "traffic_note"):
        if k in dom:
            roof[k] = dom[k]
    for k in ("traffic", "traffic_over_alg", "traffic_source"):
        if k in dom:
            roof["dominant"][k] = dom[k]
    wf = dom["whole_forward"]
    if wf.get("pmc_bytes"):
        roof["traffic"] = wf["pmc_bytes"]
        roof["traffic_over_alg"] = wf["pmc_over_alg"]
        roof["whole_forward"]["pmc_bytes"], roof["whole_forward"]["pmc_over_alg"] = wf["pmc_bytes"], wf["pmc_over_alg"]
        roof["traffic_source"] = ("HBM bytes of ONE forward (all launches) from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                  "`bench.py --pmc-pass` (KiB units, gfx950 x2 read correction: scripts/pmc_traffic.py), table committed under profiles/")
    roof["per_kernel"] = dom["per_kernel"]


def flush_c_stdio():
    """RCCL prints a version banner through C stdio when a communicator is created; with stdout redirected that text sits in libc's
    buffer until exit and would land AFTER the JSON line.  Push it out now so that the JSON line stays the last line."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:   # pragma: no cover
        pass
    sys.stdout.flush()


def kernel_source_sha():
    """digest of the kernel + host sources the loaded library was built from (what a PMC traffic table must have been measured on)"""
    import hashlib
    csrc = os.path.join(ROOT, "mi-gan_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hpp", ".h", ".inc", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def attach_traffic(roof, path_rel, applies):
    """HBM bytes per launch from the PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    runs; scripts/pmc_traffic.py applies the guide's KiB unit and gfx950 x2 read correction), committed under profiles/.
    The table carries the digest of the kernel sources it was measured on: `traffic_stale` says whether that is the code
    that just ran.  A kernel symbol that is not in the file is reported, not silently dropped."""
    path = os.path.join(ROOT, path_rel)
    if not applies:
        roof["traffic_note"] = "no PMC traffic file for this configuration"
        return
    if not os.path.exists(path):
        roof["traffic_note"] = f"{path_rel} not found"
        return
    try:
        table = json.load(open(path))
    except Exception as e:   # pragma: no cover
        roof["traffic_note"] = f"{path_rel}: {e}"
        return
    meta = table.get("_meta", {})
    roof["traffic_stale"] = meta.get("kernel_source_sha") != kernel_source_sha()
    roof["traffic_measured_on"] = meta.get("kernel_source_sha")
    # Like for like: the table holds, per kernel symbol, the bytes and the dispatch count of ONE forward of exactly the
    # launches timed here (bench.py --pmc-pass); a symbol whose launch count differs from this run's is reported, not compared.
    for k, v in roof.get("per_kernel", {}).items():
        t = table.get(k)
        if not t:
            continue
        per_fwd = t.get("hbm_bytes_per_forward")
        if per_fwd is None:                                   # (a table of rounds 1-4: average per dispatch of another launch mix)
            v["pmc_bytes_per_launch"] = round(t["hbm_bytes_per_launch"])
            v["pmc_note"] = "old table format: average per dispatch, launch mix unknown"
            continue
        v["pmc_bytes_per_forward"] = round(per_fwd)
        v["pmc_launches_per_forward"] = t.get("launches_per_forward")
        if t.get("launches_per_forward") == v.get("launches") and v.get("alg_bytes_per_launch"):
            v["pmc_over_alg"] = round(per_fwd / (v["alg_bytes_per_launch"] * v["launches"]), 3)
        else:
            v["pmc_note"] = f"launch count differs: table {t.get('launches_per_forward')}, this run {v.get('launches')}"
    t = table.get(roof["kernel"])
    if not t:
        roof["traffic_note"] = f"kernel symbol not in {path_rel} (stale PMC file: re-run the `pmc` step of scripts/gpu_visit.sh); symbols there: {len(table)}"
        print(f"bench.py: warning: dominant kernel {roof['kernel']} has no entry in {path_rel}", file=sys.stderr)
        return
    if t.get("hbm_bytes_per_forward") is not None and t.get("launches_per_forward") == roof["launches"]:
        roof["traffic"] = round(t["hbm_bytes_per_forward"] / roof["launches"])
        roof["traffic_over_alg"] = round(t["hbm_bytes_per_forward"] / (roof["alg_per_launch"]["bytes"] * roof["launches"]), 3)
        roof["traffic_source"] = (f"{path_rel} (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE over the same {roof['launches']} launches per forward, "
                                  "divided by that count)")
    else:
        roof["traffic"] = round(t["hbm_bytes_per_launch"])
        roof["traffic_source"] = f"{path_rel} (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, avg per dispatch; launch mix of the table differs from this run)"
    tot = sum(t2.get("hbm_bytes_per_forward", 0.0) for k2, t2 in table.items() if k2 != "_meta" and k2 in roof.get("per_kernel", {}))
    if tot and all(k2 in table for k2 in roof.get("per_kernel", {})):
        roof["whole_forward"]["pmc_bytes"] = round(tot)
        roof["whole_forward"]["pmc_over_alg"] = round(tot / roof["whole_forward"]["alg_bytes"], 3)


# ------------------------------------------------------------------------------------------------------------------------
# workloads: build(dev) -> dict with step(), timed(), parity_and_cpu(), descriptions
def reference_migan(res, sd):
    """lib/model_zoo/migan_inference.py::Generator of the reference repository with these weights, or None where the repository
    is not present (MIGAN_REFERENCE or /root/reference; the GPU box has neither)"""
    root = os.environ.get("MIGAN_REFERENCE", "/root/reference")
    if not os.path.exists(os.path.join(root, "lib", "model_zoo", "migan_inference.py")):
        return None
    try:
        sys.dont_write_bytecode = True
        if root not in sys.path:
            sys.path.append(root)
        mod = importlib.import_module("lib.model_zoo.migan_inference")
        m = mod.Generator(resolution=res)
        m.load_state_dict({k: torch.from_numpy(np.array(v, copy=True)) for k, v in sd.items()}, strict=True)
        return m.eval()
    except Exception as e:   # pragma: no cover
        print(f"bench.py: reference module not usable ({type(e).__name__}: {e}); timing the port", file=sys.stderr)
        return None


def physical_cores():
    """distinct (socket, core) pairs of /proc/cpuinfo; the logical count where that cannot be read"""
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip() and phys is not None:
                pairs.add((phys, core))
                phys = core = None
        return len(pairs) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def all_cores_cpu_rate(model, threads, cap_s=60):
    """the CPU port on `threads` threads, one image after one warm-up image, in a child process with a time limit"""
    import subprocess
    code = ("import sys, time, os, importlib, torch; sys.path.insert(0, %r); pkg = importlib.import_module('mi-gan_amd'); "
            "from oracle import migan_torch_cpu as torc; res = %d; torch.set_num_threads(%d); "
            "sd = pkg.synth.make_state_dict(res, seed=0, regime='export'); x = pkg.synth.make_input(1, res, seed=100, kind='demo'); "
            "torc.generator(x, sd, res); "
            "t = time.perf_counter(); torc.generator(x, sd, res); print('RATE', 1 / (time.perf_counter() - t))"
            % (ROOT, model, threads))
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=cap_s)
        for line in r.stdout.splitlines():
            if line.startswith("RATE"):
                return round(float(line.split()[1]), 4), None
        return None, "child failed: " + r.stderr[-200:]
    except subprocess.TimeoutExpired:
        return None, f"two images (warm-up + timed) did not finish within the {cap_s} s cap on {threads} threads"


def build_migan(pkg, args, res, batch, dev, rank):
    for kv in args.tune:
        k, _, v = kv.partition("=")
        pkg.load_library().set_tuning(k, int(v))
    sd = pkg.synth.make_state_dict(res, seed=0, regime="export")
    model = pkg.Generator(resolution=res, activation_dtype=args.dtype, nan_policy=args.nan_policy)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    model = model.to(dev).eval()
    if args.dtype == "f32":
        model.set_gemm(args.gemm)
    model.set_streams(args.streams)
    # repeated inference on fixed weights: the 16-bit operand planes of the 1x1 weights are prepared once (first warm-up
    # step), like any weight packing; SURVEY 8d "weights pre-packed (packing excluded)"
    model.freeze_weights()
    # distinct images per rank (weak scaling), demo.py-style mask+image input
    x_np = pkg.synth.make_input(batch, res, seed=100 + rank, kind="demo")
    x = torch.from_numpy(x_np).to(dev)
    gemm = args.gemm if args.dtype == "f32" else "f16"

    def cpu_ref(n, threads, timed_runs=2):
        from oracle import migan_torch_cpu as torc
        torch.set_num_threads(threads)
        storage = None if args.dtype == "f32" else args.dtype
        ref = torc.generator(x_np[:n], sd, res, storage=storage)                    # warm-up + parity reference
        refmod = reference_migan(res, sd)                   # the reference's own module when its repository is present (not on the GPU box)
        fwd = (lambda: refmod(torch.from_numpy(x_np[:n]))) if refmod is not None else (lambda: torc.generator(x_np[:n], sd, res))
        with torch.no_grad():
            if refmod is not None:
                fwd()
            times = []
            for _ in range(timed_runs):
                c0 = time.perf_counter()
                fwd()
                times.append(time.perf_counter() - c0)
        ref32 = ref if storage is None else torc.generator(x_np[:n], sd, res)
        cpu_ref.kind = "reference" if refmod is not None else "port"
        return ref, ref32, float(np.median(times))

    step, out_shape, out_dtype, post = (lambda: model(x)), (batch, 3, res, res), torch.float32, None
    if args.io == "u8":
        img_np, mask_np = pkg.synth.make_uint8_input(batch, res, seed=100 + rank)      # the uint8 source of x_np
        img_u8, mask_u8 = torch.from_numpy(img_np).to(dev), torch.from_numpy(mask_np).to(dev)
        step, out_shape, out_dtype = (lambda: model.forward_uint8(img_u8, mask_u8)), (batch, res, res, 3), torch.uint8

        def post(ref_y, n):                                  # reference-side post-processing of the oracle's fp32 output
            from oracle import migan_prepost as pp
            return torch.from_numpy(pp.compose(ref_y.numpy() if hasattr(ref_y, "numpy") else ref_y, img_np[:n], mask_np[:n]).astype(np.int16))

    return dict(model=model, x=x, step=step, inplace=(args.io == "f32"), timed=lambda: model.forward_timed(x), launches=model.launch_info,
                gemm=gemm, cpu_ref=cpu_ref, out_shape=out_shape, out_dtype=out_dtype, post=post,
                cpu_desc=f"oracle/migan_torch_cpu.py (torch-CPU/oneDNN op-for-op port of the reference module)",
                traffic=(("profiles/pmc_traffic_latest.json", res == 512 and batch == 32 and args.dtype == "f32" and gemm == "f16x2")
                         if args.dtype == "f32" else
                         ("profiles/pmc_traffic_migan256_bf16_latest.json", res == 256 and batch == 32 and args.dtype == "bf16" and gemm == "f16")),
                data="synthetic (seeded export-like weights, demo.py-style mask+image batches)",
                gemm_text=GEMM_TEXT.get(gemm, gemm),
                extra_cfg=dict({"activation_storage": args.dtype, "streams": args.streams, **({"tuning": args.tune} if args.tune else {}),
                                "nan_policy": args.nan_policy + (" (a NaN activation stays a NaN, like Tensor.clamp in the reference module, :21-23)"
                                                                 if args.nan_policy == "propagate" else
                                                                 " (opt-in build: v_med3_f32 turns a NaN activation into -256, like the reference's CUDA plugin)"),
                                "weights": "static (migan_assume_static_weights: 1x1 operand planes prepared once)"},
                               **({"io": "uint8 HWC image + mask in, composed uint8 image out (scripts/demo.py:56-66,135-140 inside the "
                                         "first / last kernels); parity in uint8 steps against compose(oracle output)"}
                                  if args.io == "u8" else {})))


def build_comodgan(pkg, args, res, batch, dev, rank):
    cs, cm = pkg.comodgan_schema, pkg.comodgan
    cfg = cs.Config(resolution=res, num_ws=cs.default_num_ws(res))
    sd = pkg.synth.make_comodgan_state_dict(cfg, 0)
    model = cm.Generator(cm.Mapping(num_ws=cfg.num_ws), cm.Encoder(resolution=res), cm.Synthesis(resolution=res))
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    model = model.to(dev).eval()
    model.freeze_weights()
    x_np, z_np = pkg.synth.make_input(batch, res, seed=100 + rank), pkg.synth.make_latent(batch, 512, seed=100 + rank)
    x, z = torch.from_numpy(x_np).to(dev), torch.from_numpy(z_np).to(dev)

    def cpu_ref(n, threads, timed_runs=1):
        from oracle import comodgan_oracle as orc
        torch.set_num_threads(threads)
        # the batch-wide style normalisation (stylegan.py:139) is cancelled by the demodulation up to its epsilon, so a
        # sub-batch of the same images is the same computation per image
        ref = torch.from_numpy(orc.generator(x_np[:n], z_np[:n], sd, res, cfg.num_ws))
        c0 = time.perf_counter()
        orc.generator(x_np[:n], z_np[:n], sd, res, cfg.num_ws)
        return ref, ref, time.perf_counter() - c0

    return dict(model=model, x=x, step=lambda: model(x, z=z, noise_mode="const"), timed=lambda: model.forward_timed(x, z),
                launches=model.launch_info, gemm="f16x2", cpu_ref=cpu_ref, out_shape=(batch, 3, res, res),
                cpu_desc="oracle/comodgan_oracle.py (torch-CPU port of the reference module)",
                traffic=("profiles/pmc_traffic_comodgan_latest.json", res == 512 and batch == 16),
                data="synthetic (seeded N(0,1) weights, demo.py-style mask+image batches, fixed z, noise_mode=const)",
                gemm_text="3x3 convs as implicit GEMM on f16x2-split MFMA (3 fp16 products per fp32 product, fp32 accumulate)",
                extra_cfg={"noise_mode": "const (the reference default 'random' adds a torch.randn of every layer's noise per forward: "
                                         "measured as value_noise_random)",
                           "weights": "static (comodgan_assume_static_weights)"},
                alt={"key": "value_noise_random", "step": lambda: model(x),
                     "note": "model(x) exactly as scripts/demo.py:133-134 calls it: z = torch.randn([N, 512]) (comodgan.py:438-439) and "
                             "noise_mode='random' (torch.randn of every synthesis layer's noise plane per forward, stylegan.py:284-285), "
                             "both drawn on the GPU inside the timed step"})


def run_workload(args, rank, local_rank, world, dist, dev):
    """warm-up, K timed steps (barrier + synchronize on both sides, max over ranks), roofline and CPU baseline on rank 0"""
    pkg = importlib.import_module("mi-gan_amd")
    name, res, batch = split_model(args)
    wl = (build_comodgan if name == "comodgan" else build_migan)(pkg, args, res, batch, dev, rank)
    gather = world > 1 and not args.no_gather
    # N > 1: every step's output shards are all-gathered (RCCL) into one of two buffers; the gather of step i runs on
    # RCCL's stream while step i+1 computes, and every gather completes inside the timed region (fence()).
    gdt = gather_dtype_of(args, wl.get("out_dtype", torch.float32))
    # the migan forward with fp32 output writes straight into the collective's receive buffer (OutputGather.forward_and_submit: one in-place
    # collective per step, or one per sub-batch with --gather-mode parts); other producers (uint8 I/O, Co-Mod-GAN, a converted payload) copy
    # their shard into it first
    inplace = bool(gather and wl.get("inplace") and gdt == torch.float32 and not args.reserve_cus and args.gather_mode != "copy" and not args.dry)
    chunks = wl["model"].sub_batches(batch, dev) if (inplace and args.gather_mode == "parts") else None
    pipe = pkg.distributed.OutputGather(wl["out_shape"], gdt, dev, chunks=chunks) if gather else None
    # --reserve-cus K: the forward runs on a CU-masked stream so that the RCCL kernels of the overlapped gather find free CUs
    masked = pkg.distributed.cu_masked_stream(dev, args.reserve_cus) if (args.reserve_cus and not args.dry) else None

    occupy = None
    if args.occupy and not args.dry:
        import ctypes
        f = [int(v) for v in args.occupy.split(",")]
        o_wgs, o_us, o_thr, o_lds = f[0], f[1], (f[2] if len(f) > 2 else 256), (f[3] if len(f) > 3 else 0)
        o_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "ubench", "_bin", "liboccupy.so")
        if not os.path.exists(o_path):
            raise SystemExit(f"--occupy needs {o_path}: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/ubench/occupy.hip -o {o_path}")
        o_lib, o_stream = ctypes.CDLL(o_path), torch.cuda.Stream(dev)

        def occupy():
            o_stream.wait_stream(torch.cuda.current_stream(dev))
            rc = o_lib.occupy_launch(ctypes.c_void_p(o_stream.cuda_stream), o_wgs, o_thr, o_lds, o_us)
            if rc:
                raise RuntimeError(f"occupy_launch failed with {rc}")

    def step():
        if inplace:
            slot = pipe.forward_and_submit(wl["model"], wl["x"])
            if occupy is not None:
                occupy()
            return pipe.shards(slot)[0]
        if masked is not None:
            masked.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(masked):
                y = wl["step"]()
            torch.cuda.current_stream(dev).wait_stream(masked)
        else:
            y = wl["step"]()
        if gather:
            pipe.submit(y if y.dtype == gdt else y.to(gdt))
        if occupy is not None:
            occupy()
        return y

    def fence():
        if gather:
            pipe.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(fn, steps):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = fn()
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, y

    if args.pmc_pass:
        with torch.no_grad():
            for _ in range(args.pmc_pass + 2):
                wl["timed"]()
        torch.cuda.synchronize()
        if rank == 0:
            print(json.dumps({"pmc_pass": True, "forwards": args.pmc_pass + 2, "launches_per_forward": len(wl["launches"]())}))
        return None
    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        elapsed, y = timed_loop(step, args.steps)
        compute_only = None
        if gather:                                           # the same steps without the collective, for the record
            for _ in range(2):
                wl["step"]()
            el2, _ = timed_loop(wl["step"], args.steps)
            compute_only = el2 / args.steps * 1e3
        alt = None
        if wl.get("alt") and world == 1:                     # the same workload the way the reference's demo.py calls it
            for _ in range(2):
                wl["alt"]["step"]()
            el3, _ = timed_loop(wl["alt"]["step"], max(3, args.steps // 2))
            alt = {"value": round(batch * max(3, args.steps // 2) / el3, 2), "ms_per_step": round(el3 / max(3, args.steps // 2) * 1e3, 4),
                   "note": wl["alt"]["note"]}
    ms_per_step = elapsed / args.steps * 1e3
    value = world * batch * args.steps / elapsed
    ranks_seen = world
    if world > 1:
        ids = torch.full((1,), rank, dtype=torch.int64, device=dev)
        got = torch.empty((world,), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(got, ids)
        ranks_seen = int(torch.unique(got).numel())
    if rank != 0:
        return None
    # ---- roofline of the dominant kernel: hipEvent pair around every launch (same stream) -----
    rounds = []
    with torch.no_grad():
        for i in range(3 + min(args.steps, 5)):
            _, ms = wl["timed"]()
            if i >= 3:
                rounds.append(ms)
    launches = wl["launches"]()
    dom = roofline_from_launches(launches, rounds, batch, wl["gemm"])
    attach_traffic(dom, *wl["traffic"])
    # the line's `roofline` describes the whole forward (time-weighted over all launches); the dominant kernel symbol is a sub-object
    roof = forward_roofline(dom, ms_per_step)
    finish_traffic(roof, dom)
    if args.dump_layers:
        med = np.median(np.asarray(rounds), axis=0)
        with open(args.dump_layers, "w") as f:
            json.dump([dict(L, ms=float(t)) for L, t in zip(launches, med)], f, indent=1)
    # ---- CPU baseline + parity on the same inputs (rank 0, N = 1 protocol) ---------------------
    cpu = parity = parity32 = envelope = ymax = None
    if args.cpu_images > 0 and world == 1:      # the CPU baseline is an N = 1 measurement (rank 0 only)
        n = min(args.cpu_images if name == "migan" else max(1, args.cpu_images // 4), batch)
        cores = os.cpu_count() or 1
        threads = min(args.cpu_threads or cores, cores)
        ref, ref32, sec = wl["cpu_ref"](n, threads)
        cpu = {"value": round(n / sec, 4), "unit": "images/sec", "cores": int(threads), "kind": getattr(wl["cpu_ref"], "kind", "port"),
               "sample": f"{n} image(s) of the same {name}-{res} batch, fp32, {wl['cpu_desc']}, timed after 1 warm-up, "
                         f"{threads} threads (the fastest setting measured on this class of host), host has {cores} logical cores"}
        if cpu["kind"] == "reference":
            cpu["sample"] = cpu["sample"].replace(wl["cpu_desc"], "lib/model_zoo/migan_inference.py::Generator of the reference repository itself")
        if name == "migan" and cores > threads and (args.cpu_all_cores or primary_default(args)):
            # north star: "the node's host cores (core count stated)": every PHYSICAL core beside the fastest setting, in a capped child
            # process (default run: 60 s).  One thread per logical core oversubscribes the port's small oneDNN ops -- measured in round 3
            # on the 256-thread GPU-box host: 0.014 images/s, 70 s per image -- and is what --cpu-all-cores times instead.
            phys = cores if args.cpu_all_cores else min(cores, physical_cores())
            rate, why = all_cores_cpu_rate(res, phys, cap_s=600 if args.cpu_all_cores else 60)
            cpu["value_all_cores"], cpu["all_cores"] = rate, phys
            cpu["all_cores_note"] = why or (f"one image after a warm-up image in a child process with torch.set_num_threads({phys}) "
                                            f"({'logical' if phys == cores else 'physical'} cores of {cores} hardware threads); `value` "
                                            f"({threads} threads) is the fastest setting measured on this class of host")
        if wl.get("post"):                                   # uint8 output: compare in uint8 steps with the composed oracle output
            parity = float((y[:n].cpu().to(torch.int16) - wl["post"](ref, n)).abs().max())
            parity32 = float((y[:n].cpu().to(torch.int16) - wl["post"](ref32, n)).abs().max())
        else:
            parity = float((y[:n].cpu() - ref).abs().max())
            parity32 = float((y[:n].cpu() - ref32).abs().max())
        envelope = float((ref - ref32).abs().max())
        ymax = float(ref32.abs().max())
    tag = BASELINE_CONFIG.get((name, res, args.dtype), "")
    out = {
        "metric": f"images/sec {name}-{res} generator fwd" + (" + fused uint8 pre/post-processing" if wl.get("post") else ""),
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype if name == "migan" else "f32", "data": wl["data"],
        "config": dict({"workload": f"{name}-{res} generator forward, batch={batch} per GPU, {res}x{res}, "
                                    f"{'fp32' if args.dtype == 'f32' or name != 'migan' else args.dtype + ' activation storage, fp32 arithmetic'}"
                                    + (f" ({tag})" if tag else ""),
                        "global_batch": world * batch, "resolution": res, "gemm": wl["gemm_text"],
                        "parallelism": f"batch-shard x{world}" + (" + RCCL all-gather of every step's outputs, overlapped with the next step" if gather else "")},
                       **wl["extra_cfg"]),
        "max_abs_vs_ref": parity,
        "roofline": roof,
        "cpu_baseline": cpu,
        "rccl_ranks": ranks_seen,
        "device": torch.cuda.get_device_name(local_rank),
    }
    out["arith"] = {"f32": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate",
                    "bf16x3": "bf16x3-split MFMA (6 bf16 products per fp32 product), fp32 accumulate",
                    "f16x2": "fp16x2-split MFMA, fp32 accumulate",
                    "f16": "fp16 MFMA on fp16-rounded operands, fp32 accumulate"}.get(wl["gemm"], wl["gemm"]) + \
                   "; depthwise 3x3, FIR, activations, epilogues in fp32 VALU"
    if name == "migan" and world == 1 and not args.no_latency and args.io == "f32":
        try:
            lat, y1, x1 = latency_batch1(wl["model"], pkg, res, dev)
            if args.cpu_images > 0 and not getattr(args, "is_secondary", False):      # (the secondary lines report the time only)
                from oracle import migan_torch_cpu as torc
                sd1 = pkg.synth.make_state_dict(res, seed=0, regime="export")
                ref1 = torc.generator(x1, sd1, res, storage=None if args.dtype == "f32" else args.dtype)
                lat["max_abs_vs_ref"] = float((y1.cpu() - ref1).abs().max())
            out["latency_b1_ms"] = lat.pop("latency_b1_ms")
            out["latency_b1"] = lat
        except Exception as e:   # pragma: no cover
            out["latency_b1"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and (args.force_pg or primary_default(args)):
        try:
            out["rccl_world1"] = rccl_world1(wl, pkg, batch, dev, max(3, min(args.steps, 10)))
        except Exception as e:
            out["rccl_world1"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
    if parity32 is not None and args.dtype != "f32" and name == "migan":
        out["max_abs_vs_ref"] = parity
        out["max_abs_vs_fp32_ref"] = parity32
        out["storage_mode_envelope"] = envelope
        out["ref_abs_max"] = ymax
        out["tolerance_abs"] = 2.0 * envelope
        out["parity_note"] = ("max_abs_vs_ref: against the torch-CPU oracle in the same storage mode -- which reproduces BIT FOR BIT what the "
                              "reference module gives when forward hooks round every stored feature map to the storage format "
                              "(tests/golden/make_golden_bf16.py, tests/test_oracle_golden.py); max_abs_vs_fp32_ref: against the fp32 reference; "
                              "storage_mode_envelope: oracle(mode) vs oracle(fp32) on the same inputs = the quantisation noise of the mode; "
                              "tolerance_abs = 2 x that envelope, absolute (a rounding step turns a 1-ulp summation-order difference into a "
                              "full quantisation step, so two correct implementations of the mode agree to the noise level, not to the bit)")
    if parity is not None:
        # (inside `config` as well: the driver's parsed record keeps `config` whole and lists everything else as extra keys)
        tol = out.get("tolerance_abs", 1.0 if wl.get("post") else 1e-3)
        out["config"]["parity"] = {"max_abs_vs_ref": parity, "tolerance_abs": tol, "within_tolerance": bool(parity <= tol), "images_checked": n,
                                   "against": ("the reference module itself" if cpu and cpu["kind"] == "reference" else
                                               "the torch-CPU port of the reference module (bit-exact with it on the committed goldens)")
                                              + (", composed to uint8 like scripts/demo.py:135-140 (unit: uint8 steps)" if wl.get("post") else "")}
    if wl.get("post"):
        out["parity_unit"] = "uint8 steps of the composed image (scripts/demo.py:135-140 applied to the oracle's fp32 output)"
    if alt is not None:
        out[wl["alt"]["key"]] = alt
    if gather:
        out["compute_only_ms_per_step"] = round(compute_only, 4)
        out["gather_mb_per_rank_per_step"] = round(float(np.prod(wl["out_shape"])) * torch.empty(0, dtype=gdt).element_size() / 1e6, 1)
        out["gather_dtype"] = str(gdt).replace("torch.", "")
        out["gather"] = {"mode": args.gather_mode if inplace else "copy", "in_place": inplace, "collectives_per_step": len(chunks) if chunks else 1,
                         "note": ("the forward writes its images into this rank's slice of all_gather_into_tensor's receive buffer (no local copy), "
                                  + ("one collective per sub-batch, enqueued behind that sub-batch (migan_forward_parts)" if chunks else
                                     "one in-place collective per step") + ", overlapped with the next step")
                                 if inplace else "the step's output is copied into the receive buffer, then gathered in place (overlapped with the next step)"}
    if args.reserve_cus:
        out["reserved_cus"] = args.reserve_cus
    if args.occupy:
        out["occupy"] = args.occupy
    return out


def primary_default(args):
    """the driver's default line: BASELINE configs[2] with nothing overridden"""
    return (args.model == "migan-512" and not args.resolution and not args.batch and args.dtype == "f32" and args.gemm == "f16x2"
            and args.io == "f32" and not getattr(args, "is_secondary", False) and not args.tune and not args.no_secondary and args.streams == 2
            and args.nan_policy == "propagate")


def latency_batch1(wl_model, pkg, res, dev, n=60):
    """scripts/demo.py:122-134 calls the model one image at a time: hipEvent time of a batch-1 forward (input on device -> output on
    device), median / min / p90 of n after 10 warm-up calls, and its parity against the CPU port"""
    x_np = pkg.synth.make_input(1, res, seed=900, kind="demo")
    x = torch.from_numpy(x_np).to(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    with torch.no_grad():
        for _ in range(10):
            y = wl_model(x)
        torch.cuda.synchronize()
        for a, b in ev:
            a.record()
            y = wl_model(x)
            b.record()
        torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return {"latency_b1_ms": round(ms[len(ms) // 2], 4), "min_ms": round(ms[0], 4), "p90_ms": round(ms[int(0.9 * (len(ms) - 1))], 4), "calls": n,
            "measured": "hipEvent pair on the caller's stream around model(x), x = [1,4,R,R] resident in HBM"}, y, x_np


def rccl_world1(wl, pkg, batch, dev, steps):
    """a one-rank NCCL (= RCCL) process group on this GPU: the pipelined all-gather of bench.py --gpus N at world size 1"""
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(free_port())
    created = not dist.is_initialized()
    if created:
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    try:
        warm = torch.zeros(8, device=dev)
        dist.all_reduce(warm)                               # creates the communicator (and prints RCCL's banner) now
        torch.cuda.synchronize()
        flush_c_stdio()
        inplace = bool(wl.get("inplace")) and wl.get("out_dtype", torch.float32) == torch.float32
        chunks = wl["model"].sub_batches(batch, dev) if inplace else None
        odt = wl.get("out_dtype", torch.float32)

        def run(mode):
            # parts: one in-place collective per sub-batch (forward_parts); inplace: the forward writes the receive buffer, one collective per
            # step; copy: rounds 2-5 -- the step's output tensor is copied into the receive buffer, then gathered
            if mode == "parts":
                pipe = pkg.distributed.OutputGather(wl["out_shape"], odt, dev, chunks=chunks)
                submit = lambda: pipe.forward_and_submit(wl["model"], wl["x"])
            elif mode == "inplace":
                pipe = pkg.distributed.OutputGather(wl["out_shape"], odt, dev)
                submit = lambda: pipe.forward_and_submit(wl["model"], wl["x"])
            else:
                pipe = pkg.distributed.OutputGather(wl["out_shape"], odt, dev)
                submit = lambda: pipe.submit(wl["step"]())
            y = wl["step"]()
            slot = submit()
            same = bool(torch.equal(pipe.result(slot), y))
            for _ in range(2):
                submit()
            pipe.drain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                submit()
            pipe.drain()
            torch.cuda.synchronize()
            return same, time.perf_counter() - t0

        def plain():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                wl["step"]()
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        with torch.no_grad():
            modes = (["parts", "inplace"] if inplace else []) + ["copy"]
            res = {}
            for mode in modes:
                # each mode beside its own run of the same steps without the collective, back to back on the same clocks: what the gather of
                # step i, queued behind the persistent one-workgroup-per-CU kernels of step i + 1, adds to a step
                same_m, el_m = run(mode)
                el0_m = plain()
                res[mode] = {"gathered_equals_forward": same_m, "ms_per_step": round(el_m / steps * 1e3, 4),
                             "ms_per_step_without_gather": round(el0_m / steps * 1e3, 4), "gather_exposed_ms": round((el_m - el0_m) / steps * 1e3, 4)}
            default_mode = "inplace" if inplace else "copy"
            same = all(r["gathered_equals_forward"] for r in res.values())
            el, el0 = res[default_mode]["ms_per_step"] * steps / 1e3, res[default_mode]["ms_per_step_without_gather"] * steps / 1e3
        return {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "gathered_equals_forward": same,
                "ms_per_step": round(el / steps * 1e3, 4), "steps": steps,
                "ms_per_step_without_gather": round(el0 / steps * 1e3, 4), "gather_exposed_ms": round((el - el0) / steps * 1e3, 4),
                "gather_mb_per_step": round(float(np.prod(wl["out_shape"])) * torch.empty((), dtype=wl.get("out_dtype", torch.float32)).element_size() / 1e6, 2),
                "in_place": inplace, "collectives_per_step": 1, "mode": default_mode, "modes": res,
                "what": "init_process_group('nccl', world_size=1, device_id=...) + OutputGather.forward_and_submit per step: the forward writes into this "
                        "rank's slice of the receive buffer of all_gather_into_tensor, one in-place collective per step on RCCL's stream (`modes`: the "
                        "same per sub-batch, and the copying form).  At world size 1 an in-place all-gather has nothing to move: this leg shows that "
                        "the path runs, that the result is the forward's bits and what the submission costs a step -- not xGMI time or the contention "
                        "of a real collective (profiles/r05_contention.md measured that with a stand-in kernel)"}
    finally:
        if created:
            dist.destroy_process_group()


def secondary_line(base_args, **over):
    """run another workload with the same protocol in this process and return its (trimmed) line"""
    a = argparse.Namespace(**vars(base_args))
    a.dump_layers = ""
    a.is_secondary = True
    for k, v in over.items():
        setattr(a, k, v)
    try:
        out = run_workload(a, 0, 0, 1, None, torch.device("cuda", 0))
    except Exception as e:   # a secondary workload must not take the primary line down with it
        return {"model": a.model, "error": f"{type(e).__name__}: {e}"}
    out["roofline"].pop("per_kernel", None)
    return out


def gather_dtype_of(args, out_dtype):
    """what the ranks exchange (--gather-dtype)"""
    if args.gather_dtype == "auto":
        return out_dtype
    if args.gather_dtype == "u8":
        if args.io != "u8":
            raise SystemExit("--gather-dtype u8 needs --io u8 (the composed uint8 image is what migan_forward_u8 writes)")
        return torch.uint8
    if args.io == "u8":
        raise SystemExit("--io u8 exchanges uint8 shards; --gather-dtype f32 / f16 apply to the fp32 output")
    return torch.float16 if args.gather_dtype == "f16" else torch.float32


def dry_run(args, rank, world, dist):
    """--dry: the launch / rendezvous / gather / JSON plumbing on any backend, without the HIP library"""
    pkg = importlib.import_module("mi-gan_amd")
    name, res, batch = split_model(args)
    dev = torch.device("cpu")
    u8 = args.io == "u8"
    shape = (batch, 8, 8, 3) if u8 else (batch, 3, 8, 8)
    y = torch.full(shape, rank, dtype=torch.uint8 if u8 else torch.float32)
    gather = world > 1 and not args.no_gather
    gdt = gather_dtype_of(args, y.dtype)
    pipe = pkg.distributed.OutputGather(shape, gdt, dev) if gather else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if gather:
            slot = pipe.submit(y.to(gdt))
    if gather:
        pipe.drain()
        full = pipe.result(slot)
        assert full.shape[0] == world * batch and float(full[-1, 0, 0, 0]) == world - 1
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    ranks_seen = world
    if world > 1:
        got = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(got, torch.tensor([rank], dtype=torch.int64))
        ranks_seen = len({int(g) for g in got})
    if rank != 0:
        return None
    return {"metric": f"images/sec {name}-{res} generator fwd", "value": 0.0, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(el / max(1, args.steps) * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "none (--dry: launch plumbing only, no forward was run)",
            "config": {"workload": "dry run", "global_batch": world * batch, "parallelism": f"batch-shard x{world}"},
            "dry": True, "rccl_ranks": ranks_seen, "backend": args.backend, "gather_dtype": str(gdt).replace("torch.", ""),
            "gather_mb_per_rank_per_step": round(float(np.prod(shape)) * torch.empty(0, dtype=gdt).element_size() / 1e6, 6)}


def worker(rank, local_rank, world, args):
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry:
        if world > 1:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
        out = dry_run(args, rank, world, dist)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
        if args.backend != "nccl":
            raise SystemExit("--backend gloo is for --dry only; GPU runs use nccl (= RCCL on ROCm)")
        if world > 1 and torch.cuda.device_count() < world and "LOCAL_RANK" not in os.environ:
            raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
            warm = torch.zeros(8, device=dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
            flush_c_stdio()
        t_wall = time.perf_counter()
        out = run_workload(args, rank, local_rank, world, dist, dev)
        t_primary = time.perf_counter() - t_wall
        name, res, batch = split_model(args)
        if out is not None and world == 1 and not args.no_secondary and args.model == "migan-512" and not args.resolution \
                and not args.batch and args.dtype == "f32" and args.gemm == "f16x2" and args.io == "f32" and args.nan_policy == "propagate":
            # the default driver run: also time the exact-fp32-MFMA variant of the same workload and the other two
            # single-GPU BASELINE configs, same protocol
            ex = secondary_line(args, gemm="f32", cpu_images=0, steps=max(5, args.steps // 2), warmup=3)
            out["value_exact_f32"] = ex.get("value")
            out["config"]["value_exact_f32"] = ex.get("value")      # (also inside `config`, which the driver's parsed record keeps whole)
            out["exact_f32"] = {k: ex.get(k) for k in ("value", "ms_per_step", "error") if k in ex}
            if "roofline" in ex:
                out["exact_f32"]["whole_forward"] = ex["roofline"]["whole_forward"]
                out["exact_f32"]["frac_vs_fp32_mfma_peak"] = round(
                    ex["roofline"]["whole_forward"]["alg_mfma_flop"] / (ex["ms_per_step"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
                out["exact_f32"]["note"] = ("same workload with the 1x1 convs on v_mfma_f32_32x32x2_f32 (exact fp32 products): the number "
                                            "comparable to SURVEY's 5 940 images/s fp32-MFMA ceiling")
            # the opt-in NaN -> -256 build of the same library (no NaN test in the kernels): what the reference-faithful default costs
            nc = secondary_line(args, nan_policy="clamp", cpu_images=0, steps=max(10, args.steps), warmup=5, no_latency=True)
            out["value_nan_clamp"] = nc.get("value")
            out["config"]["value_nan_clamp"] = nc.get("value")
            out["nan_clamp"] = {"value": nc.get("value"), "ms_per_step": nc.get("ms_per_step"), **({"error": nc["error"]} if "error" in nc else {}),
                                "note": "same workload, same protocol, Generator(nan_policy='clamp') = libmigan_hip_nanclamp.so (-DMIGAN_NAN_CLAMP); "
                                        "`value` is the default library, whose clamp propagates a NaN like the reference module's Tensor.clamp"}
            t_exact = time.perf_counter() - t_wall - t_primary
            out["secondary"] = [
                secondary_line(args, model="migan-256", dtype="bf16", steps=max(10, args.steps), cpu_images=2),
                secondary_line(args, model="comodgan-512", steps=max(5, args.steps // 2), warmup=3, cpu_images=4),
                # the primary workload with demo.py's pre/post-processing fused in (uint8 in, composed uint8 out)
                secondary_line(args, io="u8", steps=max(10, args.steps // 2), warmup=3, cpu_images=2),
            ]
            # where the wall time of this command went (the timed region itself is steps x ms_per_step)
            out["wall_s"] = {"primary_incl_cpu_baseline_latency_rccl": round(t_primary, 1), "exact_f32": round(t_exact, 1),
                             "secondary": round(time.perf_counter() - t_wall - t_primary - t_exact, 1),
                             "since_process_start": round(time.perf_counter() - T_PROCESS_START, 1)}
    ok = True
    if out is not None:
        if out.get("rccl_ranks") != args.gpus or out.get("n_gpus") != args.gpus:
            out["error"] = f"--gpus {args.gpus} but {out.get('rccl_ranks')} rank(s) took part (n_gpus {out.get('n_gpus')})"
            ok = False
        flush_c_stdio()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return ok


def _spawned(local_rank, world, port, argv):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ok = worker(local_rank, local_rank, world, parse(argv))
    if not ok:
        raise SystemExit(3)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "WORLD_SIZE" in os.environ:                        # launched as one of N ranks (torch.distributed.run)
        world = int(os.environ["WORLD_SIZE"])
        ok = worker(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world, args)
        if not ok:
            raise SystemExit(3)
        return
    if args.gpus <= 1:
        if not worker(0, 0, 1, args):
            raise SystemExit(3)
        return
    # --gpus N without a launcher: create the N ranks here, one process per GPU (the reference's own launcher does the same
    # for training: main.py:27 mp.spawn, lib/utils.py:41-46 init_process_group on tcp://127.0.0.1)
    import torch.multiprocessing as mp
    mp.spawn(_spawned, args=(args.gpus, free_port(), argv), nprocs=args.gpus, join=True)


if __name__ == "__main__":
    main()
