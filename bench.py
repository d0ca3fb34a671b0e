#!/usr/bin/env python3
"""Benchmark of the north-star path: images/sec of the migan-512 generator forward, batch 32 per
GPU, fp32, on MI355X -- BASELINE.json's metric/config.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one Generator.forward over one synthetic batch already resident in HBM (input tensor on
device -> output tensor on device).  For N > 1 every rank runs its own batch of 32 (weak scaling:
BASELINE config 4 = 256 images over 8 GPUs) and each step ends with the RCCL all-gather of the
output shards, inside the timed region.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     dominant kernel (largest share of GPU time), measured live with hipEvent pairs around
               every launch on the launch stream: achieved = algorithmic flops (or bytes) of its
               launches / their summed duration, against the MI355X peak of the binding resource.
  cpu_baseline the torch-CPU port of the reference module (oracle/migan_torch_cpu.py; the reference
               itself is Python and is not present on the GPU box) timed on the host cores on a small
               sample of the same inputs -- reported, not the target.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 peak
PEAK_BF16_MFMA_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak
# Split GEMM variants: every fp32 product costs six bf16 MFMA products ("bf16x3") or three fp16 MFMA
# products ("f16x2"), so the matrix-core ceiling in ALGORITHMIC flops is 2500 / 6 resp. 2500 / 3
MFMA_PRODUCTS = {"f32": None, "bf16x3": 6.0, "f16x2": 3.0}
GEMM_TEXT = {"f32": "1x1 convs on exact fp32 MFMA",
             "bf16x3": "1x1 convs on bf16x3-split MFMA (6 bf16 products per fp32 product, fp32 accumulate)",
             "f16x2": "1x1 convs on f16x2-split MFMA (3 fp16 products per fp32 product on scaled operands, fp32 accumulate)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--resolution", type=int, default=512)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--cpu-images", type=int, default=4, help="sample size of the CPU baseline (0 = skip)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the output all-gather")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads of the CPU baseline (0 = all logical cores); 16 is the fastest setting measured for this\n"
                         "graph of small oneDNN convs on the 256-thread GPU-box host (8: 1.04, 16: 0.94, 32: 1.07, 64: 1.72, 128: 4.4 s/img)")
    ap.add_argument("--dump-layers", type=str, default="", help="write per-launch hipEvent durations to this JSON file")
    return ap.parse_args()


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
        g["n"] += 1
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
        "whole_forward": {
            "sum_kernel_ms": round(total_ms, 4),
            "mfma_tflops": round(tot_mfma / (total_ms * 1e-3) / 1e12, 3),
            "mfma_frac": round(tot_mfma / (total_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "hbm_gbs": round(tot_bytes / (total_ms * 1e-3) / 1e9, 1),
            "hbm_frac": round(tot_bytes / (total_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
        },
        "per_kernel": {k: {"ms": round(v["ms"], 4), "launches": v["n"],
                           "mfma_tflops": round(v["mfma"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0.0,
                           "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0.0}
                       for k, v in groups.items()},
    }
    return roof


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)

    pkg = importlib.import_module("mi-gan_amd")
    R, B = args.resolution, args.batch
    sd = pkg.synth.make_state_dict(R, seed=0, regime="export")
    model = pkg.Generator(resolution=R)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    model = model.to(dev).eval()
    # distinct images per rank (weak scaling), demo.py-style mask+image input
    x_np = pkg.synth.make_input(B, R, seed=100 + rank, kind="demo")
    x = torch.from_numpy(x_np).to(dev)
    gather = world > 1 and not args.no_gather
    # N > 1: every step's output shards are all-gathered (RCCL) into one of two buffers; the gather of step i
    # runs on RCCL's stream while step i+1 computes, and every gather completes inside the timed region (fence()).
    pipe = pkg.distributed.OutputGather((B, 3, R, R), torch.float32, dev) if gather else None

    def step():
        y = model(x)
        if gather:
            pipe.submit(y)
        return y

    def fence():
        if gather:
            pipe.drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = step()
        fence()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel: hipEvent pair around every launch (same stream) -----
        launches = model.launch_info()
        rounds = []
        with torch.no_grad():
            for i in range(3 + min(args.steps, 5)):
                _, ms = model.forward_timed(x)
                if i >= 3:
                    rounds.append(ms)
        gemm = model._lib.gemm_variant()
        roof = roofline_from_launches(launches, rounds, B, gemm)
        # HBM bytes per launch of the dominant kernel from the PMC passes of this same command
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs; scripts/pmc_traffic.py applies the
        # guide's KiB unit and gfx950 x2 read correction).  Committed under profiles/; null if absent.
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
        if R == 512 and B == 32 and os.path.exists(tpath):
            try:
                t = json.load(open(tpath)).get(roof["kernel"])
                if t:
                    roof["traffic"] = round(t["hbm_bytes_per_launch"])
                    roof["traffic_source"] = "profiles/pmc_traffic_latest.json (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, avg per launch)"
            except Exception:
                pass
        if args.dump_layers:
            med = np.median(np.asarray(rounds), axis=0)
            with open(args.dump_layers, "w") as f:
                json.dump([dict(L, ms=float(t)) for L, t in zip(launches, med)], f, indent=1)

        # ---- CPU baseline + parity on the same inputs (rank 0, N = 1 protocol) ---------------------
        cpu = None
        parity = None
        if args.cpu_images > 0 and world == 1:      # the CPU baseline is an N = 1 measurement (rank 0 only)
            from oracle import migan_torch_cpu as torc
            n = min(args.cpu_images, B)
            xs = x_np[:n]
            torch.set_num_threads(min(args.cpu_threads or (os.cpu_count() or 1), os.cpu_count() or 1))
            ref = torc.generator(xs, sd, R)                    # warm-up + parity reference
            times = []
            for _ in range(2):
                c0 = time.perf_counter()
                torc.generator(xs, sd, R)
                times.append(time.perf_counter() - c0)
            cpu = {"value": round(n / float(np.median(times)), 4), "unit": "images/sec",
                   "cores": int(torch.get_num_threads()), "kind": "port",
                   "sample": f"{n} images of the same migan-{R} batch, fp32, oracle/migan_torch_cpu.py "
                             f"(torch-CPU/oneDNN op-for-op port of the reference module), median of 2 after 1 warm-up, "
                             f"host has {os.cpu_count()} logical cores"}
            parity = float((y[:n].cpu() - ref).abs().max())

        out = {
            "metric": "images/sec migan-512 generator fwd" if R == 512 else f"images/sec migan-{R} generator fwd",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (seeded export-like weights, demo.py-style mask+image batches)",
            "config": {"workload": f"migan-{R} generator forward, batch={B} per GPU, {R}x{R}, fp32 (BASELINE configs[2])",
                       "global_batch": world * B, "resolution": R,
                       "gemm": GEMM_TEXT.get(gemm, gemm),
                       "parallelism": f"batch-shard x{world}" + (" + RCCL all-gather of every step's outputs, overlapped with the next step" if gather else "")},
            "max_abs_vs_ref": parity,
            "roofline": roof,
            "cpu_baseline": cpu,
            "device": torch.cuda.get_device_name(local_rank),
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
