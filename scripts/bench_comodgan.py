#!/usr/bin/env python3
"""Measurement of the second model (BASELINE.json configs[4]): images/sec of the comodgan-512
generator forward, batch 16, fp32, one MI355X; same protocol and JSON shape as bench.py (which stays on the
north-star metric).  `roofline`: hipEvent pair around every launch, dominant kernel = largest share of GPU time,
against the f16x2-split matrix-core ceiling (2500/3 TFLOP/s algorithmic) or HBM.  `cpu_baseline`: the torch-CPU
oracle (oracle/comodgan_oracle.py, a port of the reference module) on a small sample of the same batch.

    python scripts/bench_comodgan.py [--resolution 512] [--batch 16] [--steps 10] [--warmup 3] [--cpu-images 1]
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench as north_star


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--resolution", type=int, default=512)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-images", type=int, default=1)
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--dump-layers", type=str, default="")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    dev = torch.device("cuda", 0)
    pkg = importlib.import_module("mi-gan_amd")
    cs, cm = pkg.comodgan_schema, pkg.comodgan
    R, B = args.resolution, args.batch
    cfg = cs.Config(resolution=R, num_ws=cs.default_num_ws(R))
    sd = pkg.synth.make_comodgan_state_dict(cfg, 0)
    model = cm.Generator(cm.Mapping(num_ws=cfg.num_ws), cm.Encoder(resolution=R), cm.Synthesis(resolution=R))
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    model = model.to(dev).eval()
    x_np, z_np = pkg.synth.make_input(B, R, seed=100), pkg.synth.make_latent(B, 512, seed=100)
    x, z = torch.from_numpy(x_np).to(dev), torch.from_numpy(z_np).to(dev)
    with torch.no_grad():
        for _ in range(args.warmup):
            y = model(x, z=z, noise_mode="const")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = model(x, z=z, noise_mode="const")
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        rounds = []
        for i in range(5):
            _, ms = model.forward_timed(x, z)
            if i >= 2:
                rounds.append(ms)
    launches = model.launch_info()
    roof = north_star.roofline_from_launches(launches, rounds, B, "f16x2")
    # HBM bytes per launch of the dominant kernel from the PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    # separate runs, scripts/gpu_comodgan_traffic.sh + scripts/pmc_traffic.py); committed under profiles/, null if absent
    tpath = os.path.join(ROOT, "profiles", "r01_comodgan_v12_pmc_traffic.json")
    if R == 512 and B == 16 and os.path.exists(tpath):
        try:
            t = json.load(open(tpath)).get(roof["kernel"])
            if t:
                roof["traffic"] = round(t["hbm_bytes_per_launch"])
                roof["traffic_source"] = "profiles/r01_comodgan_v12_pmc_traffic.json (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, avg per launch)"
        except Exception:
            pass
    if args.dump_layers:
        med = np.median(np.asarray(rounds), axis=0)
        with open(args.dump_layers, "w") as f:
            json.dump([dict(L, ms=float(t)) for L, t in zip(launches, med)], f, indent=1)
    cpu = parity = None
    if args.cpu_images > 0:
        from oracle import comodgan_oracle as orc
        n = min(args.cpu_images, B)
        torch.set_num_threads(min(args.cpu_threads, os.cpu_count() or 1))
        # the batch-wide style normalisation (stylegan.py:139) is cancelled by the demodulation up to its epsilon, so a
        # sub-batch of the same images is the same computation per image
        ref = orc.generator(x_np[:n], z_np[:n], sd, R, cfg.num_ws)
        c0 = time.perf_counter()
        orc.generator(x_np[:n], z_np[:n], sd, R, cfg.num_ws)
        dt = time.perf_counter() - c0
        cpu = {"value": round(n / dt, 4), "unit": "images/sec", "cores": int(torch.get_num_threads()), "kind": "port",
               "sample": f"{n} image(s) of the same comodgan-{R} batch, fp32, oracle/comodgan_oracle.py (torch-CPU port of the reference "
                         f"module), 1 timed run after 1 warm-up, host has {os.cpu_count()} logical cores"}
        parity = float(np.abs(y[:n].cpu().numpy() - ref).max())
    out = {"metric": f"images/sec comodgan-{R} generator fwd", "value": round(B * args.steps / elapsed, 2), "unit": "images/sec", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic (seeded N(0,1) weights, demo.py-style mask+image batches, fixed z, noise_mode=const)",
           "config": {"workload": f"comodgan-{R} generator forward, batch={B}, {R}x{R}, fp32 (BASELINE configs[4])", "global_batch": B, "resolution": R,
                      "gemm": "3x3 convs as implicit GEMM on f16x2-split MFMA (3 fp16 products per fp32 product, fp32 accumulate)"},
           "max_abs_vs_ref": parity, "roofline": roof, "cpu_baseline": cpu, "device": torch.cuda.get_device_name(0)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
