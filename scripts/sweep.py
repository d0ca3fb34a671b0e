#!/usr/bin/env python3
"""A/B sweep of plan variants on one MI355X: images/s of the generator forward (inputs resident in HBM, static weights) for a
list of tuning settings, in one process.  Each variant also checks parity of 2 images against the torch-CPU port once.

    python scripts/sweep.py [--model migan-512] [--batch 32] [--steps 10] [--out gpurun_out/sweep.json] [--only label,label]
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

DEFAULTS = dict(kc16=0, kc16_minw=3, w3=3, wide=3, nt256=1, persist_min=8192, persist_grid=512, single_b=0, stagger=-1, stagger_pct=22)

VARIANTS = [
    # label, tuning overrides, streams, dtype
    ("base_s1", {}, 1, "f32"),
    ("base_s2", {}, 2, "f32"),
    ("base_s3", {}, 3, "f32"),
    ("bf16_s3", {}, 3, "bf16"),
] + [(f"s2_stag{k}", {"stagger": k}, 2, "f32") for k in (8, 10, 12, 14, 16, 18, 22)] + [
    (f"s4_stag{k}", {"stagger": k}, 4, "f32") for k in (3, 5, 7, 9, 11, 14)] + [
    ("kc16_plain_w3_s1", {"kc16": 1, "kc16_minw": 3}, 1, "f32"),
    ("kc16_plain_w3_s2", {"kc16": 1, "kc16_minw": 3, "stagger": 14}, 2, "f32"),
    ("nopersist_s1", {"persist_min": 1 << 30}, 1, "f32"),
    ("w3_2_s1", {"w3": 2}, 1, "f32"),
    ("w3_2_s2", {"w3": 2}, 2, "f32"),
    ("bf16_w3_2_s1", {"w3": 2}, 1, "bf16"),
    ("bf16_w3_2_s2", {"w3": 2}, 2, "bf16"),
    ("singleb_nopersist_s1", {"single_b": 1, "persist_min": 1 << 30}, 1, "f32"),
    ("singleb_s1", {"single_b": 1}, 1, "f32"),
    ("bf16_singleb_nopersist_s1", {"single_b": 1, "persist_min": 1 << 30}, 1, "bf16"),
    ("w3_7_s1", {"w3": 7}, 1, "f32"),
    ("w3_7_s2", {"w3": 7}, 2, "f32"),
    ("w3_none_s1", {"w3": 0}, 1, "f32"),
    ("w3_none_s2", {"w3": 0}, 2, "f32"),
    ("bf16_w3_none_s1", {"w3": 0}, 1, "bf16"),
    ("bf16_w3_none_s2", {"w3": 0}, 2, "bf16"),
    ("w3_plain_s1", {"w3": 1}, 1, "f32"),
    ("w3_rgb_s1", {"w3": 2}, 1, "f32"),
    ("w3_up_s1", {"w3": 4}, 1, "f32"),
    ("w3_all_s1", {"w3": 7}, 1, "f32"),
    ("w3_plain_s2", {"w3": 1}, 2, "f32"),
    ("w3_all_s2", {"w3": 7}, 2, "f32"),
    ("bf16_nopersist_s1", {"persist_min": 1 << 30}, 1, "bf16"),
    ("bf16_s1", {}, 1, "bf16"),
    ("bf16_s2", {}, 2, "bf16"),
    ("bf16_s2_stag8", {"stagger": 8}, 2, "bf16"),
    ("bf16_s2_stag12", {"stagger": 12}, 2, "bf16"),
    ("bf16_s2_stag14", {"stagger": 14}, 2, "bf16"),
    ("bf16_s4_stag7", {"stagger": 7}, 4, "bf16"),
    ("f16_s2", {}, 2, "f16"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="migan-512")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--out", default="")
    ap.add_argument("--only", default="")
    ap.add_argument("--layers", action="store_true", help="also print per-launch hipEvent durations of every variant")
    args = ap.parse_args()
    pkg = importlib.import_module("mi-gan_amd")
    lib = pkg.load_library()
    from oracle import migan_torch_cpu as torc
    res = int(args.model.split("-")[1])
    dev = torch.device("cuda", 0)
    sd = pkg.synth.make_state_dict(res, seed=0, regime="export")
    x_np = pkg.synth.make_input(args.batch, res, seed=100, kind="demo")
    x = torch.from_numpy(x_np).to(dev)
    ref = {}
    rows = []
    only = set(filter(None, args.only.split(",")))
    for label, tune, streams, dtype in VARIANTS:
        if only and label not in only:
            continue
        for k, v in DEFAULTS.items():
            lib.set_tuning(k, v)
        for k, v in tune.items():
            lib.set_tuning(k, v)
        m = pkg.Generator(resolution=res, activation_dtype=dtype)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        m = m.to(dev).eval()
        m.set_streams(streams)
        m.freeze_weights()
        try:
            with torch.no_grad():
                for _ in range(args.warmup):
                    y = m(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    y = m(x)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                _, ms = m.forward_timed(x)
                _, ms = m.forward_timed(x)
        except Exception as e:
            rows.append(dict(label=label, error=str(e)))
            print(f"{label:22s} ERROR {e}", flush=True)
            continue
        if dtype not in ref:
            ref[dtype] = torc.generator(x_np[:2], sd, res, storage=None if dtype == "f32" else dtype)     # (16-bit: GEMM variant "f16")
        err = float((y[:2].cpu() - ref[dtype]).abs().max())
        row = dict(label=label, tuning=tune, streams=streams, dtype=dtype, ms_per_step=el / args.steps * 1e3,
                   images_per_s=args.batch * args.steps / el, sum_kernel_ms=float(np.sum(ms)), max_abs_err=err)
        if args.layers:
            row["layers"] = [dict(layer=L["layer"], kernel=L["kernel"], ms=float(t)) for L, t in zip(m.launch_info(), ms)]
        rows.append(row)
        print(f"{label:22s} {row['images_per_s']:9.1f} img/s  {row['ms_per_step']:7.3f} ms/step  sum of kernels (1 stream) {row['sum_kernel_ms']:7.3f} ms"
              f"  err {err:.2e}", flush=True)
        del m
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
