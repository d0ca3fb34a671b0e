#!/usr/bin/env python3
"""Measurement of the second model (BASELINE.json configs[4]): images/sec of the comodgan-512
generator forward, batch 16, fp32, one MI355X.  Thin front end of `bench.py --model comodgan-512` (same protocol and
JSON schema; the default `bench.py` run also reports this workload under "secondary").

    python scripts/bench_comodgan.py [--batch 16] [--steps 10] [--warmup 3] [--cpu-images 4] [--dump-layers f.json]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench

if __name__ == "__main__":
    argv = sys.argv[1:]
    if not any(a.startswith("--model") for a in argv):
        argv = ["--model", "comodgan-512"] + argv
    if not any(a.startswith("--steps") for a in argv):
        argv += ["--steps", "10", "--warmup", "3"]
    bench.main(argv)
