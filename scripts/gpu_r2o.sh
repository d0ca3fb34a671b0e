#!/bin/bash
# persistent vs one-tile-per-workgroup variants per layer, with the FromRGB layer already on 3-workgroup tiles.  -> gpurun_out/r2o/
set -u
OUT=gpurun_out/r2o
mkdir -p $OUT
timeout 600 python scripts/sweep.py --steps 12 --layers --only base_s1,nopersist_s1,bf16_s1,bf16_nopersist_s1 --out $OUT/sweep512.json > $OUT/sweep512.log 2>&1; grep -v amdgpu $OUT/sweep512.log | grep "img/s\|ERROR"
timeout 300 python scripts/sweep.py --steps 12 --layers --model migan-256 --only base_s1,nopersist_s1,bf16_s1,bf16_nopersist_s1 --out $OUT/sweep256.json > $OUT/sweep256.log 2>&1; grep -v amdgpu $OUT/sweep256.log | grep "img/s\|ERROR"
