#!/bin/bash
set -u
OUT=gpurun_out/r2g
mkdir -p $OUT
timeout 600 python scripts/sweep.py --steps 12 --out $OUT/sweep512.json > $OUT/sweep512.log 2>&1; grep -v amdgpu $OUT/sweep512.log
timeout 300 python scripts/sweep.py --steps 12 --model migan-256 --only base_s1,base_s2,s2_stag10,s2_stag14,s4_stag5,s4_stag9,bf16_s1,bf16_s2,bf16_s2_stag14,bf16_s4_stag7 --out $OUT/sweep256.json > $OUT/sweep256.log 2>&1; grep -v amdgpu $OUT/sweep256.log
