#!/bin/bash
# Round 2: default plan (fused-FromRGB layer as 3-workgroup tiles) vs w3=0, fp32 and bf16, then every GPU test.  -> gpurun_out/r2n/
set -u
OUT=gpurun_out/r2n
mkdir -p $OUT
timeout 600 python scripts/sweep.py --steps 12 --layers --only base_s1,w3_none_s1,s2_stag10,base_s2,w3_none_s2,bf16_s1,bf16_w3_none_s1,bf16_s2,bf16_w3_none_s2 --out $OUT/sweep512.json > $OUT/sweep512.log 2>&1; grep -v amdgpu $OUT/sweep512.log | grep "img/s\|ERROR"
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
