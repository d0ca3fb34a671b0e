#!/bin/bash
set -u
OUT=gpurun_out/r2j
mkdir -p $OUT
timeout 600 python scripts/sweep.py --steps 12 --layers --only base_s1,s2_stag10,bf16_s1,bf16_s2,f16_s2 --out $OUT/sweep512.json > $OUT/sweep512.log 2>&1; grep -v amdgpu $OUT/sweep512.log
timeout 300 python scripts/sweep.py --steps 12 --layers --model migan-256 --only base_s1,s2_stag10,bf16_s1,bf16_s2,f16_s2 --out $OUT/sweep256.json > $OUT/sweep256.log 2>&1; grep -v amdgpu $OUT/sweep256.log
timeout 1200 python -m pytest tests/test_gpu_round2.py -q -k "16bit or determin or any_size or uint8 or two_streams" > $OUT/pytest16.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest16.log; grep "gemm f16\|gemm f16x2" $OUT/pytest16.log | head
