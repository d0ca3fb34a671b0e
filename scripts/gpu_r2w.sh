#!/bin/bash
# 16-bit storage + "f16" GEMM: dwfir writes the pointwise GEMM's fp16 A operand directly (half the intermediate).  -> gpurun_out/r2w/
set -u
OUT=gpurun_out/r2w
mkdir -p $OUT
timeout 300 python scripts/sweep.py --steps 15 --layers --model migan-256 --only bf16_s1,bf16_s2,f16_s2 --out $OUT/sweep256.json > $OUT/sweep256.log 2>&1; grep -v amdgpu $OUT/sweep256.log | grep "img/s\|ERROR"
timeout 300 python scripts/sweep.py --steps 15 --layers --only bf16_s1,bf16_s2 --out $OUT/sweep512.json > $OUT/sweep512.log 2>&1; grep -v amdgpu $OUT/sweep512.log | grep "img/s\|ERROR"
timeout 900 python -m pytest tests/test_gpu_round2.py -q -k "16bit or determin or storage" > $OUT/pytest16.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest16.log
