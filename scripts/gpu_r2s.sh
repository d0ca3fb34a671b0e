#!/bin/bash
# FIR-up 64-channel tile at 3 workgroups per CU with the input prefetch issued after the depthwise stage (w3 bit 4).  -> gpurun_out/r2s/
set -u
OUT=gpurun_out/r2s
mkdir -p $OUT
timeout 600 python scripts/sweep.py --steps 12 --layers --only base_s1,w3_7_s1,base_s2,w3_7_s2 --out $OUT/sweep512.json > $OUT/sweep512.log 2>&1; grep -v amdgpu $OUT/sweep512.log | grep "img/s\|ERROR"
timeout 600 python -m pytest tests/test_gpu_round2.py -q -k "three_workgroup" > $OUT/pytest_w3.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_w3.log
