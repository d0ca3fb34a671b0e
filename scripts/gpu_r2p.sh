#!/bin/bash
# plain / ToRGB 64-channel tiles as 2-chunk unrolled 3-workgroup tiles (w3 bit 1) on top of the default.  -> gpurun_out/r2p/
set -u
OUT=gpurun_out/r2p
mkdir -p $OUT
timeout 600 python scripts/sweep.py --steps 12 --layers --only base_s1,w3_3_s1,base_s2,w3_3_s2,bf16_s1,bf16_w3_3_s1,bf16_s2,bf16_w3_3_s2 --out $OUT/sweep512.json > $OUT/sweep512.log 2>&1; grep -v amdgpu $OUT/sweep512.log | grep "img/s\|ERROR"
timeout 600 python -m pytest tests/test_gpu_round2.py -q -k "three_workgroup or determin" > $OUT/pytest_w3.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_w3.log
