#!/bin/bash
set -u
OUT=gpurun_out/r2i
mkdir -p $OUT
timeout 600 python scripts/sweep.py --steps 12 --layers --only base_s1,s2_stag10,nopersist_s1,bf16_s2 --out $OUT/sweep512.json > $OUT/sweep512.log 2>&1; grep -v amdgpu $OUT/sweep512.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q -x -k "up or generator or determin or sepconv_operator" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
