#!/bin/bash
# single-buffered 1x1 weight tiles / one-tile workgroups for the pointwise (second half of down=2) layers.  -> gpurun_out/r2q/
set -u
OUT=gpurun_out/r2q
mkdir -p $OUT
timeout 600 python scripts/sweep.py --steps 12 --layers --only base_s1,singleb_nopersist_s1,singleb_s1,bf16_s1,bf16_singleb_nopersist_s1 --out $OUT/sweep512.json > $OUT/sweep512.log 2>&1; grep -v amdgpu $OUT/sweep512.log | grep "img/s\|ERROR"
