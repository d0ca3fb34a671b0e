#!/bin/bash
# per-launch times at batch 1, 2, 4 (the scripts/demo.py regime).  -> gpurun_out/r2v/
set -u
OUT=gpurun_out/r2v
mkdir -p $OUT
for b in 1 4; do
  timeout 200 python scripts/sweep.py --batch $b --steps 30 --warmup 5 --layers --only base_s1 --out $OUT/b$b.json > $OUT/b$b.log 2>&1; grep -v amdgpu $OUT/b$b.log | grep "img/s\|ERROR"
  timeout 200 python scripts/sweep.py --batch $b --steps 30 --warmup 5 --layers --only bf16_s1 --out $OUT/b${b}_bf16.json > $OUT/b${b}_bf16.log 2>&1; grep -v amdgpu $OUT/b${b}_bf16.log | grep "img/s\|ERROR"
done
