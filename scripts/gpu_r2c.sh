#!/bin/bash
set -u
OUT=gpurun_out/r2c
mkdir -p $OUT
for args in "128 256 bf16 prev" "128 256 bf16" "128 256 f32 prev" "64 512 bf16 prev" "128 64 bf16 prev"; do
  echo "=== $args"; timeout 300 python scripts/gpu_diag_torgb.py $args 2>&1 | grep -v amdgpu.ids | tee -a $OUT/torgb.log
done
