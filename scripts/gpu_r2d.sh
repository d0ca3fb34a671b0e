#!/bin/bash
set -u
OUT=gpurun_out/r2d
mkdir -p $OUT
export REPS=6
for e in 1 2 3; do
  for args in "128 256 bf16 prev" "128 512 bf16 prev" "64 512 bf16 prev"; do
  echo "=== exp $e: $args"; MIGAN_HIP_LIBRARY=$PWD/mi-gan_amd/csrc/libmigan_exp$e.so timeout 300 python scripts/gpu_diag_torgb.py $args 2>&1 | grep -v amdgpu.ids | tail -n +5 | head -7 | tee -a $OUT/exp2.log
  done
done
