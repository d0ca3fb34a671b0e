#!/bin/bash
# Round 2, GPU visit E: the fused-ToRGB rounding fix (determinism over repeated launches), full GPU suite, per-layer timings.
set -u
OUT=gpurun_out/r2e
mkdir -p $OUT
export TMPDIR=/tmp REPS=10
for args in "128 256 bf16 prev" "128 256 f16 prev" "64 512 bf16 prev" "256 128 bf16 prev" "256 128 f16 prev" "128 512 bf16 prev"; do
  echo "=== $args"; timeout 300 python scripts/gpu_diag_torgb.py $args 2>&1 | grep -v amdgpu.ids | head -14 | tee -a $OUT/torgb.log
done
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -25 $OUT/pytest_gpu.log
timeout 600 python scripts/sweep.py --layers --only base_s1,bf16_s1,f16_s2,bf16_s2,s2_stag14 --out $OUT/sweep512.json > $OUT/sweep512.log 2>&1; tail -6 $OUT/sweep512.log
timeout 300 python scripts/sweep.py --model migan-256 --layers --only base_s1,base_s2,bf16_s1,bf16_s2,f16_s2 --out $OUT/sweep256.json > $OUT/sweep256.log 2>&1; tail -6 $OUT/sweep256.log
