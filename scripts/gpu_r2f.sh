#!/bin/bash
set -u
OUT=gpurun_out/r2f
mkdir -p $OUT
for args in "256 32 bf16" "512 32 bf16" "512 32 f16"; do
  echo "=== $args"; timeout 600 python scripts/gpu_diag_streams.py $args 2>&1 | grep -v amdgpu.ids | tee -a $OUT/streams2.log
done
timeout 900 python -m pytest tests/test_gpu_round2.py -q -k "deterministic or two_streams or 16bit_storage_full or uint8" > $OUT/pytest_det.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_det.log
