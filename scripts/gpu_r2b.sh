#!/bin/bash
# Round 2, GPU visit B: 16-bit storage diagnostics (per-layer taps), the round-2 parity tests, full GPU suite.
set -u
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
for dt in bf16 f16; do
  timeout 300 python scripts/gpu_diag_taps.py 64 2 31 $dt > $OUT/taps64_$dt.log 2>&1; echo "taps64 $dt rc=$?"
done
timeout 300 python scripts/gpu_diag_taps.py 256 2 31 bf16 > $OUT/taps256_bf16.log 2>&1; echo "taps256 rc=$?"
head -60 $OUT/taps64_bf16.log
timeout 900 python -m pytest tests/test_gpu_round2.py -q -k "16bit" -x > $OUT/pytest_16bit.log 2>&1; echo "pytest16 rc=$?"; tail -30 $OUT/pytest_16bit.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -40 $OUT/pytest_gpu.log
