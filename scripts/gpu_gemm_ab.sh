#!/bin/bash
set -u
mkdir -p gpurun_out/gemm
export TMPDIR=/tmp
MIGAN_GEMM=bf16x3 timeout 900 python -m pytest tests -m gpu -q > gpurun_out/gemm/pytest_bf16x3.log 2>&1; tail -4 gpurun_out/gemm/pytest_bf16x3.log
for v in f32 bf16x3; do
  MIGAN_GEMM=$v timeout 600 python bench.py --steps 10 --warmup 3 --cpu-images 2 --dump-layers gpurun_out/gemm/layers_$v.json > gpurun_out/gemm/bench_$v.json 2> gpurun_out/gemm/bench_$v.err
  python -c "import json; d=json.load(open('gpurun_out/gemm/bench_$v.json')); print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms/step  max_abs_vs_ref', d['max_abs_vs_ref'])" || tail -3 gpurun_out/gemm/bench_$v.err
done
