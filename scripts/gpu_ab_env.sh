#!/bin/bash
# A/B of one tuning knob: gpu_ab_env.sh TAG VAR val1 val2 ...   (bench only, per-layer dumps)
set -u
TAG=$1; VAR=$2; shift 2
mkdir -p gpurun_out/$TAG
for v in "$@"; do
  env $VAR=$v timeout 600 python bench.py --steps 10 --warmup 3 --cpu-images 0 --dump-layers gpurun_out/$TAG/layers_$v.json > gpurun_out/$TAG/bench_$v.json 2> gpurun_out/$TAG/bench_$v.err
  python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_$v.json')); print('$VAR=$v', d['value'], 'img/s', d['ms_per_step'], 'ms/step')" || tail -3 gpurun_out/$TAG/bench_$v.err
done
