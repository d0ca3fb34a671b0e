#!/bin/bash
# A/B of tuning knobs / alternative builds in one visit.  usage: gpu_ab.sh TAG "name1:ENV=V ..." ...
set -u
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $envs timeout 600 python bench.py --steps 10 --warmup 3 --cpu-images 0 --dump-layers gpurun_out/$TAG/layers_$name.json > gpurun_out/$TAG/bench_$name.json 2> gpurun_out/$TAG/bench_$name.err
  python -c "import json; d=json.load(open('gpurun_out/$TAG/bench_$name.json')); print('$name', d['value'], 'img/s', d['ms_per_step'], 'ms/step')" || tail -3 gpurun_out/$TAG/bench_$name.err
done
