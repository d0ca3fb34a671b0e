#!/bin/bash
set -u
mkdir -p gpurun_out/bs
for b in 1 2 4 8 16 32 64; do
  timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --cpu-images 0 --dump-layers gpurun_out/bs/layers_$b.json > gpurun_out/bs/b_$b.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/bs/b_$b.json')); print('batch $b', d['value'], 'img/s', round(d['ms_per_step']/$b,4), 'ms/img')"
done
