#!/bin/bash
set -u
mkdir -p gpurun_out/abl
export MIGAN_HIP_LIBRARY=$PWD/mi-gan_amd/csrc/libmigan_hip_ablate.so
for g in bf16x3 f32; do
for a in 0 1 2 4 8 16 12 28 30; do
  MIGAN_GEMM=$g MIGAN_ABLATE=$a timeout 300 python bench.py --steps 5 --warmup 2 --cpu-images 0 --dump-layers gpurun_out/abl/layers_${g}_$a.json > gpurun_out/abl/b_${g}_$a.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/abl/b_${g}_$a.json')); print('$g ablate=$a', d['ms_per_step'])"
done
done
