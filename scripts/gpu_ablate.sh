#!/bin/bash
# Ablation of the standard sepconv kernels (measurement build with -DMIGAN_ABLATE): which stage costs what.
# bits: 1 no epilogue stores, 2 no epilogue, 4 no depthwise stage, 8 no MFMA, 16 no global input loads
set -u
mkdir -p gpurun_out/abl
export MIGAN_HIP_LIBRARY=$PWD/mi-gan_amd/csrc/libmigan_hip_ablate.so
for a in 0 1 2 4 8 16 12 28 30; do
  MIGAN_WIDE=0 MIGAN_ABLATE=$a timeout 300 python bench.py --steps 5 --warmup 2 --cpu-images 0 --dump-layers gpurun_out/abl/layers_$a.json > gpurun_out/abl/b_$a.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/abl/b_$a.json')); print('ablate=$a', d['ms_per_step'])"
done
