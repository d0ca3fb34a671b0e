#!/bin/bash
# round 5, visit a: sepconv_wide2_kernel on hardware -- parity, per-layer A/B against the 128-pixel tile, stage ablation, headline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5a; mkdir -p $OUT
bash scripts/gpu_visit.sh r5a "tests=tests/test_gpu_wide2.py -m gpu" "layers" "layers=--tune w2=0" "layers=--streams 1" 
ABL=$R/mi-gan_amd/csrc/libmigan_hip_ablate.so
for k in 0 1 2 4 8 16 32 48 64; do
  MIGAN_HIP_LIBRARY=$ABL MIGAN_ABLATE=$k timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 5 --warmup 2 --dump-layers $OUT/abl_$k.json > $OUT/abl_$k.out 2> $OUT/abl_$k.err; echo "ablate $k rc=$?"
done
bash scripts/gpu_visit.sh r5a "bench=--no-secondary --cpu-images 2"
