#!/bin/bash
# round 5, visit b: W2 with the DMA issue under the depthwise rows; start-up stagger; half grids with two streams; ablation
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5b; mkdir -p $OUT
L() { tag=$1; shift; timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 10 --warmup 3 --dump-layers $OUT/layers_$tag.json "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "$tag rc=$?"; }
L base
L base_s1 --streams 1
L stag20 --tune w2_stagger=20
L stag40 --tune w2_stagger=40
L stag80 --tune w2_stagger=80
L stag160 --tune w2_stagger=160
L g128 --tune pipe_grid=128
L g128_s1 --tune pipe_grid=128 --streams 1
L g128_s4 --tune pipe_grid=128 --streams 4
L g64_s4 --tune pipe_grid=64 --streams 4
ABL=$R/mi-gan_amd/csrc/libmigan_hip_ablate.so
for k in 0 1 4 8 48; do
  MIGAN_HIP_LIBRARY=$ABL MIGAN_ABLATE=$k timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 5 --warmup 2 --streams 1 --dump-layers $OUT/abl_$k.json > $OUT/abl_$k.out 2> $OUT/abl_$k.err; echo "ablate $k rc=$?"
  MIGAN_HIP_LIBRARY=$ABL MIGAN_ABLATE=$k timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 5 --warmup 2 --streams 1 --tune pipe_grid=128 --dump-layers $OUT/abl_g128_$k.json > $OUT/abl_g128_$k.out 2> $OUT/abl_g128_$k.err; echo "ablate g128 $k rc=$?"
done
timeout 600 python -m pytest tests/test_gpu_wide2.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
