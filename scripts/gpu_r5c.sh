#!/bin/bash
# round 5, visit c: W2 epilogue variants (dword / 16-byte stores, nontemporal / plain), ablation of the 16-byte form
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5c; mkdir -p $OUT
L() { tag=$1; shift; timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 10 --warmup 3 --dump-layers $OUT/layers_$tag.json "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "$tag rc=$?"; }
L w2_1 --tune w2=1
L w2_2 --tune w2=2
L w2_3 --tune w2=3
L w2_4 --tune w2=4
L w2_0 --tune w2=0
L w2_2b --tune w2=2
ABL=$R/mi-gan_amd/csrc/libmigan_hip_ablate.so
for k in 0 1 2 4 8 48; do
  MIGAN_HIP_LIBRARY=$ABL MIGAN_ABLATE=$k timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 5 --warmup 2 --streams 1 --tune w2=2 --dump-layers $OUT/abl_$k.json > $OUT/abl_$k.out 2> $OUT/abl_$k.err; echo "ablate $k rc=$?"
done
timeout 600 python -m pytest tests/test_gpu_wide2.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
