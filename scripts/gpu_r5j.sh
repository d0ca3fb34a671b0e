#!/bin/bash
# round 5, visit j: W2 with the weight planes DMA'd as whole 32-channel chunks (variant 1)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5j; mkdir -p $OUT
L() { tag=$1; shift; timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 10 --warmup 3 --dump-layers $OUT/layers_$tag.json "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "$tag rc=$?"; }
L w2_1 --tune w2=1
L w2_2 --tune w2=2
L w2_0 --tune w2=0
L w2_2b --tune w2=2
L w2_1b --tune w2=1
for v in 2; do
MIGAN_HIP_LIBRARY=$R/mi-gan_amd/csrc/libmigan_hip_prof.so timeout 300 python scripts/phase_profile.py 512 32 w2=$v > $OUT/phase_$v.txt 2> $OUT/phase_$v.err; echo "phase rc=$?"
grep -E "wide2" $OUT/phase_$v.txt
done
timeout 600 python -m pytest tests/test_gpu_wide2.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
