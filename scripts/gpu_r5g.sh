#!/bin/bash
# round 5, visit g: W2 with a DMA lookahead of two sub-steps (variants 8 / 9) against the first form
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5g; mkdir -p $OUT
L() { tag=$1; shift; timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 10 --warmup 3 --dump-layers $OUT/layers_$tag.json "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "$tag rc=$?"; }
L w2_1 --tune w2=1
L w2_9 --tune w2=9
L w2_10 --tune w2=10
L w2_0 --tune w2=0
L w2_9b --tune w2=9
for v in 9; do
MIGAN_HIP_LIBRARY=$R/mi-gan_amd/csrc/libmigan_hip_prof.so timeout 300 python scripts/phase_profile.py 512 32 w2=$v > $OUT/phase_$v.txt 2> $OUT/phase_$v.err; echo "phase rc=$?"
grep -E "wide2" $OUT/phase_$v.txt
done
timeout 600 python -m pytest tests/test_gpu_wide2.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
