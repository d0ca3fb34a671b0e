#!/bin/bash
# round 5, visit k: W2 variants -- activation + split on the MFMA waves (bit 1), weight planes as 32-channel chunks (bit 0)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5k; mkdir -p $OUT
L() { tag=$1; shift; timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 10 --warmup 3 --dump-layers $OUT/layers_$tag.json "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "$tag rc=$?"; }
L w2_2 --tune w2=2
L w2_3 --tune w2=3
L w2_4 --tune w2=4
L w2_0 --tune w2=0
L w2_4b --tune w2=4
L w2_2b --tune w2=2
for v in 4; do
MIGAN_HIP_LIBRARY=$R/mi-gan_amd/csrc/libmigan_hip_prof.so timeout 300 python scripts/phase_profile.py 512 32 w2=$v > $OUT/phase_$v.txt 2> $OUT/phase_$v.err; echo "phase rc=$?"
grep -E "wide2" $OUT/phase_$v.txt
done
timeout 600 python -m pytest tests/test_gpu_wide2.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
