#!/bin/bash
# round 5, visit o: the pointwise form of W2 for the un-fused down=2 layers
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5o; mkdir -p $OUT
L() { tag=$1; shift; timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 10 --warmup 3 --dump-layers $OUT/layers_$tag.json "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "$tag rc=$?"; }
L pw1
L pw0 --tune w2_pw=0
L pw1b
L pw0b --tune w2_pw=0
timeout 600 python -m pytest tests/test_gpu_wide2.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
