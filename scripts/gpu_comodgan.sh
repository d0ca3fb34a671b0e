#!/bin/bash
# GPU-box visit for the Co-Mod-GAN path: parity tests, bench line with per-launch dump, rocprofv3 kernel trace + stats.
# Usage: scripts/gpu_comodgan.sh TAG [extra pytest args]   -> gpurun_out/comodgan_TAG/
set -u
TAG=${1:-run}
OUT=gpurun_out/comodgan_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_comodgan.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -15 $OUT/pytest.log
timeout 600 python scripts/bench_comodgan.py --dump-layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
head -c 1500 $OUT/bench.json; echo; tail -3 $OUT/bench.err
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o trace --output-format csv -- python $R/scripts/bench_comodgan.py --steps 3 --warmup 1 --cpu-images 0 > $R/$OUT/trace.log 2>&1; echo "trace rc=$?"
cd $R; ls $OUT/trace | head
