#!/bin/bash
# Round-end check on the GPU box: every GPU test, smoke(), the north-star bench line (with CPU baseline and per-launch dump) and a
# rocprofv3 kernel-trace summary of it.  -> gpurun_out/final/
set -u
OUT=gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log
timeout 200 python bench.py --dump-layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
head -c 900 $OUT/bench.json; echo; tail -2 $OUT/bench.err
timeout 150 python scripts/bench_comodgan.py --dump-layers $OUT/comodgan_layers.json > $OUT/comodgan_bench.json 2> $OUT/comodgan_bench.err; echo "comodgan bench rc=$?"; head -c 420 $OUT/comodgan_bench.json; echo
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o trace --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --cpu-images 0 > $R/$OUT/trace.log 2>&1; echo "trace rc=$?"
