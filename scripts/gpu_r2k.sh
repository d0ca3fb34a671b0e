#!/bin/bash
# Round 2 verification visit: every GPU test, smoke(), the default bench line (primary + exact-f32 + secondaries, per-launch dump),
# then rocprofv3 kernel-trace stats + HBM-traffic PMC passes of the migan-256 bf16 workload with the "f16" GEMM variant.
# -> gpurun_out/r2k/
set -u
OUT=gpurun_out/r2k
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log
timeout 400 python bench.py --dump-layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
head -c 2500 $OUT/bench.json; echo; tail -2 $OUT/bench.err
cd /tmp
D="python $R/bench.py --model migan-256 --dtype bf16 --steps 5 --warmup 2 --cpu-images 0 --streams 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/bf16_trace -o t --output-format csv -- $D > $R/$OUT/bf16_trace.log 2>&1; echo "bf16 trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$OUT/bf16_fetch -o fetch --output-format csv -- $D > $R/$OUT/bf16_fetch.log 2>&1; echo "bf16 fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$OUT/bf16_write -o write --output-format csv -- $D > $R/$OUT/bf16_write.log 2>&1; echo "bf16 write rc=$?"
E="python $R/bench.py --model migan-512 --dtype bf16 --steps 5 --warmup 2 --cpu-images 0 --streams 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/bf16_512_trace -o t --output-format csv -- $E > $R/$OUT/bf16_512_trace.log 2>&1; echo "bf16-512 trace rc=$?"
cd $R; du -sh $OUT
