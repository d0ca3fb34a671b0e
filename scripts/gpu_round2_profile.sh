#!/bin/bash
# Round 2 profiling visit: rocprofv3 kernel-trace stats of the bench command (one and two streams), PMC passes for HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate runs, kernel-trace only), the same for comodgan-512 and migan-256 bf16.  -> gpurun_out/r2prof/
set -u
OUT=gpurun_out/r2prof
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
B="python $R/bench.py --steps 5 --warmup 2 --cpu-images 0 --no-secondary"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace_s1 -o t --output-format csv -- $B --streams 1 > $R/$OUT/trace_s1.log 2>&1; echo "trace s1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace_s2 -o t --output-format csv -- $B --streams 2 > $R/$OUT/trace_s2.log 2>&1; echo "trace s2 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$OUT/pmc_fetch -o fetch --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-secondary --streams 1 > $R/$OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$OUT/pmc_write -o write --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-secondary --streams 1 > $R/$OUT/pmc_write.log 2>&1; echo "write rc=$?"
C="python $R/bench.py --model comodgan-512 --steps 3 --warmup 2 --cpu-images 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/cm_trace -o t --output-format csv -- $C > $R/$OUT/cm_trace.log 2>&1; echo "cm trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$OUT/cm_fetch -o fetch --output-format csv -- $C > $R/$OUT/cm_fetch.log 2>&1; echo "cm fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$OUT/cm_write -o write --output-format csv -- $C > $R/$OUT/cm_write.log 2>&1; echo "cm write rc=$?"
D="python $R/bench.py --model migan-256 --dtype bf16 --steps 5 --warmup 2 --cpu-images 0 --streams 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/bf16_trace -o t --output-format csv -- $D > $R/$OUT/bf16_trace.log 2>&1; echo "bf16 trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$OUT/bf16_fetch -o fetch --output-format csv -- $D > $R/$OUT/bf16_fetch.log 2>&1; echo "bf16 fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$OUT/bf16_write -o write --output-format csv -- $D > $R/$OUT/bf16_write.log 2>&1; echo "bf16 write rc=$?"
cd $R; find $OUT -name "*.csv" | head -40; du -sh $OUT
tail -2 $OUT/trace_s1.log
