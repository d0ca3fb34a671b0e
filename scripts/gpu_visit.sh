#!/bin/bash
# One parametrised GPU visit (replaces the per-visit scripts of rounds 1-2).  usage: bash scripts/gpu_visit.sh <tag> <what>...
#   what: tests[=<pytest args>] | smoke | bench[=<bench args>] | layers[=<bench args>] | trace[=<bench args>] | pmc[=<bench args>] | cmd=<shell command>
# Everything lands in gpurun_out/<tag>/.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
n=0
for what in "$@"; do
  n=$((n+1)); kind=${what%%=*}; arg=""; [ "$kind" != "$what" ] && arg=${what#*=}
  case $kind in
    tests)  cd $R; timeout 1500 python -m pytest ${arg:-tests -m gpu} -q -x > $OUT/pytest_$n.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_$n.log; tail -6 $OUT/pytest_$n.log;;
    smoke)  cd $R; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -4 $OUT/smoke.log;;
    bench)  cd $R; timeout 900 python bench.py $arg > $OUT/bench_$n.json 2> $OUT/bench_$n.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench_$n.json; tail -3 $OUT/bench_$n.err;;
    layers) cd $R; timeout 600 python bench.py --no-secondary --cpu-images 0 --no-latency --dump-layers $OUT/layers_$n.json $arg > $OUT/layers_bench_$n.json 2> $OUT/layers_$n.err; echo "layers rc=$?";;
    trace)  cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$n -o t --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --cpu-images 0 --no-secondary --no-latency $arg > $OUT/trace_$n.log 2>&1; echo "trace rc=$?";;
    pmc)    cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${c}_$n -o pmc --output-format csv -- python $R/bench.py --pmc-pass 4 --cpu-images 0 --no-secondary --no-latency $arg > $OUT/pmc_${c}_$n.log 2>&1; echo "pmc $c rc=$?"; done
            cd $R; python scripts/pmc_traffic.py $(find $OUT/pmc_FETCH_SIZE_$n -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_WRITE_SIZE_$n -name '*counter_collection.csv' | head -1) 6 > $OUT/pmc_traffic_$n.json; echo "traffic table: $(wc -c < $OUT/pmc_traffic_$n.json) bytes";;
    cmd)    cd $R; bash -c "$arg" > $OUT/cmd_$n.log 2>&1; echo "cmd rc=$?"; tail -20 $OUT/cmd_$n.log;;
    *) echo "unknown step $what";;
  esac
done
