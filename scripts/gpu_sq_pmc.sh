#!/bin/bash
# SQ counters of the migan-512 forward (separate PMC passes, kernel-trace only) -> gpurun_out/<tag>/sq_{a,b,c}; usage: gpu_sq_pmc.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp
pass() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/sq_$n -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-secondary --no-latency --streams 1 $ARGS > $OUT/sq_$n.log 2>&1; echo "sq_$n rc=$?"; }
ARGS="$*"
pass a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS
pass b SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES
pass c SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
cd $R
python bench.py --no-secondary --cpu-images 0 --no-latency --steps 5 --streams 1 --dump-layers $OUT/sq_layers.json $ARGS > /dev/null 2> $OUT/sq_layers.err
f=$(find $OUT/sq_a -name '*counter_collection.csv' | head -1); cp $f $OUT/sq_all.csv
for n in b c; do f=$(find $OUT/sq_$n -name '*counter_collection.csv' | head -1); tail -n +2 $f >> $OUT/sq_all.csv; done
python scripts/pmc_table.py $OUT/sq_all.csv $OUT/sq_layers.json > $OUT/sq_table.txt 2>&1; tail -30 $OUT/sq_table.txt
