#!/bin/bash
# SQ instruction / stall counters of every kernel of one migan-512 forward (whole-batch launches of `bench.py --pmc-pass`): five rocprofv3 --pmc
# passes of four counters each (no tracing domains besides --kernel-trace).  scripts/sq_counters.py turns the CSVs into profiles/r06_sq_counters.md.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-sq}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o pmc --output-format csv -- python $R/bench.py --pmc-pass 4 --cpu-images 0 --no-secondary --no-latency > $OUT/p$i.log 2>&1; echo "set $i rc=$?"
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/cc_$i.csv; rm -rf $OUT/p$i
done
cd $R; timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --dump-layers $OUT/per_launch.json > $OUT/bench.json 2> $OUT/bench.err; echo "layers rc=$?"
