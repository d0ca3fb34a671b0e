#!/bin/bash
# SQ counters of the Co-Mod-GAN forward (separate PMC passes, kernel-trace only) -> gpurun_out/comodgan_pmc_TAG/
set -u
TAG=${1:-run}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/comodgan_pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
pass() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o pmc --output-format csv -- python $R/scripts/bench_comodgan.py --steps 1 --warmup 1 --cpu-images 0 > $OUT/$n.log 2>&1; echo "$n rc=$?"; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass b SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM
pass c SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS
python $R/scripts/pmc_kernels.py $OUT/a/pmc_counter_collection.csv $OUT/b/pmc_counter_collection.csv $OUT/c/pmc_counter_collection.csv > $OUT/table.txt 2>&1
tail -30 $OUT/table.txt
