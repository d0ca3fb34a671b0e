#!/bin/bash
# Stall / pipeline counters of bench.py in separate PMC passes (kernel-trace only).
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/stalls; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
pass() { n=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$n -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-images 0 > $OUT/$n.log 2>&1; echo "$n rc=$?"; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM
pass b SQ_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
pass c SQ_WAVES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
# (TA_* / TCC_* passes hung until the timeout on this pool and are not collected)
