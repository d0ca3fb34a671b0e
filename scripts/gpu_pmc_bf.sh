set -u
mkdir -p gpurun_out/gemm
export TMPDIR=/tmp MIGAN_GEMM=bf16x3
python bench.py --steps 5 --warmup 2 --cpu-images 0 --dump-layers gpurun_out/gemm/layers_pmc.json > gpurun_out/gemm/b.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $GRAFT_REPO_ROOT/gpurun_out/gemm/pmc -o sq1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-images 0 > $GRAFT_REPO_ROOT/gpurun_out/gemm/pmc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/gemm/pmc2 -o sq2 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-images 0 > $GRAFT_REPO_ROOT/gpurun_out/gemm/pmc2.log 2>&1
echo done
