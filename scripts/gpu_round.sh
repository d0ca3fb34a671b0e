#!/bin/bash
# Full GPU-box visit: parity tests, smoke, bench (with CPU baseline), rocprofv3 kernel trace + stats,
# PMC passes for HBM traffic.  Everything lands in gpurun_out/round/.
set -u
OUT=gpurun_out/round
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.is_available(), torch.cuda.device_count(), torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).gcnArchName)" > $OUT/env.log 2>&1
nproc >> $OUT/env.log
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
timeout 900 python bench.py --dump-layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
head -c 1200 $OUT/bench.json; echo; tail -2 $OUT/bench.err
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o trace --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --cpu-images 0 > $R/$OUT/trace.log 2>&1; echo "trace rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$OUT/pmc_fetch -o fetch --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-images 0 > $R/$OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$OUT/pmc_write -o write --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-images 0 > $R/$OUT/pmc_write.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $R/$OUT/pmc_sq -o sq1 --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-images 0 > $R/$OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
cd $R; ls $OUT/trace | head
