#!/bin/bash
# One GPU-box visit: tests, smoke, bench, kernel trace.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.is_available(), torch.cuda.device_count(), torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
nproc >> gpurun_out/env.log; free -g | head -2 >> gpurun_out/env.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
cat gpurun_out/bench.json | head -c 3000; tail -3 gpurun_out/bench.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-images 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof | head -30
