#!/bin/bash
# quick visit: gpu tests, bench with per-layer dump, one PMC pass (instruction counts)
set -u
TAG=${1:-quick}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.log 2>&1; tail -2 gpurun_out/$TAG/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-images 0 --dump-layers gpurun_out/$TAG/layers.json > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python -c "import json; d=json.load(open('gpurun_out/$TAG/bench.json')); print('$TAG', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['roofline']['kernel'], d['roofline']['frac'])" || tail -3 gpurun_out/$TAG/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc -o sq1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-images 0 > $GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc.log 2>&1
echo "pmc rc=$?"
