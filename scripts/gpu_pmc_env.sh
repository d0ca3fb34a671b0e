#!/bin/bash
# HBM traffic counters (separate FETCH_SIZE / WRITE_SIZE passes) of bench.py under one env setting:
#   gpu_pmc_env.sh TAG VAR=value
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  env "$@" timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/$TAG/$c -o pmc --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-images 0 > $R/gpurun_out/$TAG/$c.log 2>&1
  echo "$c rc=$?"
done
