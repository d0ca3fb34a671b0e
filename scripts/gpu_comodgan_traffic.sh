#!/bin/bash
# HBM traffic of the Co-Mod-GAN kernels: FETCH_SIZE and WRITE_SIZE in separate PMC passes (kernel-trace only), averaged per kernel.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/comodgan_traffic; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o pmc --output-format csv -- python $R/scripts/bench_comodgan.py --steps 1 --warmup 1 --cpu-images 0 > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o pmc --output-format csv -- python $R/scripts/bench_comodgan.py --steps 1 --warmup 1 --cpu-images 0 > $OUT/write.log 2>&1; echo "write rc=$?"
python $R/scripts/pmc_traffic.py $OUT/fetch/pmc_counter_collection.csv $OUT/write/pmc_counter_collection.csv > $OUT/traffic.json 2> $OUT/traffic.err; head -c 1500 $OUT/traffic.json
