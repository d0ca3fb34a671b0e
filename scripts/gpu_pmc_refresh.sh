#!/bin/bash
# Refresh the three PMC traffic tables (profiles/pmc_traffic*_latest.json are stamped with a digest of csrc/: any source change makes bench.py
# report traffic_stale) and print a fresh bench line.  Usage: gpurun -- 'bash scripts/gpu_pmc_refresh.sh'; then copy gpurun_out/pmcr/*.json.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmcr; mkdir -p $OUT
export TMPDIR=/tmp
pmc() { tag=$1; shift; cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${c}_$tag -o pmc --output-format csv -- python $R/bench.py --pmc-pass 4 --cpu-images 0 --no-secondary --no-latency "$@" > $OUT/pmc_${c}_$tag.log 2>&1; echo "pmc $tag $c rc=$?"; done
  cd $R; python scripts/pmc_traffic.py $(find $OUT/pmc_FETCH_SIZE_$tag -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_WRITE_SIZE_$tag -name '*counter_collection.csv' | head -1) > $OUT/pmc_traffic_$tag.json; rm -rf $OUT/pmc_FETCH_SIZE_$tag $OUT/pmc_WRITE_SIZE_$tag; }
pmc migan512
pmc migan256_bf16 --model migan-256 --dtype bf16
pmc comodgan512 --model comodgan-512
cd $R
cp $OUT/pmc_traffic_migan512.json profiles/pmc_traffic_latest.json
cp $OUT/pmc_traffic_migan256_bf16.json profiles/pmc_traffic_migan256_bf16_latest.json
cp $OUT/pmc_traffic_comodgan512.json profiles/pmc_traffic_comodgan_latest.json
timeout 800 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
