#!/bin/bash
# round 5, visit m: full GPU suite, default bench, W2 threshold / stream sweep, PMC passes of the three single-GPU configs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5m; mkdir -p $OUT
export TMPDIR=/tmp
cd $R; timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
L() { tag=$1; shift; timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 10 --warmup 3 --dump-layers $OUT/layers_$tag.json "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "$tag rc=$?"; }
L base
L t128 --tune w2_min_tiles=128
L t512 --tune w2_min_tiles=512
L s1 --streams 1
L s3 --streams 3
L w2off --tune w2=0
L base2
# PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs): the launches the roofline table times
pmc() { tag=$1; shift; cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${c}_$tag -o pmc --output-format csv -- python $R/bench.py --pmc-pass 4 --cpu-images 0 --no-secondary --no-latency "$@" > $OUT/pmc_${c}_$tag.log 2>&1; echo "pmc $tag $c rc=$?"; done
  cd $R; python scripts/pmc_traffic.py $(find $OUT/pmc_FETCH_SIZE_$tag -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_WRITE_SIZE_$tag -name '*counter_collection.csv' | head -1) > $OUT/pmc_traffic_$tag.json; echo "traffic table $tag: $(wc -c < $OUT/pmc_traffic_$tag.json) bytes"; rm -rf $OUT/pmc_FETCH_SIZE_$tag $OUT/pmc_WRITE_SIZE_$tag; }
pmc migan512
pmc migan256_bf16 --model migan-256 --dtype bf16
pmc comodgan512 --model comodgan-512
