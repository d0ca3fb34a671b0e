#!/bin/bash
# round 6, final visit: the whole GPU suite, smoke, PMC traffic tables of the three single-GPU BASELINE configs (stamped with the digest of csrc/),
# the driver's bench command (after the tables, so that its line carries fresh traffic), per-launch tables, rocprofv3 kernel stats.
# Everything lands in gpurun_out/r6z/ (copied to profiles/r06_* afterwards).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r6z}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -4 $OUT/smoke.log
pmc() { tag=$1; shift; cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${c}_$tag -o pmc --output-format csv -- python $R/bench.py --pmc-pass 4 --cpu-images 0 --no-secondary --no-latency "$@" > $OUT/pmc_${c}_$tag.log 2>&1; echo "pmc $tag $c rc=$?"; done
  cd $R; python scripts/pmc_traffic.py $(find $OUT/pmc_FETCH_SIZE_$tag -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_WRITE_SIZE_$tag -name '*counter_collection.csv' | head -1) 6 > $OUT/pmc_traffic_$tag.json; echo "traffic table $tag: $(wc -c < $OUT/pmc_traffic_$tag.json) bytes"; rm -rf $OUT/pmc_FETCH_SIZE_$tag $OUT/pmc_WRITE_SIZE_$tag; }
pmc migan512
pmc migan256_bf16 --model migan-256 --dtype bf16
pmc comodgan512 --model comodgan-512
cd $R
[ -s $OUT/pmc_traffic_migan512.json ] && cp $OUT/pmc_traffic_migan512.json profiles/pmc_traffic_latest.json
[ -s $OUT/pmc_traffic_migan256_bf16.json ] && cp $OUT/pmc_traffic_migan256_bf16.json profiles/pmc_traffic_migan256_bf16_latest.json
[ -s $OUT/pmc_traffic_comodgan512.json ] && cp $OUT/pmc_traffic_comodgan512.json profiles/pmc_traffic_comodgan_latest.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 600 $OUT/bench.json
timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --dump-layers $OUT/per_launch.json > $OUT/per_launch_bench.json 2> $OUT/per_launch.err; echo "layers rc=$?"
timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --batch 1 --streams 1 --dump-layers $OUT/per_launch_batch1.json > $OUT/per_launch_batch1_bench.json 2>> $OUT/per_launch.err; echo "layers b1 rc=$?"
timeout 300 python bench.py --model migan-256 --dtype bf16 --no-secondary --cpu-images 0 --no-latency --dump-layers $OUT/per_launch_migan256_bf16.json > $OUT/per_launch_migan256_bf16_bench.json 2>> $OUT/per_launch.err; echo "layers bf16 rc=$?"
cd /tmp
tr() { tag=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$tag -o t --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --cpu-images 0 --no-secondary --no-latency "$@" > $OUT/trace_$tag.log 2>&1; echo "trace $tag rc=$?"; cp $(find $OUT/trace_$tag -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_$tag.csv; rm -rf $OUT/trace_$tag; }
tr migan512_f32_streams2
tr migan512_f32_streams1 --streams 1
tr migan512_f32_nanclamp --nan-policy clamp
tr migan256_bf16 --model migan-256 --dtype bf16
tr comodgan512 --model comodgan-512
cd $R
for rep in 1 2; do for pol in propagate clamp; do
  timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 30 --warmup 8 --nan-policy $pol > $OUT/nan_${pol}_$rep.json 2>/dev/null
  python - <<PY
import json
l=[x for x in open("$OUT/nan_${pol}_$rep.json") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("nan policy $pol run $rep:", d.get("value"), "images/s", d.get("ms_per_step"), "ms")
PY
done; done
