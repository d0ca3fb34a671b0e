#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/sweep6; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
B="python bench.py --no-secondary --cpu-images 0 --no-latency --steps 30 --warmup 8"
run() { tag=$1; shift; timeout 300 $B "$@" > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
l=[x for x in open("$OUT/bench_$tag.json") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$tag", d.get("value"), d.get("ms_per_step"), d.get("roofline",{}).get("whole_forward",{}).get("sum_kernel_ms"))
PY
}
run base1
run na8_11 --tune pipe_na8=11
run na8_0 --tune pipe_na8=0
run na8_8 --tune pipe_na8=8
run na8_1 --tune pipe_na8=1
run dna8 --tune pipe_dna=8
run dna4 --tune pipe_dna=4
run base2
run stag10 --tune stagger_pct=10
run stag20 --tune stagger_pct=20
run stag25 --tune stagger_pct=25
run w2t128 --tune w2_min_tiles=128
run w2t512 --tune w2_min_tiles=512
run streams3 --streams 3
run base3
