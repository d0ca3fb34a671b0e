#!/bin/bash
# round 5, visit p: persistent grids that leave a few CUs to the other sub-batch's small launches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5p; mkdir -p $OUT
B() { tag=$1; shift; timeout 300 python bench.py --no-secondary --cpu-images 0 --no-latency --steps 20 --warmup 5 "$@" 2> $OUT/bench_$tag.err | tail -1 > $OUT/bench_$tag.json; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
b=json.loads(open(f'gpurun_out/r5p/bench_{t}.json').read())
print('%-22s %.0f img/s  %.3f ms/step' % (t, b['value'], b['ms_per_step']))
PY
}
B base
B g248 --tune pipe_grid=248
B g240 --tune pipe_grid=240
B g224 --tune pipe_grid=224
B g248_s3 --tune pipe_grid=248 --streams 3
B g248_s4 --tune pipe_grid=248 --streams 4
B g240_s4 --tune pipe_grid=240 --streams 4
B s4 --streams 4
B g248_st50 --tune pipe_grid=248 --tune stagger_pct=50
B g248_st10 --tune pipe_grid=248 --tune stagger_pct=10
B t128 --tune w2_min_tiles=128
B base2
