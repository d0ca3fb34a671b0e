#!/bin/bash
run() { python bench.py --steps 20 --warmup 5 --cpu-images 0 --no-secondary --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %8.1f img/s %7.3f ms' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step']))" "$@"; }
run --tune pipe=0
run --tune pipe=7 --tune pipe_na8=3
run --tune pipe=7 --tune pipe_na8=3
run --tune pipe=5 --tune pipe_na8=1
run --tune pipe=7 --tune pipe_na8=7
run --tune pipe=7 --tune pipe_na8=3 --streams 1
run --tune pipe=7 --tune pipe_na8=3 --streams 3
run --tune pipe=7 --tune pipe_na8=3 --streams 4
run --tune pipe=7 --tune pipe_na8=3 --tune stagger_pct=10
run --tune pipe=7 --tune pipe_na8=3 --tune stagger_pct=35
run --tune pipe=7 --tune pipe_na8=3 --tune stagger_pct=50
run --tune pipe=7 --tune pipe_na8=3 --tune pipe_grid=512
run --tune pipe=0 --streams 1
