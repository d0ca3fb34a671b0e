#!/bin/bash
# Round-2 closing visit with the final plan (3-workgroup tiles for the Cin = 64 layers): every GPU test, smoke(), the default bench line,
# rocprofv3 kernel-trace stats of the three workloads and the HBM-traffic PMC passes of the primary one.  -> gpurun_out/r2r/
set -u
OUT=gpurun_out/r2r
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log
timeout 600 python bench.py --dump-layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err; tail -1 $OUT/bench.err
cd /tmp
B="python $R/bench.py --steps 5 --warmup 2 --cpu-images 0 --no-secondary"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace_s1 -o t --output-format csv -- $B --streams 1 > $R/$OUT/trace_s1.log 2>&1; echo "trace s1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace_s2 -o t --output-format csv -- $B --streams 2 > $R/$OUT/trace_s2.log 2>&1; echo "trace s2 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$OUT/pmc_fetch -o fetch --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-secondary --streams 1 > $R/$OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$OUT/pmc_write -o write --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-secondary --streams 1 > $R/$OUT/pmc_write.log 2>&1; echo "write rc=$?"
D="python $R/bench.py --model migan-256 --dtype bf16 --steps 5 --warmup 2 --cpu-images 0 --streams 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/bf16_trace -o t --output-format csv -- $D > $R/$OUT/bf16_trace.log 2>&1; echo "bf16 trace rc=$?"
E="python $R/bench.py --model migan-512 --dtype bf16 --steps 5 --warmup 2 --cpu-images 0 --streams 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/bf16_512_trace -o t --output-format csv -- $E > $R/$OUT/bf16_512_trace.log 2>&1; echo "bf16-512 trace rc=$?"
timeout 300 python $R/bench.py --model migan-512 --dtype bf16 --steps 20 --warmup 5 --cpu-images 2 > $R/$OUT/bench_bf16_512.json 2> $R/$OUT/bench_bf16_512.err; echo "bf16-512 bench rc=$?"
timeout 300 python $R/bench.py --model migan-256 --steps 20 --warmup 5 --cpu-images 2 > $R/$OUT/bench_f32_256.json 2> $R/$OUT/bench_f32_256.err; echo "f32-256 bench rc=$?"
C="python $R/bench.py --model comodgan-512 --steps 3 --warmup 2 --cpu-images 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/cm_trace -o t --output-format csv -- $C > $R/$OUT/cm_trace.log 2>&1; echo "cm trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$OUT/cm_fetch -o fetch --output-format csv -- $C > $R/$OUT/cm_fetch.log 2>&1; echo "cm fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$OUT/cm_write -o write --output-format csv -- $C > $R/$OUT/cm_write.log 2>&1; echo "cm write rc=$?"
timeout 300 python $R/bench.py --model comodgan-512 --steps 10 --warmup 3 --cpu-images 0 --dump-layers $R/$OUT/cm_layers.json > $R/$OUT/cm_bench.json 2> $R/$OUT/cm_bench.err; echo "cm bench rc=$?"
cd $R; du -sh $OUT
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2r/bench.json').read().strip().splitlines()[-1])
print('primary', d['value'], d['ms_per_step'], 'parity', d['max_abs_vs_ref'], 'exact', d.get('value_exact_f32'), 'roof', d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['traffic'])
for s in d.get('secondary', []):
    print('  sec', s.get('metric'), s.get('value'), s.get('ms_per_step'), s.get('max_abs_vs_ref'), s.get('error'), (s.get('roofline') or {}).get('frac'))
for f in ('bench_bf16_512','bench_f32_256'):
    d=json.loads(open(f'gpurun_out/r2r/{f}.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['max_abs_vs_ref'], d.get('max_abs_vs_fp32_ref'), d.get('storage_mode_envelope'))
PY
