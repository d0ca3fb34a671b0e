set -u
OUT=gpurun_out/r3n; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --cpu-images 2 --no-secondary --no-latency"
for t in 2 3; do
  $B --streams 1 --tune wide=$t --dump-layers $OUT/layers_wide_$t.json > $OUT/bench_wide_$t.json 2>$OUT/err_$t.txt
  python -c "import json; d=json.loads(open('$OUT/bench_wide_$t.json').read().strip().splitlines()[-1]); print('wide=$t s1', d['value'], d['ms_per_step'], d['max_abs_vs_ref'], d['roofline']['whole_forward']['sum_kernel_ms'])"
  $B --streams 2 --tune wide=$t > $OUT/bench2_wide_$t.json 2>>$OUT/err_$t.txt
  python -c "import json; d=json.loads(open('$OUT/bench2_wide_$t.json').read().strip().splitlines()[-1]); print('wide=$t s2', d['value'], d['ms_per_step'], d['max_abs_vs_ref'])"
done
python - <<'PY'
import json
a=[json.load(open(f'gpurun_out/r3n/layers_wide_{t}.json')) for t in (2,3)]
for i,L in enumerate(a[0]):
    if 'wide' in L['kernel']:
        print(f"{L['layer']:26s} {L['ms']:.4f} {a[1][i]['ms']:.4f}  {a[1][i]['ms']/L['ms']:.3f}")
PY
