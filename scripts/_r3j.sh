set -u
OUT=gpurun_out/r3j; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --cpu-images 2 --no-secondary --no-latency --streams 1"
for t in 0 1 2; do
  $B --tune mt64=$t --dump-layers $OUT/layers_mt64_$t.json > $OUT/bench_mt64_$t.json 2>$OUT/err_$t.txt
  python -c "import json; d=json.loads(open('$OUT/bench_mt64_$t.json').read().strip().splitlines()[-1]); print('mt64=$t', d['value'], d['ms_per_step'], d['max_abs_vs_ref'], d['roofline']['whole_forward']['sum_kernel_ms'])"
done
python - <<'PY'
import json
a=[json.load(open(f'gpurun_out/r3j/layers_mt64_{t}.json')) for t in (0,1,2)]
for i,L in enumerate(a[0]):
    if abs(a[1][i]['ms']-L['ms'])>0.004 or abs(a[2][i]['ms']-L['ms'])>0.004:
        print(f"{L['layer']:26s} {L['ms']:.4f} {a[1][i]['ms']:.4f} {a[2][i]['ms']:.4f}   {a[1][i]['kernel'][:60]} | {a[2][i]['kernel'][:60]}")
PY
