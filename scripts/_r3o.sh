set -u
OUT=gpurun_out/r3o; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --cpu-images 2 --no-secondary --no-latency --tune wide=3"
for v in "" _who1 _who2; do
  export MIGAN_HIP_LIBRARY=$PWD/mi-gan_amd/csrc/libmigan_hip$v.so
  $B --streams 1 --dump-layers $OUT/layers$v.json > $OUT/bench$v.json 2>$OUT/err$v.txt
  python -c "import json; d=json.loads(open('$OUT/bench$v.json').read().strip().splitlines()[-1]); print('who [$v] s1', d['value'], d['ms_per_step'], d['max_abs_vs_ref'], d['roofline']['whole_forward']['sum_kernel_ms'])"
done
python - <<'PY'
import json
a=[json.load(open(f'gpurun_out/r3o/layers{v}.json')) for v in ('','_who1','_who2')]
for i,L in enumerate(a[0]):
    if 'wide' in L['kernel']:
        print(f"{L['layer']:26s} {L['ms']:.4f} {a[1][i]['ms']:.4f} {a[2][i]['ms']:.4f}")
PY
