set -u
OUT=gpurun_out/r3i; mkdir -p $OUT
B="python bench.py --steps 10 --warmup 3 --cpu-images 0 --no-secondary --no-latency"
for v in "" _ilp _memclause _trackers; do
  lib=$PWD/mi-gan_amd/csrc/libmigan_hip$v.so
  MIGAN_HIP_LIBRARY=$lib $B --dump-layers $OUT/layers$v.json > $OUT/bench$v.json 2>$OUT/bench$v.err
  python -c "import json; d=json.loads(open('$OUT/bench$v.json').read().strip().splitlines()[-1]); print('variant [$v]', d['value'], d['ms_per_step'], d['max_abs_vs_ref'], d['roofline']['whole_forward']['sum_kernel_ms'])"
done
export MIGAN_HIP_LIBRARY=$PWD/mi-gan_amd/csrc/libmigan_hip_ablate.so
for a in 0 16 32 48 4 8 64 128 192 1 240; do
  MIGAN_ABLATE=$a $B --steps 4 --warmup 2 --streams 1 --dump-layers $OUT/abl_$a.json > $OUT/abl_b_$a.json 2>/dev/null
done
python - <<'PY'
import json
base=json.load(open('gpurun_out/r3i/abl_0.json'))
wide=[i for i,L in enumerate(base) if 'wide' in L['kernel']]
print('layer', *[f'{base[i]["layer"][:18]:>18s}' for i in wide])
for a in (0,16,32,48,4,8,64,128,192,1,240):
    d=json.load(open(f'gpurun_out/r3i/abl_{a}.json'))
    print(f'abl{a:4d}', *[f'{d[i]["ms"]:18.4f}' for i in wide])
PY
