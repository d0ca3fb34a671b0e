#!/bin/bash
# Round 2: 64-channel main tiles at 3 workgroups per CU (tuning key w3) vs the default, GPU tests of those variants, and the default
# bench line with its new legs (uint8 I/O secondary, Co-Mod-GAN noise_mode=random).  -> gpurun_out/r2m/
set -u
OUT=gpurun_out/r2m
mkdir -p $OUT
timeout 600 python scripts/sweep.py --steps 12 --layers --only base_s1,w3_plain_s1,w3_rgb_s1,w3_up_s1,w3_all_s1,base_s2,s2_stag10,w3_plain_s2,w3_all_s2 --out $OUT/sweep512.json > $OUT/sweep512.log 2>&1; grep -v amdgpu $OUT/sweep512.log | grep "img/s\|ERROR"
timeout 600 python -m pytest tests/test_gpu_round2.py -q -k "three_workgroup" > $OUT/pytest_w3.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_w3.log
timeout 600 python bench.py --dump-layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2m/bench.json').read().strip().splitlines()[-1])
print('primary', d['value'], d['ms_per_step'], 'exact', d.get('value_exact_f32'))
for s in d.get('secondary', []):
    print('  sec', s.get('metric'), s.get('value'), s.get('ms_per_step'), s.get('max_abs_vs_ref'), s.get('error'), s.get('value_noise_random'), (s.get('roofline') or {}).get('traffic'))
PY
