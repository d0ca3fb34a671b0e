#!/bin/bash
# Co-Mod-GAN after the streaming-kernel work (one style launch, FromRGB / ToRGB / FIR kernels): GPU tests, bench with per-launch dump.
set -u
OUT=gpurun_out/r2t
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_comodgan.py -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python bench.py --model comodgan-512 --steps 10 --warmup 3 --cpu-images 4 --dump-layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json, collections
d=json.loads(open('gpurun_out/r2t/bench.json').read().strip().splitlines()[-1])
print('comodgan', d['value'], d['ms_per_step'], d['max_abs_vs_ref'], d.get('value_noise_random',{}).get('value'))
L=json.load(open('gpurun_out/r2t/layers.json'))
agg=collections.OrderedDict()
for l in L:
    a=agg.setdefault(l['kernel'],[0,0]); a[0]+=l['ms']; a[1]+=1
for k,(ms,n) in sorted(agg.items(), key=lambda x:-x[1][0]): print(f"{ms:7.3f} ms n={n:2d} {k}")
PY
