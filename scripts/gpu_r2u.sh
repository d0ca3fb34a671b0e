#!/bin/bash
# Co-Mod-GAN transposed convolution: four-phase launch on 128-column tiles (1 wave per SIMD) vs on 64-column tiles (now 2 waves per SIMD)
# vs four single-phase launches.  -> gpurun_out/r2u/
set -u
OUT=gpurun_out/r2u
mkdir -p $OUT
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --model comodgan-512 --steps 10 --warmup 3 --cpu-images 2 --dump-layers $OUT/layers_$tag.json > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; python - "$tag" <<'PY'
import json, sys, collections
tag=sys.argv[1]
d=json.loads(open(f'gpurun_out/r2u/bench_{tag}.json').read().strip().splitlines()[-1])
L=json.load(open(f'gpurun_out/r2u/layers_{tag}.json'))
ph=[l for l in L if 'conv0' in l['layer'] and 'fir' not in l['layer'] and 'style' not in l['layer'] and l['layer'].startswith('synthesis')]
print(tag, d['value'], d['ms_per_step'], d['max_abs_vs_ref'], 'conv0 total ms', round(sum(l['ms'] for l in ph),3))
for l in ph: print('    ', l['layer'], round(l['ms'],3), l['kernel'][22:])
PY
}
run default A=1
run up4all COMODGAN_UP4=1
run up4nt64 COMODGAN_UP4=1 COMODGAN_UP4_NT=64
run phases COMODGAN_UP4=0
