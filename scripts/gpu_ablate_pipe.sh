#!/bin/bash
# Stage ablation of the pipelined kernels (measurement build -DMIGAN_ABLATE): ms of the 512x512 layers per ablation mask.
# bits: 1 no epilogue stores, 2 no epilogue, 4 no depthwise stage, 8 no MFMA, 16 no input DMA / tile build
set -u
OUT=gpurun_out/${1:-ablp}; mkdir -p $OUT
export MIGAN_HIP_LIBRARY=$PWD/mi-gan_amd/csrc/libmigan_hip_ablate.so
for a in 0 1 2 4 8 16 20 28 30 31; do
  MIGAN_ABLATE=$a timeout 300 python bench.py --steps 5 --warmup 2 --cpu-images 0 --no-secondary --no-latency --streams 1 --dump-layers $OUT/layers_$a.json > $OUT/b_$a.json 2>/dev/null
done
python - <<PY
import json
masks=[0,1,2,4,8,16,20,28,30,31]
rows={}
for a in masks:
    try:
        for l in json.load(open("$OUT/layers_%d.json"%a)):
            rows.setdefault(l["layer"],{})[a]=l["ms"]
    except Exception as e: print("mask",a,"failed",e)
print("%-28s"%"layer"+"".join("%8d"%a for a in masks))
for k,v in rows.items():
    if "b512" in k or "b256" in k: print("%-28s"%k+"".join("%8.3f"%v.get(a,float('nan')) for a in masks))
PY
