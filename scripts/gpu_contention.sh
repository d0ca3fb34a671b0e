#!/bin/bash
# What does a collective's kernel cost the forward when it overlaps it (bench.py --gpus N gathers step i under step i+1), and do reserved CUs
# or finer persistent grids help?  The build box has one GPU: the collective's kernel is played by scripts/ubench/occupy.hip
# (bench.py --occupy WGS,US: WGS workgroups of 256 threads x 128 VGPRs that hold their CUs for US microseconds, launched behind every step).
# Output: gpurun_out/cont/*.json, one bench line each; results of round 5 in profiles/r05_contention.md.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/cont; mkdir -p $OUT
cd $R
b() { tag=$1; shift; timeout 200 python bench.py --steps 20 --warmup 5 --cpu-images 0 --no-secondary --no-latency "$@" > $OUT/$tag.json 2> $OUT/$tag.err; echo "$tag rc=$? $(python -c "import json,sys; d=json.loads(open('$OUT/$tag.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"; }
b base
b occ32 --occupy 32,2500
b occ64 --occupy 64,2500
b occ64x512 --occupy 64,2500,512
b occ32_5ms --occupy 32,5000
b base_s1 --streams 1
b grid224 --tune pipe_grid=224 --tune persist_grid=448
b res32_s1 --reserve-cus 32
b res32_s1_occ32 --reserve-cus 32 --occupy 32,2500
for g in 512 1024 2048; do b g$g --tune pipe_grid=$g; b g${g}_occ32 --tune pipe_grid=$g --occupy 32,2500; done
b base2
