#!/bin/bash
# A/B experiments of one GPU visit; results under gpurun_out/exp/.
set -u
mkdir -p gpurun_out/exp
export TMPDIR=/tmp
run_bench () {  # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py --steps 8 --warmup 3 --cpu-images 0 --dump-layers gpurun_out/exp/layers_$tag.json > gpurun_out/exp/bench_$tag.json 2> gpurun_out/exp/bench_$tag.err
  python -c "import json,sys; d=json.load(open('gpurun_out/exp/bench_$tag.json')); print('$tag', d['value'], 'img/s', d['ms_per_step'], 'ms/step', 'dominant', d['roofline']['kernel'], d['roofline']['frac'])" || tail -3 gpurun_out/exp/bench_$tag.err
}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/exp/pytest_gpu.log 2>&1; tail -3 gpurun_out/exp/pytest_gpu.log
run_bench default A=1
run_bench nt64wgs2 MIGAN_NT64_WGS=2
run_bench singleb MIGAN_SINGLE_B=1
MIGAN_HIP_LIBRARY=$PWD/mi-gan_amd/csrc/libmigan_hip_prof.so timeout 600 python scripts/phase_profile.py 512 32 > gpurun_out/exp/phase_512.txt 2>&1; cat gpurun_out/exp/phase_512.txt | cut -c1-150
python - <<'PY' > gpurun_out/exp/cpu_threads.txt 2>&1
import importlib, sys, time, os
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("mi-gan_amd")
from oracle import migan_torch_cpu as torc
sd = pkg.synth.make_state_dict(512, seed=0); x = pkg.synth.make_input(1, 512, seed=0)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    torc.generator(x, sd, 512)
    t0 = time.perf_counter(); torc.generator(x, sd, 512); dt = time.perf_counter() - t0
    print("threads", nt, "s/img", round(dt, 3), flush=True)
PY
cat gpurun_out/exp/cpu_threads.txt
