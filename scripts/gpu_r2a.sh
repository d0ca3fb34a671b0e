#!/bin/bash
# Round 2, GPU visit A: first contact of the restructured library with the hardware -- smoke, variant sweep, parity tests, bench.
set -u
OUT=gpurun_out/r2a
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.is_available(), torch.cuda.device_count(), torch.cuda.get_device_name(0))" > $OUT/env.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
timeout 600 python scripts/sweep.py --out $OUT/sweep512.json > $OUT/sweep512.log 2>&1; echo "sweep rc=$?"; cat $OUT/sweep512.log | tail -25
timeout 300 python scripts/sweep.py --model migan-256 --only base_s1,base_s2,bf16_s1,bf16_s2,f16_s2 --out $OUT/sweep256.json > $OUT/sweep256.log 2>&1; echo "sweep256 rc=$?"; tail -6 $OUT/sweep256.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -15 $OUT/pytest_gpu.log
timeout 600 python bench.py --dump-layers $OUT/layers.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/bench.err
head -c 1500 $OUT/bench.json; echo; tail -3 $OUT/bench.err
