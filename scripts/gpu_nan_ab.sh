#!/bin/bash
# A/B of the NaN policies on one box: default build (clamp4 / clamp1: pairwise v_cmp_u_f32, repair block out of line), the same with the
# repair block inline (-DMIGAN_NAN_NOEXPECT), the opt-in -DMIGAN_NAN_CLAMP build (bare v_med3_f32).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-nanab}; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
B="python bench.py --no-secondary --cpu-images 0 --no-latency --steps 30 --warmup 8"
for rep in 1 2 3; do
  for v in default nanclamp; do
    if [ $v = default ]; then unset MIGAN_HIP_LIBRARY; else export MIGAN_HIP_LIBRARY=$R/mi-gan_amd/csrc/libmigan_hip_$v.so; fi
    timeout 300 $B > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
    python - <<PY
import json
l=[x for x in open("$OUT/bench_${v}_$rep.json") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$v", $rep, d.get("value"), d.get("ms_per_step"), d.get("roofline",{}).get("whole_forward",{}).get("sum_kernel_ms"))
PY
  done
done
