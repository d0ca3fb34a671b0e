#!/bin/bash
# Same-box A/B of two library builds (MIGAN_HIP_LIBRARY): csrc/libmigan_hip_base.so (previous commit) vs csrc/libmigan_hip.so,
# alternating, per-launch tables.  -> gpurun_out/r2l/
set -u
OUT=gpurun_out/r2l
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export MIGAN_HIP_LIBRARY=$R/mi-gan_amd/csrc/libmigan_hip_base.so; else export MIGAN_HIP_LIBRARY=$R/mi-gan_amd/csrc/libmigan_hip.so; fi
    timeout 300 python scripts/sweep.py --steps 12 --layers --only base_s1,s2_stag10,bf16_s2 --out $OUT/s512_${lib}_$rep.json > $OUT/s512_${lib}_$rep.log 2>&1
    echo "== 512 $lib $rep"; grep -v amdgpu $OUT/s512_${lib}_$rep.log | grep "img/s"
    timeout 200 python scripts/sweep.py --steps 12 --layers --model migan-256 --only base_s1,bf16_s2 --out $OUT/s256_${lib}_$rep.json > $OUT/s256_${lib}_$rep.log 2>&1
    echo "== 256 $lib $rep"; grep -v amdgpu $OUT/s256_${lib}_$rep.log | grep "img/s"
  done
done
unset MIGAN_HIP_LIBRARY
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
