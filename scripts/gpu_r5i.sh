#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5i; mkdir -p $OUT
for v in 1 9; do
MIGAN_HIP_LIBRARY=$R/mi-gan_amd/csrc/libmigan_hip_prof.so timeout 300 python scripts/phase_profile.py 512 32 w2=$v > $OUT/phase_$v.txt 2> $OUT/phase_$v.err; echo "phase rc=$?"
grep -E "wide2" $OUT/phase_$v.txt
done
