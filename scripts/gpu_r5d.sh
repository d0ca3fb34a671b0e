#!/bin/bash
# round 5, visit d: phase profile of sepconv_wide2_kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5d; mkdir -p $OUT
MIGAN_HIP_LIBRARY=$R/mi-gan_amd/csrc/libmigan_hip_prof.so timeout 300 python scripts/phase_profile.py 512 32 > $OUT/phase.txt 2> $OUT/phase.err; echo "phase rc=$?"
grep -E "wide2|layer" $OUT/phase.txt
