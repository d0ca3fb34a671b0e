#!/bin/bash
set -u
OUT=gpurun_out/r2h
mkdir -p $OUT
MIGAN_HIP_LIBRARY=$PWD/mi-gan_amd/csrc/libmigan_hip_prof.so timeout 300 python scripts/phase_profile.py 512 32 2>&1 | grep -v amdgpu | tee $OUT/phase512.txt
