#!/usr/bin/env python3
"""Per-layer phase breakdown (debug build libmigan_hip_prof.so, -DMIGAN_PHASE_PROF): cycles spent by
thread 0 of every workgroup in [prologue, S1 LDS fill (+barrier waits), S2 depthwise, S3 MFMA,
accumulators->LDS, epilogue], averaged per workgroup, next to the hipEvent duration of the launch.
    MIGAN_HIP_LIBRARY=mi-gan_amd/csrc/libmigan_hip_prof.so python scripts/phase_profile.py [res] [batch]
"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

pkg = importlib.import_module("mi-gan_amd")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for kv in sys.argv[3:]:
    k, _, v = kv.partition("=")
    pkg.load_library().set_tuning(k, int(v))
model = pkg.Generator(R); model.set_streams(1)
model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pkg.synth.make_state_dict(R, seed=0).items()})
model = model.to("cuda").eval()
x = torch.from_numpy(pkg.synth.make_input(B, R, seed=1)).to("cuda")
with torch.no_grad():
    for _ in range(2):
        model(x)
    _, ms = model.forward_timed(x)
    _, ms = model.forward_timed(x)
lib = model._lib.lib
lib.migan_prof_layer.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
launches = model.launch_info()
names = ["prolog", "S1fill", "S2dw", "S3mfma", "acc2lds", "epilog"]
print(f"{'layer':26s} {'ms':>8s} {'WGs':>7s} " + " ".join(f"{n:>8s}" for n in names) + "   total cyc/WG")
for i, (L, t) in enumerate(zip(launches, ms)):
    out = (C.c_ulonglong * 16)()
    if lib.migan_prof_layer(i, out) != 0:
        continue
    n = max(1, out[8])
    if "wide2" in L["kernel"]:
        cyc = [out[k] / n / 1000.0 for k in range(16)]
        print(f"{L['layer']:26s} {t:8.3f} {out[8]:7d}  wide2 kcycles/WG A[issue in {cyc[0]:.1f} taps {cyc[9]:.1f} planes {cyc[10]:.1f} other {cyc[14]:.1f} depthwise {cyc[1]:.1f} vmcnt {cyc[2]:.1f} barrier {cyc[3]:.1f}]"
              f" B[mfma {cyc[4]:.1f} epilogue {cyc[5]:.1f} barrier {cyc[6]:.1f}]")
        continue
    if "pipe" in L["kernel"]:
        cyc = [out[k] / n / 1000.0 for k in range(16)]
        print(f"{L['layer']:26s} {t:8.3f} {out[8]:7d}  pipe kcycles/WG A[issue {cyc[0]:.0f} depthwise {cyc[1]:.0f} vmcnt {cyc[2]:.0f} barrier {cyc[3]:.0f}]"
              f" B[mfma {cyc[4]:.0f} noise-wait {cyc[5]:.0f} block-epi {cyc[9]:.0f} rgb-part {cyc[10]:.0f} rgb-fin {cyc[11]:.0f} request {cyc[12]:.0f} build {cyc[13]:.0f} rest {cyc[7]:.0f} barrier {cyc[6]:.0f}] A-stage2 {cyc[14]:.0f}")
        continue
    if "wide" in L["kernel"]:
        cyc = [out[k] / n for k in range(8)]
        print(f"{L['layer']:26s} {t:8.3f} {out[8]:7d}  wide A[w->lds, in->lds+loads, dw (wide=2; else store/wait, dw, mfma), barrier] " + " ".join(f"{c:8.0f}" for c in cyc[:4])
              + "   B: " + " ".join(f"{c:8.0f}" for c in cyc[4:]))
        continue
    cyc = [out[k] / n for k in range(6)]
    print(f"{L['layer']:26s} {t:8.3f} {out[8]:7d} " + " ".join(f"{c:8.0f}" for c in cyc) + f"   {sum(cyc):9.0f}")
