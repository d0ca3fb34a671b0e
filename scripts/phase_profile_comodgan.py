#!/usr/bin/env python3
"""Phase breakdown of cm_conv_kernel (debug build libmigan_hip_prof.so, -DMIGAN_PHASE_PROF): cycles spent by thread 0 of every
workgroup in [prologue, load issue, LDS reads + MFMAs, weight tile -> LDS (incl. wait for its loads), barrier, input tile -> LDS +
barrier, epilogue], averaged per workgroup, next to the hipEvent duration of the launch (nine-tap launches only).
    MIGAN_HIP_LIBRARY=mi-gan_amd/csrc/libmigan_hip_prof.so python scripts/phase_profile_comodgan.py [res] [batch]
"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

pkg = importlib.import_module("mi-gan_amd")
cs, cm = pkg.comodgan_schema, pkg.comodgan
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = cs.Config(resolution=R, num_ws=cs.default_num_ws(R))
sd = pkg.synth.make_comodgan_state_dict(cfg, 0)
model = cm.Generator(cm.Mapping(num_ws=cfg.num_ws), cm.Encoder(resolution=R), cm.Synthesis(resolution=R))
model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
model = model.to("cuda").eval()
x = torch.from_numpy(pkg.synth.make_input(B, R, seed=1)).to("cuda")
z = torch.from_numpy(pkg.synth.make_latent(B, 512, seed=1)).to("cuda")
with torch.no_grad():
    for _ in range(2):
        model(x, z=z, noise_mode="const")
    _, ms = model.forward_timed(x, z)
    _, ms = model.forward_timed(x, z)
lib = model._lib.lib
lib.migan_prof_layer.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
names = ["prolog", "ld_issue", "rd+mfma", "wt->lds", "barrier", "in->lds", "epilog"]
print(f"{'layer':28s} {'kernel':22s} {'ms':>7s} {'WGs':>7s} " + " ".join(f"{n:>9s}" for n in names) + "   total cyc/WG   mfma_cyc/wave")
for i, (L, t) in enumerate(zip(model.launch_info(), ms)):
    if "cm_conv_kernel" not in L["kernel"] or ", true, " not in L["kernel"]:
        continue
    out = (C.c_ulonglong * 16)()
    if lib.migan_prof_layer(i, out) != 0 or out[8] == 0:
        continue
    n = out[8]
    cyc = [out[k] / n for k in range(7)]
    nt = int(L["kernel"].split("<")[1].split(",")[0])
    mti = int(L["kernel"].rstrip(">").split(",")[-1])
    wgs_per_img = n / B
    mfma_per_wave = L["mfma_flops"] * B / n / 4 / 32768 * 3 * 32          # MFMA instructions per wave x 32 cycles
    print(f"{L['layer']:28s} {L['kernel'][22:]:22s} {t:7.3f} {n:7d} " + " ".join(f"{c:9.0f}" for c in cyc) + f"   {sum(cyc):10.0f}   {mfma_per_wave:10.0f}")
