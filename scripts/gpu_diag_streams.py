#!/usr/bin/env python
"""Which stored tensor differs between two-stream runs: keep-intermediates mode with the two-sub-batch execution left on
(tuning knob debug_split), the same forward several times, per-layer / per-sub-batch count of differing bytes against a
one-stream run.   usage: gpu_diag_streams.py RES BATCH DTYPE"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("mi-gan_amd")
res, batch, dtype = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda:0")
lib = pkg.load_library()
lib.set_tuning("debug_split", 1)
if len(sys.argv) > 4:
    lib.set_tuning("stagger", int(sys.argv[4]))
sd = pkg.synth.make_state_dict(res, seed=0)
x = torch.from_numpy(pkg.synth.make_input(batch, res, seed=100, kind="demo")).to(dev)
dsd = {k: torch.from_numpy(v.reshape(1) if v.ndim == 0 else v).to(dev) for k, v in sd.items()}
stream = int(torch.cuda.current_stream().cuda_stream)
h = pkg.hipbind.MiganHandle(lib, res, 0, dtype=dtype)
h.set_debug(True)
for name, shape, _ in h.weights():
    h.set_weight(name, dsd[name].data_ptr(), shape)
h.commit(stream)
n0 = (batch // 2 + 7) // 8 * 8
n1 = batch - n0
h.set_streams(1)
first = h.launches()[0]["layer"]
base0 = h.debug_tensor(n0, first)[0]
sub0_bytes = h.workspace_bytes(n0) - base0
esz = 4 if dtype == "f32" else 2


def regions():
    out = []
    for L in h.launches():
        name = L["layer"]
        if name.endswith(".dwfir") or name.endswith(".torgb"):
            continue
        for key in (name, name.rsplit(".", 1)[0] + ".img"):
            try:
                o0, shp0 = h.debug_tensor(n0, key)
                o1, shp1 = h.debug_tensor(n1, key)
            except Exception:
                continue
            e = 4 if key.endswith(".img") else esz
            out.append((key + "[sub0]", o0, int(np.prod(shp0)) * e, L["kernel"]))
            out.append((key + "[sub1]", base0 + sub0_bytes + (o1 - base0), int(np.prod(shp1)) * e, L["kernel"]))
    return out


R = regions()
h.set_streams(2)
need = h.workspace_bytes(batch)
print(f"migan-{res} batch {batch} ({n0}+{n1}) {dtype}: workspace {need / 2**30:.2f} GiB", flush=True)
runs = []
for i in range(3):
    ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    y = torch.empty((batch, 3, res, res), device=dev)
    h.forward(x.data_ptr(), y.data_ptr(), batch, ws.data_ptr(), need, stream)
    torch.cuda.synchronize()
    runs.append((ws, y))
for i in (1, 2):
    print(f"--- run {i} vs run 0: final image differing elements {int((runs[i][1] != runs[0][1]).sum())}")
    for key, off, nb, kern in R:
        nd = int((runs[i][0][off:off + nb] != runs[0][0][off:off + nb]).sum())
        if nd:
            d = (runs[i][0][off:off + nb] != runs[0][0][off:off + nb]).nonzero().flatten()
            print(f"  {key:34s} differing bytes {nd:8d} first at +{int(d[0])} last at +{int(d[-1])} of {nb}   {kern[:70]}", flush=True)
