#!/usr/bin/env python
"""Run-to-run determinism of every stored tensor at full size: the debug-mode forward (every layer keeps its output) twice on
the same input, per-layer count of differing elements; then the plain forward with 1 and 2 streams.
usage: gpu_diag_determinism.py RES BATCH DTYPE"""
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
sd = pkg.synth.make_state_dict(res, seed=0)
x = torch.from_numpy(pkg.synth.make_input(batch, res, seed=100, kind="demo")).to(dev)
dsd = {k: torch.from_numpy(v.reshape(1) if v.ndim == 0 else v).to(dev) for k, v in sd.items()}
stream = int(torch.cuda.current_stream().cuda_stream)
h = pkg.hipbind.MiganHandle(lib, res, 0, dtype=dtype)
h.set_debug(True)
for name, shape, _ in h.weights():
    h.set_weight(name, dsd[name].data_ptr(), shape)
h.commit(stream)
need = h.workspace_bytes(batch)
print(f"migan-{res} batch {batch} {dtype}: debug workspace {need / 2**30:.2f} GiB", flush=True)
ws = [torch.zeros(need, dtype=torch.uint8, device=dev) for _ in range(2)]
ys = [torch.empty((batch, 3, res, res), device=dev) for _ in range(2)]
for i in range(2):
    h.forward(x.data_ptr(), ys[i].data_ptr(), batch, ws[i].data_ptr(), need, stream)
torch.cuda.synchronize()
esz = 4 if dtype == "f32" else 2
for L in h.launches():
    name = L["layer"]
    if name.endswith(".dwfir") or name.endswith(".torgb"):
        continue
    for key in (name, name.rsplit(".", 1)[0] + ".img"):
        try:
            off, shape = h.debug_tensor(batch, key)
        except Exception:
            continue
        n = int(np.prod(shape)) * (4 if key.endswith(".img") else esz)
        a, b = ws[0][off:off + n], ws[1][off:off + n]
        nd = int((a != b).sum())
        if nd or key == name:
            print(f"  {key:30s} {L['kernel'][:78]:78s} differing bytes {nd}", flush=True)
print("final image differing elements", int((ys[0] != ys[1]).sum()))
del ws
m = pkg.Generator(resolution=res, activation_dtype=dtype)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
m = m.to(dev).eval()
with torch.no_grad():
    m.set_streams(1)
    a = m(x).clone(); b = m(x).clone()
    m.set_streams(2)
    c = m(x).clone(); d = m(x).clone()
print("plain forward: s1 vs s1", int((a != b).sum()), " s2 vs s2", int((c != d).sum()), " s1 vs s2", int((a != c).sum()))
