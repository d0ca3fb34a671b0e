#!/usr/bin/env python
"""Per-layer error of the HIP forward against the torch-CPU oracle port (debug taps), then the
plain (ping-pong workspace) forward.  Usage: gpu_diag_taps.py RES BATCH [SEED] [f32|bf16|f16]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import migan_torch_cpu as torc  # noqa: E402

pkg = importlib.import_module("mi-gan_amd")
res, batch = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 31
dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"
storage = None if dtype == "f32" else dtype
THR = 1e-3 if storage is None else (0.25 if dtype == "bf16" else 0.03)
dev = torch.device("cuda:0")
lib = pkg.load_library()
sd = pkg.synth.make_state_dict(res, seed=seed)
x = pkg.synth.make_input(batch, res, seed=seed)
taps = {}
want = torc.generator(x, sd, res, taps=taps, storage=storage)
dsd = {k: torch.from_numpy(v.reshape(1) if v.ndim == 0 else v).to(dev) for k, v in sd.items()}
stream = int(torch.cuda.current_stream().cuda_stream)
for debug in (True, False):
    h = pkg.hipbind.MiganHandle(lib, res, 0, dtype=dtype)
    h.set_debug(debug)
    for name, shape, _ in h.weights():
        h.set_weight(name, dsd[name].data_ptr(), shape)
    h.commit(stream)
    ws = torch.zeros(h.workspace_bytes(batch), dtype=torch.uint8, device=dev)
    xd = torch.from_numpy(x).to(dev)
    y = torch.empty((batch, 3, res, res), device=dev)
    for rep in range(2):
        h.forward(xd.data_ptr(), y.data_ptr(), batch, ws.data_ptr(), ws.numel(), stream)
        torch.cuda.synchronize()
        print(f"debug={debug} rep={rep} final err {float((y.cpu() - want).abs().max()):.3e}", flush=True)
    if not debug:
        continue
    for name, ref in taps.items():
        key = name[:-5] if name.endswith(".skip") else name
        if name.endswith(".conv1") and name.startswith("synthesis") and (name + ".skip") in taps:
            continue
        if key == f"synthesis.b{res}.img":
            continue
        off, shape = h.debug_tensor(batch, key)
        n = int(np.prod(shape))
        if key.endswith(".img") or storage is None:
            t = ws[off:off + 4 * n].view(torch.float32).reshape(shape).cpu()
        else:
            t = ws[off:off + 2 * n].view(torch.bfloat16 if dtype == "bf16" else torch.float16).reshape(shape).float().cpu()
        got = t if key.endswith(".img") else t.permute(0, 3, 1, 2)
        d = (got - ref).abs()
        bad = (d > THR).nonzero()
        print(f"  {name:28s} err {float(d.max()):.3e} absmax {float(ref.abs().max()):.2f} nbad {len(bad)}"
              + (f" first {bad[0].tolist()} last {bad[-1].tolist()}" if len(bad) else ""), flush=True)
    d = (y.cpu() - want).abs()
    print("per (b,ch) max", d.amax(dim=(2, 3)).tolist())
    bad = d.amax(dim=(0, 1)) > THR
    print("bad pixels", int(bad.sum()), "of", bad.numel())
    ys, xs = bad.nonzero(as_tuple=True)
    if len(ys):
        print("bad by y%8", np.bincount(ys.numpy() % 8, minlength=8).tolist())
        print("bad by x%16", np.bincount(xs.numpy() % 16, minlength=16).tolist())
        print("bad by y//32", np.bincount(ys.numpy() // 32, minlength=res // 32).tolist())
        print("bad by x//32", np.bincount(xs.numpy() // 32, minlength=res // 32).tolist())
        print("first", [(int(a), int(b)) for a, b in zip(ys[:12], xs[:12])])
        yy, xx = int(ys[0]), int(xs[0])
        print("got", y.cpu()[0, :, yy, xx].tolist(), "want", want[0, :, yy, xx].tolist())
