#!/usr/bin/env python
"""Diagnostics of the fused ToRGB tail at full size: one plain SeparableConv2d + ToRGB through migan_sepconv_forward, several runs,
image planes vs (a) the oracle and (b) ToRGB recomputed on the host from the feature map the SAME launch stored.
usage: gpu_diag_torgb.py C H STORAGE [prev]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import migan_oracle as orc  # noqa: E402
from tests.emu_util import from_storage, nchw, nhwc, to_storage  # noqa: E402

pkg = importlib.import_module("mi-gan_amd")
c, h, storage = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
with_prev = len(sys.argv) > 4 and sys.argv[4] == "prev"
batch, seed = 2, 9
dev = torch.device("cuda:0")
lib = pkg.load_library()
s = pkg.synth
w1 = (s.normal((c, 1, 3, 3), seed, "w1") * 0.4).astype(np.float32)
b1 = (s.normal((c,), seed, "b1") * 0.5).astype(np.float32)
w2 = (s.normal((c, c, 1, 1), seed, "w2") / np.sqrt(c)).astype(np.float32)
nc = s.normal((h, h), seed, "nc").astype(np.float32)
ns = np.asarray([0.37], dtype=np.float32)
tw = (s.normal((3, c, 1, 1), seed, "tw") / np.sqrt(c)).astype(np.float32)
tb = (s.normal((3,), seed, "tb") * 0.2).astype(np.float32)
x = orc.round_storage((s.normal((batch, c, h, h), seed, "x") * 1.5).astype(np.float32), storage)
prev = s.normal((batch, 3, h // 2, h // 2), seed, "prev").astype(np.float32) if with_prev else None


def put(a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int16).copy() if a.dtype == np.uint16 else a.copy()).to(dev)


xin = put(to_storage(nhwc(x), storage))
D = {k: put(v) for k, v in dict(w1=w1, b1=b1, w2=w2, nc=nc, ns=ns, tw=tw, tb=tb).items()}
pv = put(prev) if with_prev else None
wsp = torch.zeros((3 * c * c + 1) // 2 + 8, device=dev)
outs = []
for rep in range(int(os.environ.get("REPS", "4"))):
    y = put(to_storage(np.full((batch, h, h, c), np.nan, np.float32), storage))
    img = torch.full((batch, 3, h, h), float("nan"), device=dev)
    lib.sepconv_forward(stream=int(torch.cuda.current_stream().cuda_stream), x=xin.data_ptr(), y=y.data_ptr(),
                        conv1_weight=D["w1"].data_ptr(), conv1_bias=D["b1"].data_ptr(), conv2_weight=D["w2"].data_ptr(),
                        noise_const=D["nc"].data_ptr(), noise_strength=D["ns"].data_ptr(), torgb_weight=D["tw"].data_ptr(),
                        torgb_bias=D["tb"].data_ptr(), img_prev=None if pv is None else pv.data_ptr(), img_out=img.data_ptr(),
                        batch=batch, cin=c, cout=c, res_in=h, wsplit=wsp.data_ptr(), wsplit_bytes=wsp.numel() * 4,
                        dtype=pkg.hipbind.dtype_code(storage), gemm=2)
    torch.cuda.synchronize()
    ya = y.cpu().numpy()
    feat = nchw(from_storage(ya.view(np.uint16) if ya.dtype == np.int16 else ya, storage))
    outs.append((feat, img.cpu().numpy()))
feat0, img0 = outs[0]
for r, (f, im) in enumerate(outs[1:], 1):
    print(f"run {r} vs run 0: features differ at {int((f != feat0).sum())} elements, image at {int((im != img0).sum())} pixels-channels")
check = orc.pointwise(feat0, tw, tb)
if with_prev:
    check = check + orc.upsample2d(prev)
d = np.abs(img0 - check)
print("image vs ToRGB of the launch's own stored features: max", float(d.max()), "per channel", d.max(axis=(0, 2, 3)).tolist())
bad = np.argwhere(d > 1e-3 * max(1.0, float(np.abs(check).max())))
print("bad entries", len(bad), "of", d.size)
if len(bad):
    b, ch, yy, xx = bad.T
    print("by channel", np.bincount(ch, minlength=3).tolist(), "by batch", np.bincount(b, minlength=batch).tolist())
    print("by y%8", np.bincount(yy % 8, minlength=8).tolist())
    print("by x%16", np.bincount(xx % 16, minlength=16).tolist())
    for e in bad[:10]:
        bb, cc, y_, x_ = (int(v) for v in e)
        print("  ", e.tolist(), "got", float(img0[bb, cc, y_, x_]), "check", float(check[bb, cc, y_, x_]), "diff", float(img0[bb, cc, y_, x_] - check[bb, cc, y_, x_]))
