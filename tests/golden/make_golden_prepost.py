#!/usr/bin/env python
"""Golden vectors for the pre/post-processing around the generator, produced by the
REFERENCE code in the build container (/root/reference; not available on the GPU box):

  * x = scripts/demo.py::preprocess(img, mask, R) -- the reference function itself, called on PIL images that
    are already R x R (its two PIL resizes are then identity copies);
  * composed = the expressions of scripts/demo.py:135-140 executed with torch exactly as written there (they are
    inline in main(), not a function), without the cv2 resize (same size).

    python tests/golden/make_golden_prepost.py      -> tests/golden/prepost.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
if "cv2" not in sys.modules:                      # demo.py imports cv2 at module level; preprocess() does not use it
    sys.modules["cv2"] = types.ModuleType("cv2")
spec = importlib.util.spec_from_file_location("ref_demo", os.path.join(REF, "scripts", "demo.py"))
demo = importlib.util.module_from_spec(spec)
spec.loader.exec_module(demo)

R, N = 32, 3
rng = np.random.RandomState(1234)
img = rng.randint(0, 256, size=(N, R, R, 3)).astype(np.uint8)
mask = (rng.rand(N, R, R) > 0.4).astype(np.uint8) * 255
mask[0, :4, :4] = 254                              # "almost white" mask pixels are holes (demo.py:44,60)
mask[1, 5:9, 7] = 128
img[2, 0, :, :] = 255
img[2, 1, :, :] = 0
xs, outs, ys = [], [], []
for n in range(N):
    x = demo.preprocess(Image.fromarray(img[n]), Image.fromarray(mask[n]).convert("L"), R)     # demo.py:56-66
    xs.append(x.numpy()[0])
    y = torch.from_numpy((rng.randn(3, R, R) * 0.8).astype(np.float32))
    y[0, 0, :8] = torch.tensor([-1.5, -1.0, -0.999999, 0.0, 0.5, 0.999999, 1.0, 1.7])
    ys.append(y.numpy())
    result_image = (y * 0.5 + 0.5).clamp(0, 1) * 255                                               # demo.py:135
    result_image = result_image.to(torch.uint8).permute(1, 2, 0).detach().to("cpu").numpy()       # demo.py:136
    mask_resized = np.array(Image.fromarray(mask[n]).convert("L"))[:, :, np.newaxis] // 255       # demo.py:139
    composed_img = img[n] * mask_resized + result_image * (1 - mask_resized)                       # demo.py:140
    outs.append(composed_img.astype(np.uint8))
np.savez_compressed(os.path.join(HERE, "prepost.npz"), img=img, mask=mask, x=np.stack(xs), y=np.stack(ys),
                    composed=np.stack(outs))
print("wrote", os.path.join(HERE, "prepost.npz"), np.stack(xs).shape)
