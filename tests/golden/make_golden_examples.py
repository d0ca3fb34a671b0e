#!/usr/bin/env python3
"""Real-data goldens (SURVEY section 8d / 4 tier 5): image/mask pairs of the reference's examples/places2_512_object through
the reference's OWN scripts/demo.py functions (resize :47-53, read_mask :26-44, preprocess :56-66) and the reference
Generator (seeded weights; the pretrained checkpoints are not in the repository), plus demo.py's post-processing and
composition (:135-140, at network resolution).

Stored per case (tests/golden/examples_*.npz): the network-resolution uint8 image and mask demo.py feeds to preprocess()
(np.array(img.resize((R,R), BICUBIC)) etc.), checksums of x = preprocess(...), the reference output y (strided + sums) and the
composed uint8 image.  Build container only (imports /root/reference; cv2 is not installed and is only used by demo.py for
the final resize back to the original size, which is not part of this fixture).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_examples.py
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MIGAN_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
if "cv2" not in sys.modules:
    sys.modules["cv2"] = types.ModuleType("cv2")            # imported at module level by demo.py; unused by the functions called here
spec = importlib.util.spec_from_file_location("ref_demo", os.path.join(REF, "scripts", "demo.py"))
demo = importlib.util.module_from_spec(spec)
spec.loader.exec_module(demo)
import lib.model_zoo.migan_inference as ref                 # noqa: E402

pkg = importlib.import_module("mi-gan_amd")
torch.set_num_threads(max(1, os.cpu_count() or 1))
EX = os.path.join(REF, "examples", "places2_512_object")


def case(tag, resolution, names, seed, stride):
    sd_np = pkg.synth.make_state_dict(resolution, seed=seed, regime="export")
    g = ref.Generator(resolution=resolution)
    g.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    g.eval()
    imgs, masks, xs, sizes = [], [], [], []
    for n in names:
        img = Image.open(os.path.join(EX, "images", f"{n}.png")).convert("RGB")               # demo.py:125
        img_resized = demo.resize(img, max_size=resolution)                                     # :126
        mask = demo.read_mask(os.path.join(EX, "masks", f"{n}.png"), invert=False)              # :127
        mask_resized = demo.resize(mask, max_size=resolution, interpolation=Image.NEAREST)      # :128
        x = demo.preprocess(img_resized, mask_resized, resolution)                              # :130
        # what preprocess() turns into arrays (:57-60): the inputs of the GPU-side pack kernel
        imgs.append(np.array(img_resized.resize((resolution, resolution), Image.BICUBIC)))
        masks.append(np.array(mask_resized.resize((resolution, resolution), Image.NEAREST)))
        xs.append(x)
        sizes.append(img.size)
    x = torch.cat(xs, dim=0)
    img_u8, mask_u8 = np.stack(imgs), np.stack(masks)
    with torch.no_grad():
        y = g(x)
    result = ((y * 0.5 + 0.5).clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1).numpy()   # :135-136
    m = (mask_u8[:, :, :, np.newaxis] // 255)                                                    # :139
    composed = img_u8 * m + result * (1 - m)                                                     # :140
    print(tag, "sizes", sizes, "holes", [float((mk != 255).mean()) for mk in mask_u8], "|y|max", float(y.abs().max()))
    yn = y.numpy()
    np.savez_compressed(os.path.join(HERE, f"examples_{tag}.npz"), resolution=resolution, seed=seed, names=np.array(names),
                        orig_sizes=np.array(sizes), img_u8=img_u8, mask_u8=mask_u8,
                        x_sum=x.double().sum(dim=(2, 3)).numpy(), x_abs_sum=x.double().abs().sum(dim=(2, 3)).numpy(),
                        stride=stride, y=yn[:, :, ::stride, ::stride].astype(np.float32), y_sum=yn.astype(np.float64).sum(axis=(2, 3)),
                        y_absmax=float(np.abs(yn).max()), composed=composed.astype(np.uint8))


if __name__ == "__main__":
    case("p256", 256, ["1", "16", "5"], 61, 2)        # 512x343 (landscape), 374x512 (portrait), ... resized to 256 x 256
    case("p512", 512, ["10", "17"], 62, 4)            # migan-512 on 512x380 and 384x512 originals
