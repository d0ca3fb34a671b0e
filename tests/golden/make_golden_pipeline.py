#!/usr/bin/env python3
"""Goldens of the DEPLOYED pre/post-processing (second half): scripts/create_onnx_pipeline.py
::MIGAN_Pipeline.forward (:252-264) -- masked bounding box + padding (:132-231), crop, bilinear resize to the network
resolution and nearest resize of the mask (:233-239), the generator, bilinear resize back, 3x3 max-pool + 5x5 gaussian feathering
of the mask and the blend (:241-250) -- run in torch on CPU in the build container on image/mask pairs of the reference's
examples/ (non-square originals) with seeded generator weights (the checkpoints are not in the repository).

The reference module imports three packages that are not installed here; two are unused by the module (cv2, onnxruntime) and get
empty stand-ins, the third is torchvision, of which the pipeline calls exactly one function on tensors:
`torchvision.transforms.functional.resize`.  The stand-in below restates what that function does for a TENSOR argument in the
torchvision the reference pins (0.9, requirements.txt; transforms/functional_tensor.py::resize): cast to float32, F.interpolate
(bilinear with align_corners=False and NO antialias, or nearest), round and cast back for integer dtypes.  Everything else is the
reference's code.

Stored per case (tests/golden/pipeline_*.npz): uint8 image [3,H,W] and mask [1,H,W], the bounding box, checksums of the network
input, the result image, and the generator weights' seed.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pipeline.py
"""
import importlib
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MIGAN_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True


def tv_resize(img, size, interpolation=Image.BILINEAR, max_size=None, antialias=None):
    """torchvision 0.9 transforms.functional.resize for a tensor image (functional_tensor.resize)"""
    assert isinstance(img, torch.Tensor)
    size = [int(s) for s in size]
    mode = {Image.BILINEAR: "bilinear", Image.NEAREST: "nearest"}[int(interpolation)]
    out_dtype = img.dtype
    need_cast = out_dtype not in (torch.float32, torch.float64)
    x = img.to(torch.float32) if need_cast else img
    x = F.interpolate(x, size=size, mode=mode, align_corners=False if mode == "bilinear" else None)
    if need_cast:
        if out_dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
            x = torch.round(x)
        x = x.to(out_dtype)
    return x


for name in ("cv2", "onnxruntime"):
    sys.modules.setdefault(name, types.ModuleType(name))
tv = types.ModuleType("torchvision")
tvt = types.ModuleType("torchvision.transforms")
tvf = types.ModuleType("torchvision.transforms.functional")
tvf.resize = tv_resize
tv.transforms, tvt.functional = tvt, tvf
sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
spec = importlib.util.spec_from_file_location("ref_pipeline", os.path.join(REF, "scripts", "create_onnx_pipeline.py"))
refp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(refp)

pkg = importlib.import_module("mi-gan_amd")
torch.set_num_threads(max(1, os.cpu_count() or 1))


def run_case(tag, resolution, example_set, name, seed, padding=128, synth_mask=None):
    sd_np = pkg.synth.make_state_dict(resolution, seed=seed, regime="export")
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "g.pt")
        torch.save({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, path)
        pipe = refp.MIGAN_Pipeline(model_path=path, resolution=resolution, padding=padding)
    ex = os.path.join(REF, "examples", example_set)
    ipath = [os.path.join(ex, "images", name + e) for e in (".png", ".jpg") if os.path.exists(os.path.join(ex, "images", name + e))][0]
    img = np.array(Image.open(ipath).convert("RGB"))
    if synth_mask is None:
        mask = np.array(refp.read_mask(os.path.join(ex, "masks", f"{name}.png"), invert=False))
    else:
        mask = synth_mask(img.shape[0], img.shape[1])
    image_t = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)[None]))         # (1, 3, H, W) uint8, as main() builds it (:336)
    mask_t = torch.from_numpy(np.ascontiguousarray(mask[None, None]))
    captured = {}
    orig_pre = pipe.preprocess

    def spy_pre(image, mask):
        x = orig_pre(image, mask)
        captured["x"] = x
        return x
    pipe.preprocess = spy_pre
    with torch.no_grad():
        bbox = [int(v) for v in pipe.get_masked_bbox(mask_t)]
        out = pipe(image_t.clone(), mask_t.clone())
    x = captured["x"].numpy()
    np.savez_compressed(os.path.join(HERE, f"pipeline_{tag}.npz"), image=image_t[0].numpy(), mask=mask_t[0].numpy(), bbox=np.array(bbox, np.int32),
                        x_sum=np.float64(x.astype(np.float64).sum()), x_abs_sum=np.float64(np.abs(x.astype(np.float64)).sum()),
                        x_strided=x[:, :, ::7, ::5].copy(), result=out[0].numpy(), resolution=np.int32(resolution), seed=np.int32(seed),
                        padding=np.int32(padding))
    changed = int((out[0].numpy() != image_t[0].numpy()).any(axis=0).sum())
    print(f"{tag}: image {img.shape} bbox x[{bbox[0]},{bbox[1]}) y[{bbox[2]},{bbox[3]}) changed pixels {changed}")


def small_hole(h, w):
    m = np.full((h, w), 255, np.uint8)
    m[h // 3:h // 3 + 40, w // 2:w // 2 + 25] = 0
    m[h // 3 + 60, w // 2 - 30] = 0
    return m


def edge_hole(h, w):
    m = np.full((h, w), 255, np.uint8)
    m[:30, w - 50:] = 0                       # touches the top-right corner: the crop window is pushed back inside the image
    return m


if __name__ == "__main__":
    run_case("r256_object_3", 256, "places2_512_object", "3", seed=21)               # object mask, 512 x 343 original
    run_case("r256_small_hole", 256, "places2_512_object", "10", seed=22, synth_mask=small_hole, padding=64)
    run_case("r256_edge_hole", 256, "places2_512_object", "11", seed=23, synth_mask=edge_hole, padding=32)
    run_case("r64_freeform", 64, "places2_256_freeform", "Places365_val_00000262", seed=24, padding=16)             # free-form mask, 256 x 256
