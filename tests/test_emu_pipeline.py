"""The deployed-pipeline kernels (migan_pipeline_bbox / _pre / _post; reference scripts/create_onnx_pipeline.py:118-264) run on
the CPU through the fiber emulator, against the goldens generated from the reference's MIGAN_Pipeline and the oracle pinned to
them.  The kernels follow torch operation by operation (see migan_pipeline.hpp), and on these four cases everything is bit-exact:
the bbox, the network input x, and -- given the same generator output -- every byte of the result image."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import migan_pipeline_oracle as po
from oracle import migan_torch_cpu as torc
from tests.emu_util import emu_lib, ptr

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "pipeline_*.npz")))


def run_pipeline(lib, image, mask, generator, res, padding, gauss=None):
    """image [3,H,W] uint8 (modified in place), mask [H,W] uint8; host buffers stand in for device memory under the emulator"""
    h, w = mask.shape
    scratch = np.zeros(lib.pipeline_scratch_bytes(h, w), dtype=np.uint8)
    bbox = lib.pipeline_bbox(ptr(mask), h, w, res, padding, ptr(scratch))
    x = np.zeros((1, 4, res, res), dtype=np.float32)
    lib.pipeline_pre(ptr(image), ptr(mask), h, w, bbox, res, ptr(x))
    y = np.ascontiguousarray(generator(x), dtype=np.float32)
    lib.pipeline_post(ptr(image), ptr(mask), h, w, bbox, res, ptr(y), ptr(scratch), gauss25=gauss)
    return bbox, x, y


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[9:-4] for p in CASES])
def test_pipeline_kernels_match_the_reference_goldens(pkg, path):
    g = np.load(path)
    res, seed, padding = int(g["resolution"]), int(g["seed"]), int(g["padding"])
    sd = pkg.synth.make_state_dict(res, seed=seed, regime="export")
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    lib = emu_lib()
    image = np.array(g["image"], copy=True)
    mask = np.ascontiguousarray(g["mask"][0])
    # the generator itself is not under test here: both sides get the oracle's output for the ORACLE's x, so that the comparison
    # of the result isolates the post-processing kernel
    want_img, want_bbox, want_x = po.pipeline(g["image"], g["mask"], lambda t: torc.generator(t.numpy(), sd, res), res, padding)
    y_ref = torc.generator(want_x, sd, res)
    bbox, x, _ = run_pipeline(lib, image, mask, lambda _x: y_ref, res, padding, gauss=po.gaussian_kernel().flatten().tolist())
    assert list(bbox) == [int(v) for v in g["bbox"]] == list(want_bbox)
    np.testing.assert_array_equal(x, want_x)
    np.testing.assert_array_equal(x[:, :, ::7, ::5], g["x_strided"])
    np.testing.assert_array_equal(image, g["result"])
    # pixels outside the crop are untouched
    x0, x1, y0, y1 = bbox
    outside = np.ones(mask.shape, dtype=bool)
    outside[y0:y1, x0:x1] = False
    np.testing.assert_array_equal(image[:, outside], g["image"][:, outside])


def test_builtin_gaussian_equals_the_reference_buffer(pkg):
    """gauss25 = NULL computes GaussianSmoothing(kernel_size=5, sigma=1) in C++: same result image as with torch's weights"""
    g = np.load(CASES[-1])
    res, padding = int(g["resolution"]), int(g["padding"])
    lib = emu_lib()
    rng = np.random.default_rng(3)
    y = rng.standard_normal((1, 3, res, res)).astype(np.float32) * 0.6
    mask = np.ascontiguousarray(g["mask"][0])
    a, b = np.array(g["image"], copy=True), np.array(g["image"], copy=True)
    run_pipeline(lib, a, mask, lambda _x: y, res, padding, gauss=None)
    run_pipeline(lib, b, mask, lambda _x: y, res, padding, gauss=po.gaussian_kernel().flatten().tolist())
    assert np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= 1
    assert (a != b).mean() < 1e-3


def test_full_mask_and_no_hole(pkg):
    """all-255 mask: no masked column -> the box is centred by the reference's min/max defaults; nothing may change in the image
    beyond the reference's own -1 rounding of known pixels.  all-0 mask: the whole crop is replaced."""
    lib = emu_lib()
    rng = np.random.default_rng(5)
    h, w, res = 96, 80, 64
    img = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    y = rng.standard_normal((1, 3, res, res)).astype(np.float32) * 0.6
    for fill in (255, 0):
        mask = np.full((h, w), fill, dtype=np.uint8)
        got = np.array(img, copy=True)
        bbox, x, _ = run_pipeline(lib, got, mask, lambda _x: y, res, 8)
        want, wbox, wx = po.pipeline(img, mask[None], lambda t: torch.from_numpy(y), res, 8)
        assert list(bbox) == list(wbox)
        assert np.abs(x - wx).max() <= 2.0 / 255.0 + 1e-6
        assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


def test_mask_of_another_size_is_resized_like_torchvision_nearest(pkg):
    """MIGAN_Pipeline.forward's first line (:256): tvF.resize(mask, image size, NEAREST) -- against the oracle's torchvision restatement"""
    lib = emu_lib()
    rng = np.random.default_rng(8)
    for (ih, iw), (h, w) in (((37, 53), (96, 80)), ((128, 128), (50, 70)), ((64, 48), (64, 48))):
        m = np.ascontiguousarray((rng.random((ih, iw)) > 0.4).astype(np.uint8) * 255)
        out = np.zeros((h, w), dtype=np.uint8)
        lib.pipeline_mask_resize(ptr(m), ih, iw, ptr(out), h, w)
        want = po.tv_resize(torch.from_numpy(m)[None, None], (h, w), "nearest")[0, 0].numpy()
        np.testing.assert_array_equal(out, want)


def test_pipeline_argument_errors(pkg):
    lib = emu_lib()
    mask = np.zeros((32, 32), dtype=np.uint8)
    scratch = np.zeros(lib.pipeline_scratch_bytes(32, 32), dtype=np.uint8)
    with pytest.raises(ValueError):
        lib.pipeline_bbox(ptr(mask), 32, 32, 48, 8, ptr(scratch))           # resolution not a power of two
    with pytest.raises(ValueError):
        lib.pipeline_bbox(ptr(mask), 2, 32, 64, 8, ptr(scratch))            # too small for the reflect padding
    x = np.zeros((1, 4, 64, 64), dtype=np.float32)
    img = np.zeros((3, 32, 32), dtype=np.uint8)
    with pytest.raises(ValueError):
        lib.pipeline_pre(ptr(img), ptr(mask), 32, 32, (0, 40, 0, 32), 64, ptr(x))   # box outside the image
