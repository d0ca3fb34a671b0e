"""Pre/post-processing around the generator (reference scripts/demo.py:56-66, :135-140).

CPU: the oracle against outputs of the reference's own preprocess() (tests/golden/prepost.npz), and the product's
kernels + C ABI executed by the fiber emulator against the oracle -- bit-exact (byte / integer work and single fp32
roundings in a fixed order).  GPU: the same through libmigan_hip.so.
"""
import os

import numpy as np
import pytest

from oracle import migan_prepost as pp
from tests.emu_util import emu_lib


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "prepost.npz"))


def _aligned_u8(a):
    buf = np.empty(a.size + 16, dtype=np.uint8)
    off = (16 - buf.ctypes.data % 16) % 16
    out = buf[off:off + a.size].reshape(a.shape)
    out[...] = a
    return out


def _aligned_f32(shape, fill=np.nan):
    buf = np.full(int(np.prod(shape)) + 4, fill, dtype=np.float32)
    off = (16 - buf.ctypes.data % 16) % 16 // 4
    return buf[off:off + int(np.prod(shape))].reshape(shape)


def test_oracle_matches_the_reference_preprocess(golden):
    assert np.array_equal(pp.preprocess(golden["img"], golden["mask"]), golden["x"])
    assert np.array_equal(pp.compose(golden["y"], golden["img"], golden["mask"]), golden["composed"])


def test_every_byte_value_and_mask_value():
    img = np.arange(256, dtype=np.uint8).repeat(3).reshape(1, 16, 16, 3)
    mask = np.arange(256, dtype=np.uint8).reshape(1, 16, 16)
    x = pp.preprocess(img, mask)
    assert x.shape == (1, 4, 16, 16)
    assert set(np.unique(x[:, 0])) == {-0.5, 0.5} and (x[:, 0] == 0.5).sum() == 1        # only 255 keeps the pixel
    assert x[0, 1].reshape(-1)[255] == np.float32(1.0)                                   # 255 -> 1.0 (kept)
    assert (x[0, 1:].reshape(3, -1)[:, :255] == 0).all()                                 # holes are zeroed


@pytest.mark.parametrize("res,batch", [(8, 1), (16, 3), (64, 2)])
def test_kernels_in_the_emulator_are_bit_exact(res, batch):
    lib = emu_lib()
    rng = np.random.RandomState(res + batch)
    img = _aligned_u8(rng.randint(0, 256, size=(batch, res, res, 3)).astype(np.uint8))
    mask = _aligned_u8(((rng.rand(batch, res, res) > 0.5) * 255).astype(np.uint8))
    mask[0, 0, :3] = (254, 1, 128)
    x = _aligned_f32((batch, 4, res, res))
    lib.pack_input(img.ctypes.data, mask.ctypes.data, x.ctypes.data, batch, res)
    assert np.array_equal(x, pp.preprocess(img, mask))
    y = _aligned_f32((batch, 3, res, res))
    y[...] = (rng.randn(batch, 3, res, res) * 0.9).astype(np.float32)
    y[0, 0, 0, :8] = (-1.5, -1.0, -0.999999, 0.0, 0.5, 0.999999, 1.0, 1.7)
    out = _aligned_u8(np.zeros((batch, res, res, 3), np.uint8))
    lib.compose_output(y.ctypes.data, img.ctypes.data, mask.ctypes.data, out.ctypes.data, batch, res)
    assert np.array_equal(out, pp.compose(y, img, mask))


def test_bad_arguments():
    lib = emu_lib()
    a = _aligned_u8(np.zeros(4096, np.uint8))
    x = _aligned_f32((4096,))
    with pytest.raises(ValueError):
        lib.pack_input(a.ctypes.data, a.ctypes.data, x.ctypes.data, 1, 12)            # not a power of two
    with pytest.raises(ValueError):
        lib.pack_input(a.ctypes.data, a.ctypes.data, x.ctypes.data, 0, 16)            # empty batch
    with pytest.raises(ValueError):
        lib.pack_input(a.ctypes.data + 1, a.ctypes.data, x.ctypes.data, 1, 16)        # misaligned image
    with pytest.raises(ValueError):
        lib.compose_output(None, a.ctypes.data, a.ctypes.data, a.ctypes.data, 1, 16)  # null y


@pytest.mark.gpu
def test_gpu_pipeline_bit_exact_and_end_to_end(pkg, golden):
    import torch
    dev = torch.device("cuda:0")
    pipe = __import__("importlib").import_module("mi-gan_amd.pipeline")
    img, mask = torch.from_numpy(golden["img"]).to(dev), torch.from_numpy(golden["mask"]).to(dev)
    x = pipe.preprocess(img, mask)
    assert np.array_equal(x.cpu().numpy(), golden["x"])                                  # the reference's own preprocess()
    out = pipe.compose(torch.from_numpy(golden["y"]).to(dev), img, mask)
    assert np.array_equal(out.cpu().numpy(), golden["composed"])
    # larger, ragged-free batch at 512 against the oracle, then the demo.py call sequence end to end
    rng = np.random.RandomState(7)
    res, n = 512, 3
    img = rng.randint(0, 256, size=(n, res, res, 3)).astype(np.uint8)
    mask = ((rng.rand(n, res, res) > 0.3) * 255).astype(np.uint8)
    xd = pipe.preprocess(torch.from_numpy(img).to(dev), torch.from_numpy(mask).to(dev))
    assert np.array_equal(xd.cpu().numpy(), pp.preprocess(img, mask))
    res = 64
    img = rng.randint(0, 256, size=(2, res, res, 3)).astype(np.uint8)
    mask = ((rng.rand(2, res, res) > 0.3) * 255).astype(np.uint8)
    model = pkg.Generator(res)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pkg.synth.make_state_dict(res, seed=3).items()})
    model = model.to(dev).eval()
    imd, mkd = torch.from_numpy(img).to(dev), torch.from_numpy(mask).to(dev)
    with torch.no_grad():
        y = model(pipe.preprocess(imd, mkd))
    composed = pipe.compose(y, imd, mkd).cpu().numpy()
    assert np.array_equal(composed, pp.compose(y.cpu().numpy(), img, mask))
    keep = (mask == 255)
    assert np.array_equal(composed[keep], img[keep])                                     # known pixels pass through
    with pytest.raises(RuntimeError):
        pipe.preprocess(torch.from_numpy(img), torch.from_numpy(mask))                   # CPU tensors are refused
