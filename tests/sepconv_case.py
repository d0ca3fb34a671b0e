"""One SeparableConv2d through the C ABI entry migan_sepconv_forward in any storage format / GEMM variant / geometry,
compared with the numpy oracle in the same storage mode.  Shared by the CPU tests (product kernel source on the fiber
emulator, host memory) and the GPU tests (libmigan_hip.so, device memory).  Test infrastructure only."""
import numpy as np

from oracle import migan_oracle as orc
from tests.emu_util import aligned, from_storage, nchw, nhwc, storage_close, storage_ulp, to_storage


class HostMem:
    """numpy arrays are the 'device' memory (emulator)"""
    stream = 0

    def put(self, a):
        return a

    def ptr(self, a):
        return None if a is None else a.ctypes.data

    def get(self, a):
        return a

    def sync(self):
        pass


class CudaMem:
    """torch tensors on the GPU; 16-bit arrays travel as int16 bit patterns"""

    def __init__(self, device):
        import torch
        self.torch = torch
        self.device = device
        self.stream = int(torch.cuda.current_stream(device).cuda_stream)

    def put(self, a):
        if a is None:
            return None
        a = np.ascontiguousarray(a)
        if a.dtype == np.uint16:
            return self.torch.from_numpy(a.view(np.int16).copy()).to(self.device)
        return self.torch.from_numpy(a.copy()).to(self.device)

    def ptr(self, t):
        return None if t is None else t.data_ptr()

    def get(self, t):
        a = t.cpu().numpy()
        return a.view(np.uint16) if a.dtype == np.int16 else a

    def sync(self):
        self.torch.cuda.synchronize()


def weights(pkg, cin, cout, seed, res_out, noise, wscale=1.0):
    s = pkg.synth
    sd = {
        "m.conv1.weight": (s.normal((cin, 1, 3, 3), seed, "w1") * 0.4).astype(np.float32),
        "m.conv1.bias": (s.normal((cin,), seed, "b1") * 0.5).astype(np.float32),
        "m.conv2.weight": (s.normal((cout, cin, 1, 1), seed, "w2") / np.sqrt(cin) * np.float32(wscale)).astype(np.float32),
    }
    if noise:
        sd["m.noise_const"] = s.normal((res_out[0], res_out[1]), seed, "nc").astype(np.float32)
        sd["m.noise_strength"] = np.asarray(0.37, dtype=np.float32)
    return sd


def run_sepconv_case(lib, pkg, mem, *, cin, cout, h, w=None, batch, down=1, up=1, noise=False, skip=False, seed=1, storage="f32",
                     gemm=-1, fromrgb=False, torgb=False, with_prev=False, wscale=1.0, nan_at=None):
    w = w or h
    ho, wo = (h // 2, w // 2) if down == 2 else ((h * 2, w * 2) if up == 2 else (h, w))
    sd = weights(pkg, cin, cout, seed, (ho, wo), noise, wscale)
    osd = dict(sd)
    if down == 2:
        osd["m.downsample.filter.weight"] = np.broadcast_to(orc.fir_taps(1.0), (cin, 1, 4, 4)).astype(np.float32)
    if up == 2:
        osd["m.upsample.filter.weight"] = np.broadcast_to(orc.fir_taps(4.0), (cout, 1, 4, 4)).astype(np.float32)
    kw = {}
    keep = []

    def dev(a):
        keep.append(mem.put(aligned(a) if a.dtype == np.float32 else a))
        return keep[-1]

    tw = None
    if fromrgb:
        fw = (pkg.synth.normal((cin, 4, 1, 1), seed, "fw") * 0.7).astype(np.float32)
        fb = (pkg.synth.normal((cin,), seed, "fb") * 0.3).astype(np.float32)
        img = (pkg.synth.normal((batch, 4, h, w), seed, "img") * 0.8).astype(np.float32)
        x = orc.lrelu_agc(orc.pointwise(img, fw, fb))                  # reference :194-195 (not a stored tensor)
        xin = dev(img)
        kw.update(fromrgb_weight=mem.ptr(dev(fw)), fromrgb_bias=mem.ptr(dev(fb)))
    else:
        x = orc.round_storage((pkg.synth.normal((batch, cin, h, w), seed, "x") * 1.5).astype(np.float32), storage)
        if nan_at is not None:
            x[nan_at] = np.nan                       # NaN-propagating builds only (Generator(nan_policy="propagate")): the mask must follow the oracle
        xin = dev(to_storage(nhwc(x), storage))
    gemm16 = storage != "f32" and gemm != 2          # 16-bit storage: GEMM variant "f16" unless f16x2 (2) is asked for
    want = orc.separable_conv(x.copy(), osd, "m", gemm16)
    sk = None
    if skip:
        sk = orc.round_storage(pkg.synth.normal((batch, cout, ho, wo), seed, "skip").astype(np.float32), storage)
        want = want + sk
    want = orc.round_storage(want, storage)
    y = dev(to_storage(np.full((batch, ho, wo, cout), np.nan, dtype=np.float32), storage))
    skh = dev(to_storage(nhwc(sk), storage)) if skip else None
    w1, b1, w2 = dev(sd["m.conv1.weight"]), dev(sd["m.conv1.bias"]), dev(sd["m.conv2.weight"])
    nc = dev(sd["m.noise_const"]) if noise else None
    ns = dev(sd["m.noise_strength"].reshape(1)) if noise else None
    scratch = dev(np.full((batch, ho, wo, cin), np.nan, dtype=np.float32)) if down == 2 else None
    wsp_n = (3 * cin * cout + 1) // 2 + 8
    wsp = dev(np.full(wsp_n, np.nan, dtype=np.float32))
    img_out = want_img = None
    if torgb:
        tw = (pkg.synth.normal((3, cout, 1, 1), seed, "tw") / np.sqrt(cout)).astype(np.float32)
        tb = (pkg.synth.normal((3,), seed, "tb") * 0.2).astype(np.float32)
        img_out = dev(np.full((batch, 3, ho, wo), np.nan, np.float32))
        want_img = orc.pointwise(want, tw, tb)                          # ToRGB reads the stored (rounded) tensor
        kw.update(torgb_weight=mem.ptr(dev(tw)), torgb_bias=mem.ptr(dev(tb)), img_out=mem.ptr(img_out))
        if with_prev:
            prev = pkg.synth.normal((batch, 3, ho // 2, wo // 2), seed, "prev").astype(np.float32)
            want_img = want_img + orc.upsample2d(prev)
            kw.update(img_prev=mem.ptr(dev(prev)))
    lib.sepconv_forward(stream=mem.stream, x=mem.ptr(xin), y=mem.ptr(y), skip=mem.ptr(skh), conv1_weight=mem.ptr(w1),
                        conv1_bias=mem.ptr(b1), conv2_weight=mem.ptr(w2), noise_const=mem.ptr(nc), noise_strength=mem.ptr(ns),
                        batch=batch, cin=cin, cout=cout, res_in=h, width_in=w, down=down, up=up,
                        scratch=mem.ptr(scratch), scratch_bytes=0 if scratch is None else batch * ho * wo * cin * 4,
                        wsplit=mem.ptr(wsp), wsplit_bytes=wsp_n * 4, dtype=pkg.hipbind.dtype_code(storage), gemm=gemm, **kw)
    mem.sync()
    got = nchw(from_storage(mem.get(y), storage))
    if nan_at is not None:
        assert np.array_equal(np.isnan(got), np.isnan(want)), "NaN mask differs from the oracle's"
        assert np.isnan(want).any() and not np.isnan(want).all()
        got = np.where(np.isnan(want), 0.0, got).astype(np.float32)
        want = np.where(np.isnan(want), 0.0, want).astype(np.float32)
    assert np.isfinite(got).all(), "kernel left NaNs (unwritten output or read of unwritten LDS)"
    # "f16" GEMM variant: an fp16 operand that rounded the other way (the two sides differ in the last place before rounding)
    # moves the fp32 sum by 2^-11 of one product
    storage_close(got, want, storage, ulps=1, frac=0.05 if gemm16 else 0.02, top_ulps=0.5 if gemm16 else 0.0)
    if torgb:
        atol = 3e-5 * max(1.0, float(np.abs(want_img).max()))
        if storage != "f32":     # an activation that rounded the other way moves a pixel's RGB by one storage step x its ToRGB weight
            atol += 3.0 * float(storage_ulp(np.abs(want).max(), storage)) * float(np.abs(tw).max())
        np.testing.assert_allclose(mem.get(img_out), want_img, rtol=0, atol=atol)
    return got
