#!/usr/bin/env python3
"""Golden vectors that pin the 16-bit ACTIVATION-STORAGE modes (BASELINE configs[1]: migan-256, bf16) to the REFERENCE module.

The storage modes of the HIP library keep every parameter, the network input / output, the running RGB image and all arithmetic in
fp32 and round a feature map once, when it is stored: every SeparableConv2d output of the encoder, the synthesis `conv1` output after
the skip add, every synthesis `conv2` output (oracle/migan_oracle.py::round_storage).  Until round 6 that definition was only held to
itself (oracle(bf16) against oracle(fp32)).  Here the REFERENCE module (`lib/model_zoo/migan_inference.py::Generator`, imported from
/root/reference, unmodified) is run with forward hooks that do exactly that rounding on ITS tensors:

  * forward hook on every encoder SeparableConv2d (reference :198-199)            -> output rounded to the storage format
  * forward PRE-hook on every synthesis `conv2` (its input is `conv1(x) + enc_feat`, reference :272 / :305)  -> input rounded
  * forward hook on every synthesis `conv2` (reference :273 / :306)               -> output rounded
  * GEMM variant "f16" (the default of the 16-bit modes): forward pre-hook on every 1x1 `conv2` Conv2d of a SeparableConv2d that rounds
    its input to fp16 after the exact scaling by 2^7, and the 1x1 weights rounded to fp16 after the exact per-tensor power-of-two
    scaling (oracle/migan_oracle.py::round_gemm_operands) before they are loaded into the reference module.

All arithmetic is the reference's own (torch-CPU); the hooks only round.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bf16.py

Outputs (committed): tests/golden/storage_<mode>_r<R>[_x2].npz : full output `y` (strided for R = 256) + checksums + per-layer tap
summaries + the fp32 output of the same module without hooks (so the fixture also states the mode's quantisation envelope as the
REFERENCE sees it).
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("MIGAN_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

pkg = importlib.import_module("mi-gan_amd")
synth = pkg.synth
import lib.model_zoo.migan_inference as ref  # noqa: E402  (the reference module itself)

torch.set_num_threads(max(1, os.cpu_count() or 1))


def round_to(t: torch.Tensor, storage: str) -> torch.Tensor:
    """round-to-nearest-even to the storage format and back (torch's own conversions)"""
    if storage == "bf16":
        return t.to(torch.bfloat16).to(torch.float32)
    if storage == "f16":
        return t.to(torch.float16).to(torch.float32)
    raise ValueError(storage)


def round_gemm_input(t: torch.Tensor) -> torch.Tensor:
    return (t * 128.0).to(torch.float16).to(torch.float32) / 128.0


def round_gemm_weight(w: np.ndarray) -> np.ndarray:
    m = float(np.abs(w).max())
    if not (m > 0 and np.isfinite(m)):
        return w
    e = min(max(int(np.floor(np.log2(m))), -100), 100)
    s = np.float32(2.0 ** (13 - e))
    return ((w.astype(np.float32) * s).astype(np.float16).astype(np.float32) / s).astype(np.float32)


def tap_summary(t: torch.Tensor) -> np.ndarray:
    a = t.detach().double().numpy().ravel()
    idx = (np.arange(13, dtype=np.int64) * 2654435761 + 12345) % a.size
    return np.concatenate([[a.mean(), a.std(), np.abs(a).max()], a[idx]]).astype(np.float64)


def storage_case(tag, resolution, batch, seed, storage, gemm16, stride=1):
    sd_np = synth.make_state_dict(resolution, seed=seed, regime="export")
    x = synth.make_input(batch, resolution, seed=seed)

    def build(rounded_weights):
        g = ref.Generator(resolution=resolution)
        sd = {}
        for k, v in sd_np.items():
            is_pw = k.endswith(".conv2.weight") and v.ndim == 4 and v.shape[2] == 1          # the 1x1 of a SeparableConv2d
            sd[k] = torch.from_numpy((round_gemm_weight(v) if (rounded_weights and is_pw) else v).copy())
        g.load_state_dict(sd, strict=True)
        return g.eval()

    with torch.no_grad():
        y32 = build(False)(torch.from_numpy(x.copy())).numpy()

    g = build(gemm16)
    taps, hooks = {}, []
    for name, mod in g.named_modules():
        if not isinstance(mod, ref.SeparableConv2d):
            continue
        enc = name.startswith("encoder.")
        is_conv2 = name.endswith(".conv2")
        if gemm16:
            hooks.append(mod.conv2.register_forward_pre_hook(lambda m, inp: (round_gemm_input(inp[0]),)))
        if enc or is_conv2:
            def out_hook(m, inp, out, name=name):
                o = round_to(out, storage)
                taps[name] = tap_summary(o)
                return o
            hooks.append(mod.register_forward_hook(out_hook))
        if not enc and is_conv2:
            def in_hook(m, inp, name=name):
                i = round_to(inp[0], storage)
                taps[name[:-1] + "1.skip"] = tap_summary(i)           # "synthesis.bR.conv1.skip": conv1 output + enc_feat, as stored
                return (i,)
            hooks.append(mod.register_forward_pre_hook(in_hook))
    with torch.no_grad():
        y = g(torch.from_numpy(x.copy())).numpy()
    for h in hooks:
        h.remove()
    d = {
        "resolution": np.int64(resolution), "batch": np.int64(batch), "seed": np.int64(seed), "stride": np.int64(stride),
        "storage": np.array(storage), "gemm16": np.int64(1 if gemm16 else 0),
        "y": y[:, :, ::stride, ::stride].copy(),
        "y_sum": y.astype(np.float64).sum(axis=(2, 3)),
        "y_abs_sum": np.abs(y.astype(np.float64)).sum(axis=(2, 3)),
        "y_absmax": np.float64(np.abs(y).max()),
        "y_f32": y32[:, :, ::stride, ::stride].copy(),
        "envelope": np.float64(np.abs(y - y32).max()),          # the mode's quantisation noise on THIS input, measured on the reference module
        "envelope_rms": np.float64(np.sqrt(np.mean((y.astype(np.float64) - y32) ** 2))),
    }
    for k, v in taps.items():
        d["tap/" + k] = v
    np.savez_compressed(os.path.join(HERE, f"storage_{tag}.npz"), **d)
    print(f"storage_{tag}.npz  R={resolution} N={batch} {storage} gemm16={gemm16} |y|max={d['y_absmax']:.4f} "
          f"envelope max {d['envelope']:.4e} rms {d['envelope_rms']:.4e} taps={len(taps)}")


if __name__ == "__main__":
    storage_case("bf16_r64", 64, 2, 33, "bf16", True)
    storage_case("bf16_r64_x2", 64, 2, 33, "bf16", False)            # the same storage mode with exact (f16x2-split) GEMM operands
    storage_case("f16_r64", 64, 2, 33, "f16", True)
    storage_case("bf16_r256", 256, 2, 33, "bf16", True, stride=2)   # BASELINE configs[1]'s model
    storage_case("bf16_r256_x2", 256, 2, 33, "bf16", False, stride=2)
