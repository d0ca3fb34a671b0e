"""State-dict schema of the MI-GAN inference generator.

One table drives everything on the host side: the ``nn.Module`` tree of
``migan_inference.Generator`` (names and shapes identical to the reference so
``load_state_dict(strict=True)`` accepts its checkpoints), the synthetic weight
generator, the C-ABI weight binding, and the golden-schema test.

Reference: lib/model_zoo/migan_inference.py
  * channel rule ``min(ch_base // res, ch_max)``           :222-223, :342-343
  * encoder blocks b{R}..b8 (down=2) then b4 (down=1)       :217-233
  * synthesis b4 (no noise, no upsample) then b8..b{R}      :338-345
  * SeparableConv2d members conv1/conv2/downsample/upsample/noise  :122-150
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

CH_BASE = 32768
CH_MAX = 512
MIN_RESOLUTION = 8
MAX_RESOLUTION = 4096     # channels(4096) = 8.  Above 512 the layers have fewer than 64 channels and run a plain (untuned) kernel, fp32 storage
                          # only (round 6); the reference publishes no checkpoint above 512


def channels(res: int) -> int:
    """Feature width at spatial size ``res`` (reference :222-223, :342-343)."""
    return min(CH_BASE // res, CH_MAX)


def check_resolution(resolution: int) -> int:
    """Same failure mode as the reference: ValueError when not a power of two
    (:215-216, :330-331).  Returns log2(resolution)."""
    if not isinstance(resolution, int) or resolution <= 0:
        raise ValueError("resolution must be a positive power of two")
    log2 = resolution.bit_length() - 1
    if (1 << log2) != resolution:
        raise ValueError
    if resolution < MIN_RESOLUTION:
        # the reference constructs but cannot run below 8 (b4 has no fromrgb)
        raise ValueError("resolution must be >= 8")
    return log2


@dataclass(frozen=True)
class Entry:
    name: str                 # full state_dict key
    shape: Tuple[int, ...]
    kind: str                 # 'param' | 'buffer'
    role: str                 # dw_w dw_b pw_w rgb_w rgb_b fir_down fir_up filter_const noise_const noise_strength


@dataclass(frozen=True)
class SepConv:
    """One SeparableConv2d instance (reference :106-170)."""
    prefix: str               # e.g. 'encoder.b512.conv2'
    cin: int
    cout: int
    res_in: int
    res_out: int
    down: bool
    up: bool
    noise: bool


def _sepconv_entries(sc: SepConv) -> List[Entry]:
    p = sc.prefix
    out: List[Entry] = []
    # registration order of the reference: own params, own buffers, then children
    if sc.noise:
        out.append(Entry(f"{p}.noise_strength", (), "param", "noise_strength"))
        out.append(Entry(f"{p}.noise_const", (sc.res_out, sc.res_out), "buffer", "noise_const"))
    out.append(Entry(f"{p}.conv1.weight", (sc.cin, 1, 3, 3), "param", "dw_w"))
    out.append(Entry(f"{p}.conv1.bias", (sc.cin,), "param", "dw_b"))
    out.append(Entry(f"{p}.conv2.weight", (sc.cout, sc.cin, 1, 1), "param", "pw_w"))
    if sc.down:
        out.append(Entry(f"{p}.downsample.filter.weight", (sc.cin, 1, 4, 4), "param", "fir_down"))
    if sc.up:
        out.append(Entry(f"{p}.upsample.filter_const", (1, 1, sc.res_out, sc.res_out), "buffer", "filter_const"))
        out.append(Entry(f"{p}.upsample.filter.weight", (sc.cout, 1, 4, 4), "param", "fir_up"))
    return out


def encoder_res(resolution: int) -> List[int]:
    log2 = check_resolution(resolution)
    return [1 << i for i in range(log2, 1, -1)]          # R, R/2, ..., 4


def synthesis_res(resolution: int) -> List[int]:
    log2 = check_resolution(resolution)
    return [1 << i for i in range(2, log2 + 1)]          # 4, 8, ..., R


def sepconvs(resolution: int) -> List[SepConv]:
    """All SeparableConv2d layers in execution order (encoder then synthesis)."""
    layers: List[SepConv] = []
    for res in encoder_res(resolution):
        c = channels(res)
        if res > 4:
            cn = channels(res // 2)
            layers.append(SepConv(f"encoder.b{res}.conv1", c, c, res, res, False, False, False))
            layers.append(SepConv(f"encoder.b{res}.conv2", c, cn, res, res // 2, True, False, False))
        else:
            layers.append(SepConv("encoder.b4.conv1", c, c, 4, 4, False, False, False))
            layers.append(SepConv("encoder.b4.conv2", c, c, 4, 4, False, False, False))
    for res in synthesis_res(resolution):
        c = channels(res)
        if res == 4:
            layers.append(SepConv("synthesis.b4.conv1", c, c, 4, 4, False, False, False))
            layers.append(SepConv("synthesis.b4.conv2", c, c, 4, 4, False, False, False))
        else:
            cp = channels(res // 2)
            layers.append(SepConv(f"synthesis.b{res}.conv1", cp, c, res // 2, res, False, True, True))
            layers.append(SepConv(f"synthesis.b{res}.conv2", c, c, res, res, False, False, True))
    return layers


def entries(resolution: int) -> List[Entry]:
    """Every state_dict entry, in the reference's ``state_dict()`` order
    (Generator registers synthesis before encoder, reference :359-360)."""
    check_resolution(resolution)
    by_prefix = {sc.prefix: sc for sc in sepconvs(resolution)}
    out: List[Entry] = []
    for res in synthesis_res(resolution):
        c = channels(res)
        b = f"synthesis.b{res}"
        out += _sepconv_entries(by_prefix[f"{b}.conv1"])
        out += _sepconv_entries(by_prefix[f"{b}.conv2"])
        out.append(Entry(f"{b}.torgb.weight", (3, c, 1, 1), "param", "rgb_w"))
        out.append(Entry(f"{b}.torgb.bias", (3,), "param", "rgb_b"))
        if res > 4:
            out.append(Entry(f"{b}.upsample.filter_const", (1, 1, res, res), "buffer", "filter_const"))
            out.append(Entry(f"{b}.upsample.filter.weight", (3, 1, 4, 4), "param", "fir_up"))
    for res in encoder_res(resolution):
        c = channels(res)
        b = f"encoder.b{res}"
        if res == resolution:
            out.append(Entry(f"{b}.fromrgb.weight", (c, 4, 1, 1), "param", "rgb_w"))
            out.append(Entry(f"{b}.fromrgb.bias", (c,), "param", "rgb_b"))
        out += _sepconv_entries(by_prefix[f"{b}.conv1"])
        out += _sepconv_entries(by_prefix[f"{b}.conv2"])
    return out


# FIR taps: setup_filter([1,3,3,1], gain) (reference :31-55) gives
# outer(f,f)/64*gain; per axis that is [1,3,3,1]/8 (down, gain 1) and
# [1,3,3,1]/4 (up, gain 4).
FIR_TAPS = (1.0, 3.0, 3.0, 1.0)


def fir_kernel_2d(gain: float):
    """The 4x4 tap table the reference stores in ``filter.weight`` (:71-72, :95-96)."""
    return [[(a * b) / 64.0 * gain for b in FIR_TAPS] for a in FIR_TAPS]
