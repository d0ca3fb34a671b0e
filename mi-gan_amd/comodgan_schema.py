"""State-dict schema of the Co-Mod-GAN inference generator.

One table drives the ``nn.Module`` tree of ``comodgan.Generator`` (names and shapes identical to the
reference so ``load_state_dict(strict=True)`` accepts its checkpoints), the synthetic weight generator, the
C-ABI weight binding and the golden-schema test.

Reference: lib/model_zoo/comodgan.py (Encoder :114-204, Synthesis :346-420, Generator :423-455) and
lib/model_zoo/stylegan.py (dense :64-99, conv2d_layer :197-244, synthesis_layer :247-309,
torgb_layer :312-344, Mapping :356-439, synthesis_block :445-529, discrim_block :650-714).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple


@dataclass(frozen=True)
class Config:
    """Constructor arguments that fix the tensor shapes (reference defaults)."""
    resolution: int = 512
    ch_base: int = 32768      # comodgan.py:119,351
    ch_max: int = 512         # comodgan.py:120,352
    z_dim: int = 512          # stylegan.py:358
    w_dim: int = 512          # stylegan.py:360, comodgan.py:347
    w0_dim: int = 1024        # comodgan.py:118 (Encoder oc_n) == :348 (Synthesis w0_dim)
    map_layers: int = 8       # stylegan.py:362
    num_ws: int = 16          # scripts/demo.py:96,102 (14 at 256, 16 at 512)

    def channels(self, res: int) -> int:
        return min(self.ch_base // res, self.ch_max)

    @property
    def block_res(self) -> List[int]:
        out, r = [], 4
        while r <= self.resolution:
            out.append(r)
            r *= 2
        return out


def default_num_ws(resolution: int) -> int:
    """comodgan.py:367-370 (only 256 and 512 are defined there; other sizes follow 2*log2(R)-2)."""
    return 2 * (resolution.bit_length() - 1) - 2


def check_config(cfg: Config) -> None:
    r = cfg.resolution
    if not isinstance(r, int) or r <= 0 or (r & (r - 1)) != 0:
        raise ValueError                                 # comodgan.py:134-135,358-359
    if r < 8 or r > 512:
        raise ValueError("resolution must be in [8, 512]")
    for res in cfg.block_res:
        c = cfg.channels(res)
        if c % 64 != 0:
            raise ValueError(f"channel count {c} at resolution {res} must be a multiple of 64")


@dataclass(frozen=True)
class Entry:
    name: str
    shape: Tuple[int, ...]
    kind: str       # 'param' | 'buffer'
    role: str       # conv_w conv_b dense_w dense_b affine_w affine_b rgb_w rgb_b fir noise_const noise_strength w_avg


def entries(cfg: Config) -> List[Entry]:
    """state_dict entries in the reference's registration order."""
    e: List[Entry] = []
    P, B = "param", "buffer"
    # ---- mapping (stylegan.py:384-394): fc0..fc7 then the w_avg buffer
    feats = [cfg.z_dim] + [cfg.w_dim] * cfg.map_layers
    for i in range(cfg.map_layers):
        e.append(Entry(f"mapping.fc{i}.weight", (feats[i + 1], feats[i]), P, "dense_w"))
        e.append(Entry(f"mapping.fc{i}.bias", (feats[i + 1],), P, "dense_b"))
    e.append(Entry("mapping.w_avg", (cfg.w_dim,), B, "w_avg"))
    # ---- synthesis (registered before the encoder: Generator_StyleGan.__init__ runs first, comodgan.py:430)
    wl = cfg.w_dim + cfg.w0_dim
    c4 = cfg.channels(4)
    s = "synthesis.b4."
    e.append(Entry(s + "fc.weight", (c4 * 16, cfg.w0_dim), P, "dense_w"))
    e.append(Entry(s + "fc.bias", (c4 * 16,), P, "dense_b"))
    e.append(Entry(s + "conv.weight", (c4, c4, 3, 3), P, "conv_w"))
    e.append(Entry(s + "conv.bias", (c4,), P, "conv_b"))
    e.append(Entry(s + "conv.noise_strength", (), P, "noise_strength"))
    e.append(Entry(s + "conv.resample_filter", (4, 4), B, "fir"))
    e.append(Entry(s + "conv.noise_const", (4, 4), B, "noise_const"))
    e.append(Entry(s + "conv.affine.weight", (c4, wl), P, "affine_w"))
    e.append(Entry(s + "conv.affine.bias", (c4,), P, "affine_b"))
    e.append(Entry(s + "torgb.weight", (3, c4, 1, 1), P, "rgb_w"))
    e.append(Entry(s + "torgb.bias", (3,), P, "rgb_b"))
    e.append(Entry(s + "torgb.affine.weight", (c4, wl), P, "affine_w"))
    e.append(Entry(s + "torgb.affine.bias", (c4,), P, "affine_b"))
    for res in cfg.block_res[1:]:
        ci, co = cfg.channels(res // 2), cfg.channels(res)
        s = f"synthesis.b{res}."
        e.append(Entry(s + "resample_filter", (4, 4), B, "fir"))
        for name, cin, up in (("conv0", ci, True), ("conv1", co, False)):
            e.append(Entry(s + f"{name}.weight", (co, cin, 3, 3), P, "conv_w"))
            e.append(Entry(s + f"{name}.bias", (co,), P, "conv_b"))
            e.append(Entry(s + f"{name}.noise_strength", (), P, "noise_strength"))
            if up:
                e.append(Entry(s + f"{name}.resample_filter", (4, 4), B, "fir"))
            e.append(Entry(s + f"{name}.noise_const", (res, res), B, "noise_const"))
            e.append(Entry(s + f"{name}.affine.weight", (cin, wl), P, "affine_w"))
            e.append(Entry(s + f"{name}.affine.bias", (cin,), P, "affine_b"))
        e.append(Entry(s + "torgb.weight", (3, co, 1, 1), P, "rgb_w"))
        e.append(Entry(s + "torgb.bias", (3,), P, "rgb_b"))
        e.append(Entry(s + "torgb.affine.weight", (co, wl), P, "affine_w"))
        e.append(Entry(s + "torgb.affine.bias", (co,), P, "affine_b"))
    # ---- encoder (comodgan.py:143-190)
    first = True
    for res in reversed(cfg.block_res[1:]):
        c, cn = cfg.channels(res), cfg.channels(res // 2)
        s = f"encoder.b{res}."
        e.append(Entry(s + "resample_filter", (4, 4), B, "fir"))
        if first:
            e.append(Entry(s + "fromrgb.weight", (c, 4, 1, 1), P, "conv_w"))
            e.append(Entry(s + "fromrgb.bias", (c,), P, "conv_b"))
            first = False
        e.append(Entry(s + "conv0.weight", (c, c, 3, 3), P, "conv_w"))
        e.append(Entry(s + "conv0.bias", (c,), P, "conv_b"))
        e.append(Entry(s + "conv1.weight", (cn, c, 3, 3), P, "conv_w"))
        e.append(Entry(s + "conv1.bias", (cn,), P, "conv_b"))
        e.append(Entry(s + "conv1.resample_filter", (4, 4), B, "fir"))
    s = "encoder.b4."
    e.append(Entry(s + "conv.weight", (c4, c4, 3, 3), P, "conv_w"))
    e.append(Entry(s + "conv.bias", (c4,), P, "conv_b"))
    e.append(Entry(s + "fc.weight", (cfg.w0_dim, c4 * 16), P, "dense_w"))
    e.append(Entry(s + "fc.bias", (cfg.w0_dim,), P, "dense_b"))
    return e


def fir_kernel_2d() -> List[List[float]]:
    """upfirdn2d.setup_filter([1,3,3,1]): outer product / 64 (stored un-gained; the up path applies gain 4)."""
    k = [1.0, 3.0, 3.0, 1.0]
    return [[a * b / 64.0 for b in k] for a in k]
