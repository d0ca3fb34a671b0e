"""Drop-in for ``lib.model_zoo.comodgan`` of Picsart-AI-Research/MI-GAN.

``Generator(Mapping(num_ws), Encoder(resolution), Synthesis(resolution))`` -- the way the reference's
scripts/demo.py:95-106 assembles ``comodgan-256|512`` -- keeps the reference's constructors, sub-module tree /
``state_dict`` schema (``load_state_dict(torch.load(path))``, demo.py:110) and ``forward`` contract
(comodgan.py:435-455), but ``Generator.forward`` is one call into the MI355X HIP library through the C ABI
(include/comodgan_hip.h).  PyTorch is used for device memory, streams and drawing ``z`` / the per-pixel noise of
``noise_mode='random'`` only.  There is no CPU or pure-PyTorch path: a CPU tensor, a missing libmigan_hip.so or
a missing GPU raises.  Not supported: autograd, ``c`` (class conditioning: c_dim = 0 in every reference config),
``truncation_cutoff``, ``return_intermediate_outs``.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import comodgan_schema as cs
from .hipbind import CoModGANHandle, MiganLib, load_library


version = '3'          # reference comodgan.py:10-11 (lib/model_zoo/__init__.py imports `version` from this module)
symbol = 'comodgan'


class _Node(nn.Module):
    def __init__(self, kind: str = "Module"):
        super().__init__()
        self._kind = kind

    def _get_name(self):
        return self._kind

    def forward(self, *args, **kwargs):
        raise NotImplementedError(f"{self._kind}: only comodgan.Generator.forward is implemented on the MI355X HIP path; "
                                  "sub-modules hold the reference-named parameters")


def _init_tensor(e: cs.Entry) -> torch.Tensor:
    """Constructor-time values of the reference layers (stylegan.py:79-80,213-215,273,275-276,393)."""
    if e.role in ("conv_w", "rgb_w", "affine_w"):
        return torch.randn(e.shape)
    if e.role == "dense_w":
        return torch.randn(e.shape) / (0.01 if e.name.startswith("mapping.") else 1.0)
    if e.role in ("conv_b", "rgb_b", "dense_b", "w_avg"):
        return torch.zeros(e.shape)
    if e.role == "affine_b":
        return torch.ones(e.shape)
    if e.role == "noise_strength":
        return torch.zeros(())
    if e.role == "noise_const":
        return torch.randn(e.shape)
    if e.role == "fir":
        return torch.tensor(cs.fir_kernel_2d())
    raise AssertionError(e.role)


def _populate(root: nn.Module, prefix: str, cfg: cs.Config) -> None:
    for e in cs.entries(cfg):
        if not e.name.startswith(prefix + "."):
            continue
        node: nn.Module = root
        parts = e.name[len(prefix) + 1:].split(".")
        for p in parts[:-1]:
            if not hasattr(node, p):
                node.add_module(p, _Node(p))
            node = getattr(node, p)
        t = _init_tensor(e)
        if e.kind == "param":
            node.register_parameter(parts[-1], nn.Parameter(t))
        else:
            node.register_buffer(parts[-1], t)


class Mapping(_Node):
    """stylegan.py:356-439 (c_dim = 0).  Parameter container; evaluated inside Generator.forward."""

    def __init__(self, z_dim: int = 512, c_dim: int = 0, w_dim: int = 512, num_ws: int = 14, num_layers: int = 8, **unused):
        super().__init__("Mapping")
        if c_dim:
            raise NotImplementedError("class-conditional mapping (c_dim > 0) is not part of the inference path")
        self.z_dim, self.c_dim, self.w_dim, self.num_ws, self.num_layers = z_dim, c_dim, w_dim, num_ws, num_layers
        _populate(self, "mapping", cs.Config(resolution=8, z_dim=z_dim, w_dim=w_dim, map_layers=num_layers, num_ws=num_ws))


class Encoder(_Node):
    """comodgan.py:114-204."""

    def __init__(self, resolution: int = 256, ic_n: int = 4, oc_n: int = 1024, ch_base: int = 32768, ch_max: int = 512, **unused):
        super().__init__("Encoder")
        if ic_n != 4:
            raise NotImplementedError("the inference path takes 4 input channels (mask, rgb)")
        cfg = cs.Config(resolution=resolution, ch_base=ch_base, ch_max=ch_max, w0_dim=oc_n)
        cs.check_config(cfg)                                     # ValueError like comodgan.py:134-135
        self.resolution, self.ic_n, self.oc_n, self.ch_base, self.ch_max = resolution, ic_n, oc_n, ch_base, ch_max
        _populate(self, "encoder", cfg)


class Synthesis(_Node):
    """comodgan.py:346-420."""

    def __init__(self, w_dim: int = 512, w0_dim: int = 1024, resolution: int = 256, rgb_n: int = 3, ch_base: int = 32768,
                 ch_max: int = 512, **unused):
        super().__init__("Synthesis")
        if rgb_n != 3:
            raise NotImplementedError("rgb_n must be 3")
        cfg = cs.Config(resolution=resolution, ch_base=ch_base, ch_max=ch_max, w_dim=w_dim, w0_dim=w0_dim)
        cs.check_config(cfg)                                     # ValueError like comodgan.py:358-359
        self.w_dim, self.w0_dim, self.resolution, self.rgb_n, self.ch_base, self.ch_max = w_dim, w0_dim, resolution, rgb_n, ch_base, ch_max
        self.num_ws = cs.default_num_ws(resolution)              # comodgan.py:367-370 (14 at 256, 16 at 512)
        _populate(self, "synthesis", cfg)


class Generator(nn.Module):
    """Co-Mod-GAN generator (reference comodgan.py:423-455) on MI355X."""

    def __init__(self, mapping: Mapping, encoder: Encoder, synthesis: Synthesis):
        super().__init__()
        if synthesis.num_ws != mapping.num_ws:
            raise ValueError                                     # stylegan.py:606-607
        if (encoder.resolution, encoder.ch_base, encoder.ch_max, encoder.oc_n) != (
                synthesis.resolution, synthesis.ch_base, synthesis.ch_max, synthesis.w0_dim) or mapping.w_dim != synthesis.w_dim:
            raise ValueError("encoder and synthesis geometries differ")
        self.mapping, self.synthesis, self.encoder = mapping, synthesis, encoder
        self.num_ws, self.z_dim, self.c_dim, self.w_dim = mapping.num_ws, mapping.z_dim, mapping.c_dim, mapping.w_dim
        self.img_resolution, self.img_channels, self.ic_n = synthesis.resolution, synthesis.rgb_n, encoder.ic_n
        self._cfg = cs.Config(resolution=synthesis.resolution, ch_base=synthesis.ch_base, ch_max=synthesis.ch_max, z_dim=mapping.z_dim,
                              w_dim=mapping.w_dim, w0_dim=synthesis.w0_dim, map_layers=mapping.num_layers, num_ws=mapping.num_ws)
        self._names: List[str] = [e.name for e in cs.entries(self._cfg)]
        self._lib: Optional[MiganLib] = None
        self._handle: Optional[CoModGANHandle] = None
        self._handle_device: Optional[int] = None
        self._bound: Optional[Tuple[int, ...]] = None
        self._dirty = True
        self._ws: Optional[torch.Tensor] = None
        self._frozen = False         # freeze_weights(): the caller's promise that parameters no longer change in place
        self._refreeze = True        # the handle has not been told yet
        self._cutoff: Optional[int] = None   # truncation_cutoff the handle was last told
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate())

    # ------------------------------------------------------------------ plumbing
    def freeze_weights(self, frozen: bool = True) -> "Generator":
        """Opt-in for repeated inference: promise that no parameter is modified in place from now on, so the per-forward
        weight preparation (0.45 ms of a 19 ms batch-16 forward) runs once.  load_state_dict / .to() / re-assignment of a
        parameter are picked up automatically; after an in-place write (``p.data.mul_()``, an optimizer step, a raw-pointer
        write) call ``freeze_weights()`` again, or ``freeze_weights(False)`` to go back to re-preparing every forward."""
        self._frozen = bool(frozen)
        self._refreeze = True
        return self

    def _invalidate(self) -> None:
        self._dirty = True

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._invalidate()
        return out

    def _tensors(self) -> List[torch.Tensor]:
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        return [sd[n] for n in self._names]

    def _stream(self, x: torch.Tensor) -> int:
        return int(torch.cuda.current_stream(x.device).cuda_stream)

    def _engine(self, x: torch.Tensor) -> CoModGANHandle:
        if not x.is_cuda:
            raise RuntimeError("mi-gan_amd comodgan.Generator.forward needs a tensor on an MI355X (HIP) device; there is no CPU path. "
                               "Move the model and input with .to('cuda').")
        dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
        if self._lib is None:
            self._lib = load_library()                           # raises MiganError when not built
        if self._handle is None or self._handle_device != dev:
            if self._handle is not None:
                self._handle.close()
            c = self._cfg
            self._handle = CoModGANHandle(self._lib, c.resolution, c.num_ws, c.ch_base, c.ch_max, c.z_dim, c.w_dim, c.w0_dim, c.map_layers, dev)
            self._handle_device = dev
            self._bound = None
            self._refreeze = True
            self._cutoff = None
        tensors = self._tensors()
        ptrs = tuple(t.data_ptr() for t in tensors)
        if self._dirty or ptrs != self._bound:
            for name, t in zip(self._names, tensors):
                if t.device != x.device:
                    raise RuntimeError(f"Expected all tensors to be on the same device, but {name} is on {t.device} and the input on {x.device}")
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise RuntimeError(f"{name}: parameters must be contiguous float32 (got {t.dtype})")
                self._handle.set_weight(name, t.data_ptr(), tuple(t.shape))
            self._handle.commit(self._stream(x))
            self._bound = ptrs
            self._dirty = False
        return self._handle

    def _workspace(self, h: CoModGANHandle, batch: int, device: torch.device) -> torch.Tensor:
        need = h.workspace_bytes(batch)
        if self._ws is None or self._ws.device != device or self._ws.numel() < need:
            self._ws = None
            self._refreeze = True            # a new allocation holds no prepared weight planes (even at a recycled address)
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    # ------------------------------------------------------------------ API
    def forward(self, x: torch.Tensor, z: Optional[torch.Tensor] = None, c=None, truncation_psi: float = 1, truncation_cutoff=None,
                noise_mode: str = "random", return_intermediate_outs: bool = False, _timed: bool = False):
        """Args: x: 4 channel rgb+mask [N,4,R,R] (comodgan.py:437-441); z: [N,z_dim] (drawn with torch.randn when None, :438-439)."""
        assert noise_mode in ["random", "const", "none"]         # stylegan.py:280
        if c is not None or return_intermediate_outs:
            raise NotImplementedError("c (class conditioning: c_dim = 0 in every published config) and return_intermediate_outs (a training-"
                                      "loss hook) are not part of the MI355X inference path")
        if truncation_cutoff is not None and (int(truncation_cutoff) != truncation_cutoff or truncation_cutoff < 0):
            raise ValueError(f"truncation_cutoff must be a non-negative integer or None, got {truncation_cutoff!r}")
        r = self.img_resolution
        if x.dim() != 4 or x.shape[1] != 4 or x.shape[2] != r or x.shape[3] != r:
            raise RuntimeError(f"expected input of shape [N, 4, {r}, {r}] (mask-0.5, img*mask), got {list(x.shape)}")
        if x.dtype != torch.float32:
            raise RuntimeError(f"Input type ({x.dtype}) and weight type (torch.float32) should be the same")
        n = x.shape[0]
        if n == 0:
            raise RuntimeError("empty batch")
        h = self._engine(x)
        x = x.contiguous()
        if z is None:
            z = torch.randn([n, self.z_dim]).to(x.device)        # comodgan.py:439
        if z.shape != (n, self.z_dim):
            raise RuntimeError(f"expected z of shape [{n}, {self.z_dim}], got {list(z.shape)}")
        z = z.to(device=x.device, dtype=torch.float32).contiguous()
        noise = None
        if noise_mode == "random":
            noise = torch.randn(n * h.noise_floats(), dtype=torch.float32, device=x.device)    # stylegan.py:284-285, all layers at once
        cutoff = None if truncation_cutoff is None else int(truncation_cutoff)
        if cutoff != self._cutoff:                               # (part of the workspace layout: set before sizing it)
            h.set_truncation_cutoff(cutoff)
            self._cutoff = cutoff
        ws = self._workspace(h, n, x.device)
        # freeze_weights(): the fp16 operand planes / demodulation statistics of the 3x3 weights at the head of the workspace
        # are prepared once and reused (the library re-prepares them by itself after a re-binding, on another workspace or
        # another stream).  Without it they are rebuilt every forward, so in-place parameter updates are always seen.
        if self._refreeze:
            h.assume_static_weights(self._frozen)          # (re-)asserting drops the planes prepared so far
            self._refreeze = False
        y = torch.empty((n, 3, r, r), dtype=torch.float32, device=x.device)
        ms = h.forward(x.data_ptr(), z.data_ptr(), y.data_ptr(), n, ws.data_ptr(), ws.numel(), float(truncation_psi), noise_mode,
                       None if noise is None else noise.data_ptr(), self._stream(x), timed=_timed)
        return (y, ms) if _timed else y

    def forward_timed(self, x: torch.Tensor, z: torch.Tensor, noise_mode: str = "const"):
        """forward() with a hipEvent pair around every kernel launch: (y, [ms per launch])."""
        return self.forward(x, z, noise_mode=noise_mode, _timed=True)

    def launch_info(self):
        if self._handle is None:
            raise RuntimeError("launch_info() needs one forward first")
        return self._handle.launches()
