"""Drop-in for ``lib.model_zoo.migan_inference`` of Picsart-AI-Research/MI-GAN.

``Generator(resolution)`` has the reference's constructor, sub-module tree /
``state_dict`` schema (so ``model.load_state_dict(torch.load(path))`` from
scripts/demo.py:110 works unchanged) and ``forward(x)`` contract
(x: [N,4,R,R] = cat([mask-0.5, img*mask]) -> [N,3,R,R]; reference :355-369),
but ``forward`` is a single call into the MI355X HIP library through the C ABI
(include/migan_hip.h).  PyTorch is used for device memory and streams only.

There is no CPU or pure-PyTorch path here: a CPU tensor, a missing
libmigan_hip.so or a missing GPU raises.  Not supported (reference "out of
contract" list): autograd, torch.jit.trace / ONNX export.

Inject as ``sys.modules['lib.model_zoo.migan_inference']`` to run the reference
scripts unmodified.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import schema
from .hipbind import DTYPE_NAMES, MiganError, MiganHandle, MiganLib, dtype_code, load_library

# reference class of each node of the module tree, for repr() only
_NODE_KIND = {
    "synthesis": "Synthesis", "encoder": "Encoder", "conv1": "SeparableConv2d", "conv2": "SeparableConv2d",
    "fromrgb": "Conv2d", "torgb": "Conv2d", "downsample": "Downsample2d", "upsample": "Upsample2d", "filter": "Conv2d",
}


class _Node(nn.Module):
    """Parameter container mirroring one reference leaf module (fromrgb / torgb / conv1 / conv2 ``nn.Conv2d``, ``Downsample2d``,
    ``Upsample2d``): names and parameters only.  Their arithmetic is fused into the SeparableConv2d kernels."""

    def __init__(self, kind: str = "Module"):
        super().__init__()
        self._kind = kind

    def _get_name(self):
        return self._kind

    def forward(self, *args, **kwargs):
        raise NotImplementedError(
            f"{self._kind}: this leaf is fused into the SeparableConv2d kernels on the MI355X HIP path and cannot be called on its "
            "own; call the SeparableConv2d / block / Encoder / Synthesis / Generator that contains it")


def _dev_check(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: needs a tensor on an MI355X (HIP) device; there is no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{what}: Input type ({t.dtype}) and weight type (torch.float32) should be the same")


class _SepConv(_Node):
    """``SeparableConv2d`` (reference :106-170) as a callable sub-module: depthwise 3x3 + bias + lrelu_agc, [Downsample2d |
    Upsample2d], pointwise 1x1, + noise, lrelu_agc -- one ``migan_sepconv_forward`` call (include/migan_hip.h) on NHWC copies of the
    NCHW arguments.  The fused Generator.forward does not go through here; this is the reference's module-level API."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x)[0]

    def _run(self, x, skip=None, fromrgb=None, torgb=None, img_prev=None):
        _dev_check(x, "SeparableConv2d")
        lib = load_library()
        dw, pw = self.conv1, self.conv2                     # the depthwise and pointwise nn.Conv2d of the reference (:122-136)
        cin, cout = int(dw.weight.shape[0]), int(pw.weight.shape[0])
        down = 2 if getattr(self, "downsample", None) is not None else 1
        up = 2 if getattr(self, "upsample", None) is not None else 1
        n = int(x.shape[0])
        if fromrgb is not None:
            if x.dim() != 4 or x.shape[1] != 4:
                raise RuntimeError(f"expected the 4-channel network input [N, 4, H, W], got {list(x.shape)}")
            xin = x.contiguous()                                # NCHW: FromRGB reads the planes
        else:
            if x.dim() != 4 or x.shape[1] != cin:
                raise RuntimeError(f"Given groups={cin}, expected input[N, {cin}, H, W], got {list(x.shape)}")
            xin = x.permute(0, 2, 3, 1).contiguous()            # NHWC
        h, w = int(x.shape[2]), int(x.shape[3])
        if down == 2 and (h % 2 or w % 2):
            raise RuntimeError("Downsample2d needs even sizes")
        ho, wo = (h // 2, w // 2) if down == 2 else ((h * 2, w * 2) if up == 2 else (h, w))
        y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
        kw = dict(x=xin.data_ptr(), y=y.data_ptr(), conv1_weight=dw.weight.data_ptr(), conv1_bias=dw.bias.data_ptr(),
                  conv2_weight=pw.weight.data_ptr(), batch=n, cin=cin, cout=cout, res_in=h, width_in=w, down=down, up=up)
        keep = [xin]
        if getattr(self, "use_noise", False):
            nc = self.noise_const
            if tuple(nc.shape) != (ho, wo):
                raise RuntimeError(f"The size of noise_const {list(nc.shape)} must match the output size [{ho}, {wo}] (reference :166)")
            kw.update(noise_const=nc.data_ptr(), noise_strength=self.noise_strength.data_ptr())
        if skip is not None:
            _dev_check(skip, "SeparableConv2d skip")
            if tuple(skip.shape) != (n, cout, ho, wo):
                raise RuntimeError(f"The size of tensor a {[n, cout, ho, wo]} must match the size of tensor b {list(skip.shape)}")
            sk = skip.permute(0, 2, 3, 1).contiguous()
            keep.append(sk)
            kw.update(skip=sk.data_ptr())
        if fromrgb is not None:
            kw.update(fromrgb_weight=fromrgb.weight.data_ptr(), fromrgb_bias=fromrgb.bias.data_ptr())
        img_out = None
        if torgb is not None:
            img_out = torch.empty((n, 3, ho, wo), dtype=torch.float32, device=x.device)
            kw.update(torgb_weight=torgb.weight.data_ptr(), torgb_bias=torgb.bias.data_ptr(), img_out=img_out.data_ptr())
            if img_prev is not None:
                _dev_check(img_prev, "SynthesisBlock img")
                if tuple(img_prev.shape) != (n, 3, ho // 2, wo // 2):
                    raise RuntimeError(f"expected img [N, 3, {ho // 2}, {wo // 2}], got {list(img_prev.shape)}")
                ip = img_prev.contiguous()
                keep.append(ip)
                kw.update(img_prev=ip.data_ptr())
        if down == 2:
            scratch = torch.empty((n * ho * wo * cin,), dtype=torch.float32, device=x.device)
            keep.append(scratch)
            kw.update(scratch=scratch.data_ptr(), scratch_bytes=scratch.numel() * 4)
        wsplit = torch.empty((16 + 6 * cout * cin + 16,), dtype=torch.uint8, device=x.device)
        keep.append(wsplit)
        kw.update(wsplit=wsplit.data_ptr(), wsplit_bytes=wsplit.numel())
        with torch.cuda.device(x.device):
            lib.sepconv_forward(stream=int(torch.cuda.current_stream(x.device).cuda_stream), **kw)
        for t in keep:                                           # the launches are asynchronous: tie the temporaries to the stream
            t.record_stream(torch.cuda.current_stream(x.device))
        return y.permute(0, 3, 1, 2).contiguous(), img_out


class _EncoderBlock(_Node):
    """``EncoderBlock.forward(x, img)`` (reference :192-200) -> (x, feat)"""

    def forward(self, x, img):
        if self.fromrgb is not None:
            if x is not None:
                raise NotImplementedError("EncoderBlock with fromrgb: the HIP path fuses FromRGB into conv1 and takes x = None "
                                          "(the only way the reference's Encoder calls it, :236-241)")
            feat = self.conv1._run(img, fromrgb=self.fromrgb)[0]
        else:
            feat = self.conv1(x)
        return self.conv2(feat), feat


class _Encoder(_Node):
    """``Encoder.forward(img)`` (reference :235-246) -> (x, feats)"""

    def forward(self, img):
        x, feats = None, {}
        blocks = sorted(((int(n[1:]), m) for n, m in self.named_children()), reverse=True)
        for res, block in blocks:
            x, feat = block(x, img)
            feats[res] = feat
        return x, feats


class _SynthesisBlock(_Node):
    """``SynthesisBlockFirst.forward(x, enc_feat)`` (:271-280) / ``SynthesisBlock.forward(x, enc_feat, img)`` (:302-315) -> (x, img)"""

    def forward(self, x, enc_feat, img=None):
        x = self.conv1._run(x, skip=enc_feat)[0]                 # x = conv1(x); x = x + enc_feat
        if getattr(self, "torgb", None) is None:
            raise NotImplementedError("SynthesisBlock without torgb is not part of the inference generator")
        return self.conv2._run(x, torgb=self.torgb, img_prev=img)     # img = upsample(img) + torgb(x)


class _Synthesis(_Node):
    """``Synthesis.forward(x, enc_feats)`` (reference :347-352) -> img"""

    def forward(self, x, enc_feats):
        img = None
        for res, block in sorted((int(n[1:]), m) for n, m in self.named_children()):
            x, img = block(x, enc_feats[res]) if res == 4 else block(x, enc_feats[res], img)
        return img


def _make_node(parts) -> nn.Module:
    """the node class of module path `parts` (without the parameter name)"""
    depth, name = len(parts), parts[-1]
    if depth == 1:
        return _Encoder("Encoder") if name == "encoder" else _Synthesis("Synthesis")
    if depth == 2:
        if parts[0] == "encoder":
            return _EncoderBlock("EncoderBlock")
        return _SynthesisBlock("SynthesisBlockFirst" if name == "b4" else "SynthesisBlock")
    if depth == 3 and name in ("conv1", "conv2"):
        return _SepConv("SeparableConv2d")
    return _Node("Conv2d" if name in ("conv1", "conv2") else _NODE_KIND.get(name, "Module"))


def _init_tensor(e: schema.Entry) -> torch.Tensor:
    """Constructor-time values, same distributions as the reference's layers."""
    shp = e.shape
    if e.role in ("dw_w", "pw_w", "rgb_w"):
        w = torch.empty(shp)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))           # nn.Conv2d default, reference :122-136
        return w
    if e.role in ("dw_b", "rgb_b"):
        fan_in = {"dw_b": 9}.get(e.role)
        if fan_in is None:
            fan_in = 4 if "fromrgb" in e.name else schema.channels(int(e.name.split(".")[1][1:]))
        bound = 1.0 / math.sqrt(fan_in)
        return torch.empty(shp).uniform_(-bound, bound)
    if e.role == "noise_strength":
        return torch.zeros(())                                 # reference :150
    if e.role == "noise_const":
        return torch.randn(shp)                                # reference :149
    if e.role == "fir_down":
        return torch.tensor(schema.fir_kernel_2d(1.0)).repeat(shp[0], 1, 1, 1)    # reference :71-72
    if e.role == "fir_up":
        return torch.tensor(schema.fir_kernel_2d(4.0)).repeat(shp[0], 1, 1, 1)    # reference :95-96
    if e.role == "filter_const":
        w = torch.tensor([[1.0, 0.0], [0.0, 0.0]])
        return w.repeat(1, 1, shp[2] // 2, shp[3] // 2)        # reference :83-85
    raise AssertionError(e.role)


class Generator(nn.Module):
    """MI-GAN inference generator (reference migan_inference.py:355-369) on MI355X."""

    def __init__(self, resolution: int = 256, activation_dtype="f32", nan_policy: str = "propagate"):
        """resolution: as the reference (:356).  activation_dtype (extension): storage format of the feature maps between
        layers on the GPU -- "f32" (the reference's precision, <= 1e-3 parity), "bf16" (BASELINE configs[1]) or "f16";
        parameters, input, output and all arithmetic stay float32 (include/migan_hip.h, MIGAN_DTYPE_*).
        nan_policy (extension): "propagate" (default) -- a NaN activation stays a NaN through lrelu_agc's clamp, as Tensor.clamp does in
        the reference module (:21-23); "clamp" -- it leaves the clamp as -256, as in the reference's CUDA plugin (bias_act.cu:139): the
        -DMIGAN_NAN_CLAMP build of the library, ~2 % faster (no NaN test at all; finite inputs give the same bits either way)."""
        super().__init__()
        if nan_policy not in ("clamp", "propagate"):
            raise ValueError(f"nan_policy must be 'clamp' or 'propagate', got {nan_policy!r}")
        self._nan_policy = nan_policy
        schema.check_resolution(resolution)                    # ValueError like reference :215-216
        if resolution > schema.MAX_RESOLUTION:
            raise NotImplementedError(
                f"resolution {resolution}: feature width {schema.channels(resolution)} (reference :222-223); supported "
                f"resolutions are {schema.MIN_RESOLUTION}..{schema.MAX_RESOLUTION}")
        self.resolution = resolution
        self._act_dtype = dtype_code(activation_dtype)
        if resolution > 512 and self._act_dtype != 0:
            raise NotImplementedError("resolutions above 512 (layers with fewer than 64 channels: a plain kernel, csrc/migan_kernels.hpp "
                                      "narrow_sepconv_kernel) run with activation_dtype='f32' only")
        self._gemm: Optional[str] = None                       # None: the library default (f16x2-split MFMA)
        self._streams: Optional[int] = None
        self._frozen = False
        self._names: List[str] = []
        for e in schema.entries(resolution):
            node: nn.Module = self
            parts = e.name.split(".")
            for depth, p in enumerate(parts[:-1]):
                if not hasattr(node, p):
                    node.add_module(p, _make_node(parts[:depth + 1]))
                node = getattr(node, p)
            t = _init_tensor(e)
            if e.kind == "param":
                node.register_parameter(parts[-1], nn.Parameter(t))
            else:
                node.register_buffer(parts[-1], t)
            self._names.append(e.name)
        # attributes the reference's own tools read on the module tree (scripts/export_inference_model.py::copy_weights :33-83):
        # `fromrgb` is None on the encoder blocks that have none, the 1x1 convolutions carry `bias = None` (bias=False,
        # reference :130-136), every SeparableConv2d says whether it has noise
        for name, node in list(self.named_modules()):
            parts = name.split(".")
            if len(parts) == 2 and parts[0] == "encoder" and not hasattr(node, "fromrgb"):
                node.fromrgb = None
            if len(parts) == 3 and parts[2] in ("conv1", "conv2"):                 # a SeparableConv2d
                node.use_noise = hasattr(node, "noise_const")
                if not hasattr(node, "downsample"):
                    node.downsample = None
                if not hasattr(node, "upsample"):
                    node.upsample = None
                if hasattr(node, "conv2") and not hasattr(node.conv2, "bias"):
                    node.conv2.register_parameter("bias", None)
        # engine state (never part of state_dict)
        self._lib: Optional[MiganLib] = None
        self._handle: Optional[MiganHandle] = None
        self._handle_device: Optional[int] = None
        self._bound: Optional[Tuple[int, ...]] = None
        self._dirty = True
        self._ws: Optional[torch.Tensor] = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate())

    # ------------------------------------------------------------------ plumbing
    def _invalidate(self) -> None:
        self._dirty = True

    # ------------------------------------------------------------------ options of the MI355X engine (not in the reference)
    @property
    def activation_dtype(self) -> str:
        return DTYPE_NAMES[self._act_dtype]

    def set_activation_dtype(self, dtype) -> "Generator":
        """'f32' | 'bf16' | 'f16' (or torch.float32 / torch.bfloat16 / torch.float16): see __init__."""
        code = dtype_code(dtype)
        if code != self._act_dtype:
            self._act_dtype = code
            if self._handle is not None:
                self._handle.close()
            self._handle = None
            self._ws = None
        return self

    def set_gemm(self, variant: Optional[str]) -> "Generator":
        """How the 1x1 convolutions are multiplied.  fp32 storage: 'f16x2' (default: 3 fp16 MFMA products per fp32 product on
        scaled 2-way split operands), 'bf16x3' (6 bf16 products) or 'f32' (exact fp32 MFMA) -- all fp32-grade.  16-bit storage:
        'f16' (default: operands rounded to fp16, one MFMA per product) or 'f16x2'.  All accumulate in fp32 (include/migan_hip.h)."""
        self._gemm = variant
        if self._handle is not None:
            self._handle.set_gemm(variant or ("f16x2" if self._act_dtype == 0 else "f16"))
        return self

    def set_streams(self, n: int) -> "Generator":
        """n = 2 (default) .. 4: batches of >= 8 n images run as n staggered sub-batches on n HIP streams; 1: one stream."""
        self._streams = int(n)
        if self._handle is not None:
            self._handle.set_streams(self._streams)
        return self

    def freeze_weights(self, frozen: bool = True) -> "Generator":
        """Opt-in for repeated inference: promise that no parameter is modified in place from now on, so the per-forward
        preparation of the 1x1 weights (16-bit operand planes of the split GEMM, 0.13 ms of a 10 ms batch-32 forward) runs
        once.  load_state_dict / .to() / re-assignment of a parameter are picked up automatically; after an in-place write
        (``p.data.mul_()``, an optimizer step, a raw-pointer write) call ``freeze_weights()`` again, or
        ``freeze_weights(False)`` to go back to preparing them every forward (the default: in-place updates always seen)."""
        self._frozen = bool(frozen)
        if self._handle is not None:
            self._handle.assume_static_weights(self._frozen)
        return self

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._invalidate()
        return out

    def _tensors(self) -> List[torch.Tensor]:
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        return [sd[n] for n in self._names]

    def _require_device(self, x: torch.Tensor) -> int:
        if not x.is_cuda:
            raise RuntimeError(
                "mi-gan_amd Generator.forward needs a tensor on an MI355X (HIP) device; there is no CPU path. "
                "Move the model and input with .to('cuda').")
        return x.device.index if x.device.index is not None else torch.cuda.current_device()

    def _stream(self, x: torch.Tensor) -> int:
        return int(torch.cuda.current_stream(x.device).cuda_stream)

    def _engine(self, x: torch.Tensor) -> MiganHandle:
        dev = self._require_device(x)
        if self._lib is None:
            self._lib = load_library(nan_policy=self._nan_policy)      # raises MiganError when not built
        if self._handle is None or self._handle_device != dev:
            if self._handle is not None:
                self._handle.close()
            self._handle = MiganHandle(self._lib, self.resolution, dev, dtype=self._act_dtype)
            if self._gemm is not None:
                self._handle.set_gemm(self._gemm)
            if self._streams is not None:
                self._handle.set_streams(self._streams)
            self._handle.assume_static_weights(self._frozen)
            self._handle_device = dev
            self._bound = None
        tensors = self._tensors()
        ptrs = tuple(t.data_ptr() for t in tensors)
        if self._dirty or ptrs != self._bound:
            for name, t in zip(self._names, tensors):
                if t.device != x.device:
                    raise RuntimeError(
                        f"Expected all tensors to be on the same device, but {name} is on {t.device} and the input on {x.device}")
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise RuntimeError(f"{name}: parameters must be contiguous float32 (got {t.dtype})")
                self._handle.set_weight(name, t.data_ptr(), tuple(t.shape))
            self._handle.commit(self._stream(x))
            self._bound = ptrs
            self._dirty = False
        return self._handle

    def _workspace(self, h: MiganHandle, batch: int, device: torch.device, hw: Optional[Tuple[int, int]] = None) -> torch.Tensor:
        need = h.workspace_bytes(batch) if hw is None else h.workspace_bytes_hw(batch, hw[0], hw[1])
        if self._ws is None or self._ws.device != device or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def _check_input(self, x: torch.Tensor, any_size: bool = False) -> torch.Tensor:
        r = self.resolution
        if x.dim() != 4 or x.shape[1] != 4 or (not any_size and (x.shape[2] != r or x.shape[3] != r)):
            raise RuntimeError(f"expected input of shape [N, 4, {r}, {r}] (mask-0.5, img*mask), got {list(x.shape)}")
        if x.dtype != torch.float32:
            raise RuntimeError(f"Input type ({x.dtype}) and weight type (torch.float32) should be the same")
        if x.shape[0] == 0:
            raise RuntimeError("empty batch")
        return x.contiguous()

    # ------------------------------------------------------------------ API
    def forward(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Args: x: 4 channel rgb+mask [N,4,R,R]; returns img [N,3,R,R] (reference :362-369).
        out (extension): a contiguous float32 [N,3,R,R] tensor on x's device to write the image into instead of a fresh one -- e.g. this
        rank's slice of an all-gather's receive buffer (distributed.OutputGather), so that the collective runs in place."""
        x = self._check_input(x)
        h = self._engine(x)
        n = x.shape[0]
        ws = self._workspace(h, n, x.device)
        y = self._check_out(out, (n, 3, self.resolution, self.resolution), x.device)
        h.forward(x.data_ptr(), y.data_ptr(), n, ws.data_ptr(), ws.numel(), self._stream(x))
        return y

    def _check_out(self, out: Optional[torch.Tensor], shape, device) -> torch.Tensor:
        if out is None:
            return torch.empty(shape, dtype=torch.float32, device=device)
        if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or out.device != device or not out.is_contiguous():
            raise RuntimeError(f"out must be a contiguous float32 tensor of shape {list(shape)} on {device}, got {list(out.shape)} {out.dtype} on {out.device}")
        return out

    def sub_batches(self, batch: int, device=None) -> List[int]:
        """images per sub-batch of a forward of `batch` images on this model's handle (migan_forward_split): [batch] below 16 images or with
        set_streams(1), else two (up to four) staggered sub-batches"""
        if self._handle is None:
            if device is None:
                raise RuntimeError("sub_batches before the first forward needs the device")
            self._engine(torch.empty((1, 4, self.resolution, self.resolution), dtype=torch.float32, device=device))
        return self._handle.forward_split(batch)

    def forward_parts(self, x: torch.Tensor, outs: List[torch.Tensor], part_streams: List["torch.cuda.Stream"]) -> List[int]:
        """migan_forward_parts (extension, multi-GPU callers): the forward of x with the sub-batch hand-over made explicit.  Sub-batch k
        (sizes: sub_batches(N)) writes outs[k] ([n_k,3,R,R], contiguous float32) and, for k >= 1, runs on part_streams[k - 1]; the streams
        are NOT joined -- work enqueued on the current stream afterwards is ordered behind sub-batch 0 only, work on part_streams[k - 1]
        behind sub-batch k.  The caller joins them (current_stream.wait_stream(s)) before the next forward of this model."""
        x = self._check_input(x)
        h = self._engine(x)
        n = x.shape[0]
        sizes = h.forward_split(n)
        if len(outs) != len(sizes) or len(part_streams) < len(sizes) - 1:
            raise RuntimeError(f"a forward of {n} images runs as {len(sizes)} sub-batch(es) {sizes}: need that many outputs and one stream per sub-batch after the first")
        r = self.resolution
        ys = [self._check_out(o, (nk, 3, r, r), x.device) for o, nk in zip(outs, sizes)]
        ws = self._workspace(h, n, x.device)
        got = h.forward_parts(x.data_ptr(), [y.data_ptr() for y in ys], n, ws.data_ptr(), ws.numel(), self._stream(x),
                              [int(s.cuda_stream) for s in part_streams[:len(sizes) - 1]])
        assert got == sizes
        return sizes

    def forward_any_size(self, x: torch.Tensor) -> torch.Tensor:
        """Fully convolutional forward (reference README.md:87 asks for ``filter_const`` / ``noise_const`` to be made dynamic):
        x [N,4,H,W] with H, W multiples of resolution / 4 -> [N,3,H,W].  Each ``noise_const`` is tiled periodically and
        cropped to its layer's size; H = W = resolution is ``forward``."""
        x = self._check_input(x, any_size=True)
        h = self._engine(x)
        n, hh, ww = x.shape[0], int(x.shape[2]), int(x.shape[3])
        ws = self._workspace(h, n, x.device, (hh, ww))
        y = torch.empty((n, 3, hh, ww), dtype=torch.float32, device=x.device)
        h.forward_hw(x.data_ptr(), y.data_ptr(), n, hh, ww, ws.data_ptr(), ws.numel(), self._stream(x))
        return y

    def forward_uint8(self, img_u8: torch.Tensor, mask_u8: torch.Tensor) -> torch.Tensor:
        """scripts/demo.py:56-66 + forward + :135-140 in one call, at network resolution: uint8 image [N,R,R,3] (HWC) and
        uint8 mask [N,R,R] (255 = known pixel) -> composited uint8 image [N,R,R,3].  preprocess() runs inside the first
        kernel's tile builder and the post-processing + composition inside the last ToRGB epilogue: no fp32 network input or
        output tensor exists.  Bit-identical to pipeline.preprocess -> forward -> pipeline.compose."""
        r = self.resolution
        if not (img_u8.is_cuda and mask_u8.is_cuda):
            raise RuntimeError("mi-gan_amd Generator.forward_uint8 needs tensors on an MI355X (HIP) device; there is no CPU path")
        if img_u8.dtype != torch.uint8 or mask_u8.dtype != torch.uint8:
            raise RuntimeError("image and mask must be uint8 (np.array of the PIL images, reference demo.py:59-60)")
        if img_u8.dim() != 4 or tuple(img_u8.shape[1:]) != (r, r, 3) or tuple(mask_u8.shape) != tuple(img_u8.shape[:3]):
            raise RuntimeError(f"expected image [N,{r},{r},3] and mask [N,{r},{r}], got {list(img_u8.shape)} and {list(mask_u8.shape)}")
        if img_u8.shape[0] == 0:
            raise RuntimeError("empty batch")
        img_u8, mask_u8 = img_u8.contiguous(), mask_u8.contiguous()
        h = self._engine(img_u8)
        n = img_u8.shape[0]
        ws = self._workspace(h, n, img_u8.device)
        out = torch.empty((n, r, r, 3), dtype=torch.uint8, device=img_u8.device)
        h.forward_u8(img_u8.data_ptr(), mask_u8.data_ptr(), out.data_ptr(), n, ws.data_ptr(), ws.numel(), self._stream(img_u8))
        return out

    def forward_timed(self, x: torch.Tensor):
        """forward() with a hipEvent pair around every kernel launch: (y, [ms per launch])."""
        x = self._check_input(x)
        h = self._engine(x)
        n = x.shape[0]
        ws = self._workspace(h, n, x.device)
        y = torch.empty((n, 3, self.resolution, self.resolution), dtype=torch.float32, device=x.device)
        ms = h.forward_timed(x.data_ptr(), y.data_ptr(), n, ws.data_ptr(), ws.numel(), self._stream(x))
        return y, ms

    def launch_info(self):
        """Per-launch layer / kernel names and algorithmic flops and bytes per image."""
        if self._lib is None:
            self._lib = load_library(nan_policy=self._nan_policy)
        h = self._handle or MiganHandle(self._lib, self.resolution, 0, dtype=self._act_dtype)
        return h.launches()        # kernel names reflect the variants used by the last forward on this handle
