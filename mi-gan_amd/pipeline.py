"""GPU mirrors of the steps either side of ``Generator.forward`` in the reference's scripts/demo.py
(preprocess :56-66 and the result / composition lines :135-140), at network resolution, through the
C ABI (``migan_pack_input`` / ``migan_compose_output``).  Like the generator there is no CPU path."""
from __future__ import annotations

import math

import torch

from .hipbind import load_library


def _check(img_u8: torch.Tensor, mask_u8: torch.Tensor):
    if not (img_u8.is_cuda and mask_u8.is_cuda):
        raise RuntimeError("mi-gan_amd.pipeline needs tensors on an MI355X (HIP) device; there is no CPU path")
    if img_u8.dtype != torch.uint8 or mask_u8.dtype != torch.uint8:
        raise RuntimeError("image and mask must be uint8 (np.array of the PIL images, reference demo.py:59-60)")
    if img_u8.dim() != 4 or img_u8.shape[-1] != 3 or img_u8.shape[1] != img_u8.shape[2]:
        raise RuntimeError(f"expected image batch [N,R,R,3], got {list(img_u8.shape)}")
    if tuple(mask_u8.shape) != tuple(img_u8.shape[:3]):
        raise RuntimeError(f"expected mask batch {list(img_u8.shape[:3])}, got {list(mask_u8.shape)}")
    return img_u8.contiguous(), mask_u8.contiguous()


def preprocess(img_u8: torch.Tensor, mask_u8: torch.Tensor) -> torch.Tensor:
    """uint8 image [N,R,R,3] + mask [N,R,R] (255 = known) -> x [N,4,R,R] = cat([mask-0.5, img*mask])."""
    img_u8, mask_u8 = _check(img_u8, mask_u8)
    n, r = img_u8.shape[0], img_u8.shape[1]
    x = torch.empty((n, 4, r, r), dtype=torch.float32, device=img_u8.device)
    load_library().pack_input(img_u8.data_ptr(), mask_u8.data_ptr(), x.data_ptr(), n, r,
                              int(torch.cuda.current_stream(img_u8.device).cuda_stream))
    return x


def compose(y: torch.Tensor, img_u8: torch.Tensor, mask_u8: torch.Tensor) -> torch.Tensor:
    """network output y [N,3,R,R] -> uint8 [N,R,R,3]: known pixels from the image, holes from the network."""
    img_u8, mask_u8 = _check(img_u8, mask_u8)
    n, r = img_u8.shape[0], img_u8.shape[1]
    if not y.is_cuda or y.dtype != torch.float32 or tuple(y.shape) != (n, 3, r, r):
        raise RuntimeError(f"expected y [N,3,R,R] float32 on the GPU, got {list(y.shape)} {y.dtype}")
    out = torch.empty((n, r, r, 3), dtype=torch.uint8, device=img_u8.device)
    load_library().compose_output(y.contiguous().data_ptr(), img_u8.data_ptr(), mask_u8.data_ptr(), out.data_ptr(), n, r,
                                  int(torch.cuda.current_stream(img_u8.device).cuda_stream))
    return out


class MIGAN_Pipeline(torch.nn.Module):
    """The reference's deployed pipeline, scripts/create_onnx_pipeline.py::MIGAN_Pipeline (:118-264, exported there as
    migan_pipeline_v2.onnx), on the GPU through the C ABI: masked bounding box -> crop -> resize to the network resolution ->
    generator -> resize back -> feathered blend into the image (``migan_pipeline_bbox / _pre / _post`` either side of
    ``migan_forward``).  Same constructor and ``forward(image, mask)`` contract as the reference module:

      image (1, 3, H, W) uint8, mask (1, 1, H, W) uint8 (255 = known pixel); the image is modified in place and returned.

    ``model_path`` is a reference ``migan_*.pt`` state dict, or an already built ``mi-gan_amd`` Generator.  A mask of another size is
    resized to the image's size first (nearest), like the reference's first line (:256).  No CPU path."""

    def __init__(self, model_path, resolution: int, padding: int = 128, device="cuda"):
        super().__init__()
        from .migan_inference import Generator
        if isinstance(model_path, torch.nn.Module):
            self.model = model_path
        else:
            self.model = Generator(resolution=resolution)
            self.model.load_state_dict(torch.load(model_path, map_location="cpu"))
        self.model = self.model.to(device).eval()
        self.res = int(resolution)
        self.padding = int(padding)
        # GaussianSmoothing(channels=1, kernel_size=5, sigma=1.0, dim=2).weight (:63-85), in torch fp32 like the reference buffer
        ax = torch.arange(5, dtype=torch.float32)
        g = 1 / (1.0 * math.sqrt(2 * math.pi)) * torch.exp(-((ax - 2.0) / (2 * 1.0)) ** 2)
        k = g[:, None] * g[None, :]
        self.gaussian_weight = (k / k.sum()).contiguous()          # host side: handed to migan_pipeline_post by value
        self._gauss = self.gaussian_weight.flatten().tolist()
        self._scratch = None

    def _scratch_for(self, lib, h: int, w: int, device) -> torch.Tensor:
        """scratch of the bbox / post kernels, one buffer per (device, stream): two streams running the pipeline never share it"""
        need = lib.pipeline_scratch_bytes(h, w)
        key = (device.index if device.index is not None else torch.cuda.current_device(), int(torch.cuda.current_stream(device).cuda_stream))
        if self._scratch is None:
            self._scratch = {}
        buf = self._scratch.get(key)
        if buf is None or buf.numel() < need:
            if len(self._scratch) >= 8:                    # (stream handles come and go: keep the cache bounded)
                self._scratch.clear()
            buf = torch.empty(need, dtype=torch.uint8, device=device)
            self._scratch[key] = buf
        return buf

    @staticmethod
    def _check_mask(mask: torch.Tensor) -> torch.Tensor:
        if not mask.is_cuda:
            raise RuntimeError("mi-gan_amd.pipeline needs tensors on an MI355X (HIP) device; there is no CPU path")
        if mask.dtype != torch.uint8:
            raise RuntimeError("mask must be uint8 (reference create_onnx_pipeline.py:254-255)")
        if mask.dim() != 4 or mask.shape[0] != 1 or mask.shape[1] != 1:
            raise RuntimeError(f"expected mask (1, 1, h, w), got {list(mask.shape)}")
        return mask.contiguous()

    def get_masked_bbox(self, mask: torch.Tensor):
        """(:132-231) -> x_min, x_max, y_min, y_max"""
        mask = self._check_mask(mask)
        lib = load_library()
        h, w = int(mask.shape[-2]), int(mask.shape[-1])
        with torch.cuda.device(mask.device):                      # the handle-free entry points launch on the CURRENT device
            scratch = self._scratch_for(lib, h, w, mask.device)
            return lib.pipeline_bbox(mask.data_ptr(), h, w, self.res, self.padding, scratch.data_ptr(),
                                     int(torch.cuda.current_stream(mask.device).cuda_stream))

    @torch.no_grad()
    def forward(self, image: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        if not (image.is_cuda and mask.is_cuda):
            raise RuntimeError("mi-gan_amd.pipeline needs tensors on an MI355X (HIP) device; there is no CPU path")
        if image.dtype != torch.uint8 or mask.dtype != torch.uint8:
            raise RuntimeError("image and mask must be uint8 (reference create_onnx_pipeline.py:254-255)")
        if image.dim() != 4 or image.shape[0] != 1 or image.shape[1] != 3 or not image.is_contiguous():
            raise RuntimeError(f"expected a contiguous image (1, 3, H, W), got {list(image.shape)}")
        h, w = int(image.shape[2]), int(image.shape[3])
        mask = self._check_mask(mask)
        if mask.device != image.device:
            raise RuntimeError("image and mask must be on the same device")
        with torch.cuda.device(image.device):                     # the handle-free entry points launch on the CURRENT device
            return self._forward_on_device(image, mask, h, w)

    def _forward_on_device(self, image: torch.Tensor, mask: torch.Tensor, h: int, w: int) -> torch.Tensor:
        lib = load_library()
        stream = int(torch.cuda.current_stream(image.device).cuda_stream)
        if tuple(mask.shape[2:]) != (h, w):                      # mask = tvF.resize(mask, image size, NEAREST) (:256)
            resized = torch.empty((1, 1, h, w), dtype=torch.uint8, device=image.device)
            lib.pipeline_mask_resize(mask.data_ptr(), int(mask.shape[2]), int(mask.shape[3]), resized.data_ptr(), h, w, stream)
            mask = resized
        scratch = self._scratch_for(lib, h, w, image.device)
        bbox = lib.pipeline_bbox(mask.data_ptr(), h, w, self.res, self.padding, scratch.data_ptr(), stream)
        x = torch.empty((1, 4, self.res, self.res), dtype=torch.float32, device=image.device)
        lib.pipeline_pre(image.data_ptr(), mask.data_ptr(), h, w, bbox, self.res, x.data_ptr(), stream)
        y = self.model(x).contiguous()
        lib.pipeline_post(image.data_ptr(), mask.data_ptr(), h, w, bbox, self.res, y.data_ptr(), scratch.data_ptr(),
                          gauss25=self._gauss, stream=stream)
        return image
