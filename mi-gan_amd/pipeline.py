"""GPU mirrors of the steps either side of ``Generator.forward`` in the reference's scripts/demo.py
(preprocess :56-66 and the result / composition lines :135-140), at network resolution, through the
C ABI (``migan_pack_input`` / ``migan_compose_output``).  Like the generator there is no CPU path."""
from __future__ import annotations

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
