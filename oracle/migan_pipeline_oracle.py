"""CPU restatement of the reference's DEPLOYED pre/post-processing, scripts/create_onnx_pipeline.py::MIGAN_Pipeline (:118-264;
SURVEY section 8f row N2, second half).  TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing in the product.  Pinned to
outputs of the reference module itself through tests/golden/pipeline_*.npz (tests/golden/make_golden_pipeline.py).

torch CPU ops throughout, in the reference's order.  `tv_resize` restates the one torchvision function the reference calls
(transforms.functional.resize on a TENSOR, torchvision 0.9: float32 cast, F.interpolate without antialias, round + cast back).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def tv_resize(img: torch.Tensor, size, mode: str) -> torch.Tensor:
    out_dtype = img.dtype
    need_cast = out_dtype not in (torch.float32, torch.float64)
    x = img.to(torch.float32) if need_cast else img
    x = F.interpolate(x, size=[int(s) for s in size], mode=mode, align_corners=False if mode == "bilinear" else None)
    if need_cast:
        x = torch.round(x).to(out_dtype)
    return x


def gaussian_kernel(kernel_size: int = 5, sigma: float = 1.0) -> torch.Tensor:
    """GaussianSmoothing.__init__ (:63-85): product of per-axis 1 / (std sqrt(2 pi)) exp(-((x - mean) / (2 std))^2), normalised"""
    kernel = 1
    grids = torch.meshgrid([torch.arange(kernel_size, dtype=torch.float32) for _ in range(2)], indexing="ij")
    for mgrid in grids:
        mean = (kernel_size - 1) / 2
        kernel = kernel * (1 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-((mgrid - mean) / (2 * sigma)) ** 2))
    return kernel / torch.sum(kernel)


def masked_bbox(mask_u8: np.ndarray, resolution: int, padding: int):
    """get_masked_bbox (:132-231) in plain integers.  mask_u8 [H,W]; returns x_min, x_max, y_min, y_max."""
    h, w = mask_u8.shape
    m = torch.from_numpy(np.ascontiguousarray(mask_u8)).to(torch.float32)
    xs = torch.nonzero(m.mean(dim=0) < 255.0).reshape(-1).tolist()          # :144-147
    ys = torch.nonzero(m.mean(dim=1) < 255.0).reshape(-1).tolist()
    x_min, x_max = min(xs + [w]), max(xs + [0])                              # :149-152
    y_min, y_max = min(ys + [h]), max(ys + [0])
    x_min = min(x_min, x_max); x_max = max(x_min, x_max)                      # :154-172
    y_min = min(y_min, y_max); y_max = max(y_min, y_max)
    cnt_x, cnt_y = (x_min + x_max) // 2, (y_min + y_max) // 2                # :174-175
    crop = max(x_max - x_min, y_max - y_min) + 2 * padding                   # :177-180
    crop = max(crop, resolution)                                             # :181-184
    off = crop // 2                                                          # :186
    x_min, x_max = max(cnt_x - off, 0), min(cnt_x + off, w)                  # :187-202
    y_min, y_max = max(cnt_y - off, 0), min(cnt_y + off, h)
    xe, ye = max(crop - (x_max - x_min), 0), max(crop - (y_max - y_min), 0)  # :204-211
    x_min, x_max = max(x_min - xe, 0), min(x_max + xe, w)                    # :213-229
    y_min, y_max = max(y_min - ye, 0), min(y_max + ye, h)
    return x_min, x_max, y_min, y_max


def preprocess(image: torch.Tensor, mask: torch.Tensor, resolution: int) -> torch.Tensor:
    """:233-239; image [1,3,h,w] uint8, mask [1,1,h,w] uint8 (the crop) -> x [1,4,R,R] float32"""
    image = tv_resize(image, (resolution, resolution), "bilinear")
    mask = tv_resize(mask, (resolution, resolution), "nearest")
    image = image.to(torch.float32) * 2 / 255 - 1
    mask = mask.to(torch.float32) / 255
    return torch.cat([mask - 0.5, image * mask], dim=1)


def postprocess(image: torch.Tensor, mask: torch.Tensor, model_output: torch.Tensor) -> torch.Tensor:
    """:241-250"""
    out = ((model_output * 0.5 + 0.5) * 255).clamp(0, 255)
    out = tv_resize(out, (image.size(2), image.size(3)), "bilinear")
    image = image.to(torch.float32)
    mask = mask.to(torch.float32)
    mask = F.max_pool2d(mask, 3, stride=1, padding=1)
    k = gaussian_kernel().view(1, 1, 5, 5)
    mask = F.conv2d(F.pad(mask, (2, 2, 2, 2), mode="reflect"), weight=k, groups=1, padding="valid")     # GaussianSmoothing.forward :106-115
    mask = mask / torch.tensor(255)
    composed = image * mask + out * (1 - mask)
    return composed.clamp(0, 255).to(torch.uint8)


def pipeline(image_u8: np.ndarray, mask_u8: np.ndarray, generator, resolution: int, padding: int = 128):
    """MIGAN_Pipeline.forward (:252-264).  image_u8 [3,H,W], mask_u8 [1,H,W]; `generator`: x [1,4,R,R] float32 -> y [1,3,R,R].
    Returns (result image uint8 [3,H,W], bbox, x)."""
    image = torch.from_numpy(np.array(image_u8, copy=True))[None]
    mask = torch.from_numpy(np.array(mask_u8, copy=True))[None]
    x_min, x_max, y_min, y_max = masked_bbox(mask_u8[0], resolution, padding)
    ci, cm = image[:, :, y_min:y_max, x_min:x_max], mask[:, :, y_min:y_max, x_min:x_max]
    x = preprocess(ci, cm, resolution)
    y = generator(x)
    image[:, :, y_min:y_max, x_min:x_max] = postprocess(ci, cm, y)
    return image[0].numpy(), (x_min, x_max, y_min, y_max), x.numpy()
