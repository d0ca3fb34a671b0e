"""CPU restatement of the steps either side of the generator forward in the reference's
``scripts/demo.py``, at network resolution.  TEST INFRASTRUCTURE ONLY:
imported by tests/ (and nothing in the product).  Pinned to the reference's own ``preprocess()``
through tests/golden/prepost.npz (tests/golden/make_golden_prepost.py).

numpy float32 follows torch's CPU float32 arithmetic here (single IEEE operations in the same order).
"""
import numpy as np


def preprocess(img_u8: np.ndarray, mask_u8: np.ndarray) -> np.ndarray:
    """reference scripts/demo.py:56-66 without the PIL resizes (inputs already at network resolution).
    img_u8 [N,R,R,3] uint8, mask_u8 [N,R,R] uint8 -> x [N,4,R,R] float32."""
    img = np.asarray(img_u8)
    mask = (np.asarray(mask_u8)[:, :, :, np.newaxis] // 255)                      # :60
    imgf = img.astype(np.float32) * np.float32(2) / np.float32(255) - np.float32(1)  # :61
    maskf = mask.astype(np.float32)                                               # :62
    imgf = np.transpose(imgf, (0, 3, 1, 2))                                       # :63
    maskf = np.transpose(maskf, (0, 3, 1, 2))                                     # :64
    return np.concatenate([maskf - np.float32(0.5), imgf * maskf], axis=1)       # :65


def compose(y: np.ndarray, img_u8: np.ndarray, mask_u8: np.ndarray) -> np.ndarray:
    """reference scripts/demo.py:135-140 without the cv2 resize back to the original size.
    y [N,3,R,R] float32 -> composed uint8 [N,R,R,3]."""
    y = np.asarray(y, dtype=np.float32)
    r = np.clip(y * np.float32(0.5) + np.float32(0.5), np.float32(0), np.float32(1)) * np.float32(255)   # :135
    r = np.transpose(r.astype(np.uint8), (0, 2, 3, 1))                            # :136 (.to(torch.uint8) truncates)
    m = (np.asarray(mask_u8)[:, :, :, np.newaxis] // 255).astype(np.uint8)        # :139
    return np.asarray(img_u8) * m + r * (1 - m)                                   # :140
