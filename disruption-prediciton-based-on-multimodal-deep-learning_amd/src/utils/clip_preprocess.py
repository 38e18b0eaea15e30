"""Device-side replacement for the tail of the reference's ``DatasetForVideo.get_video_data`` (src/dataset.py:124-144): centre
crop, BGR mean subtraction and the (T,H,W,C) -> (C,T,H,W) transpose, from uint8 frames already on the GPU (what ``cv2.imread``
returns, stacked), in one launch (``md_clip_preprocess``) - instead of fp32 clips assembled per sample on the host and copied
over.  The cv2 augmentations (:129-135) are host-side and random; they are not reproduced here.  SURVEY 8f item 3.
"""
import ctypes as C

import torch

from .. import _native as N
from .. import ops

BGR_MEAN = (90.0, 98.0, 102.0)          # src/dataset.py:205


def preprocess_clips(frames: torch.Tensor, crop_size: int, channels_last: bool = False, mean=BGR_MEAN) -> torch.Tensor:
    """frames (B, T, Hr, Wr, 3) uint8 on the GPU -> (B, 3, T, S, S) fp32, or with channels_last the kernels' own
    [B][T][S][S][4] layout (channel 3 zero)."""
    ops.require_cuda(frames)
    if frames.dtype != torch.uint8 or frames.dim() != 5 or frames.shape[-1] != 3:
        raise RuntimeError("mi355x hot path: preprocess_clips expects (B, T, H, W, 3) uint8 frames")
    frames = frames.contiguous()
    B, T, Hr, Wr, _ = frames.shape
    S = int(crop_size)
    out = torch.empty((B, T, S, S, 4) if channels_last else (B, 3, T, S, S), device=frames.device, dtype=torch.float32)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    N.check(N.lib().md_clip_preprocess(ops._p(frames), B, T, Hr, Wr, S, m, 1 if channels_last else 0, ops._p(out), ops._stream()),
            "md_clip_preprocess")
    return out


# ---- the six augmentations of DatasetForVideo (src/dataset.py:11-24 defaults, :129-135 order, :152-227 definitions) ------------
DEFAULT_AUGMENTATION_ARGS = {"bright_val": 10, "bright_p": 0.25, "contrast_min": 1, "contrast_max": 1.15, "contrast_p": 0.25, "blur_k": 5,
                             "blur_p": 0.25, "flip_p": 0.25, "vertical_ratio": 0.1, "vertical_p": 0.25, "horizontal_ratio": 0.1,
                             "horizontal_p": 0.25}
_SMALL_GAUSS = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
                7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}      # cv2.getGaussianKernel(k, sigma <= 0)


def gaussian_kernel(k: int):
    import math
    if k in _SMALL_GAUSS:
        return list(_SMALL_GAUSS[k])
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    w = [math.exp(-((i - (k - 1) * 0.5) ** 2) / (2 * sigma * sigma)) for i in range(k)]
    return [v / sum(w) for v in w]


def draw_augmentation(crop_size: int, args=None):
    """The decisions of ONE get_video_data call, drawn with the reference's own calls in the reference's order -- Python's
    ``random`` for the magnitudes, ``np.random`` for the coin flips (:209-228, 152-200) -- so a script that seeds both generators
    gets the reference's augmentation stream.  Returns the 10 integers md_clip_augment_preprocess takes per clip.  Literal
    behaviour kept: brightness with a non-positive draw ADDS |draw| and flips horizontally; ``randomflip`` flips twice (one draw,
    no effect); the two "shifts" zero an edge band (all of the clip when a positive ratio rounds to a zero shift)."""
    import random

    import numpy as np
    a = dict(DEFAULT_AUGMENTATION_ARGS if args is None else args)
    mode_b = bright = contrast = blur = 0
    alpha = 1
    lo_hi = [0, crop_size, 0, crop_size]
    b = int(random.uniform(-a["bright_val"], a["bright_val"]))
    if np.random.random() < a["bright_p"]:
        mode_b, bright = (1, b) if b > 0 else (2, -b)
    if np.random.random() < a["contrast_p"]:
        contrast, alpha = 1, int(random.uniform(a["contrast_min"], a["contrast_max"]))
    if np.random.random() < a["blur_p"]:
        blur = 1
    np.random.random() < a["flip_p"]
    for k, (ratio, prob) in enumerate((("vertical_ratio", "vertical_p"), ("horizontal_ratio", "horizontal_p"))):
        if np.random.random() < a[prob]:
            r = random.uniform(-a[ratio], a[ratio])
            s = int(crop_size * r)
            lo_hi[2 * k], lo_hi[2 * k + 1] = (0, crop_size - s if s else 0) if r > 0 else (-s, crop_size)
    return [mode_b, bright, contrast, alpha, blur, int(a["blur_k"])] + lo_hi


def augment_preprocess_clips(frames: torch.Tensor, crop_size: int, params, channels_last: bool = False, mean=BGR_MEAN) -> torch.Tensor:
    """preprocess_clips with the augmentations applied between crop and mean subtraction, in ONE launch.  frames (B, T, Hr, Wr, 3)
    uint8 on the GPU; params: one draw_augmentation() result per clip (all clips must use the same blur kernel size)."""
    ops.require_cuda(frames)
    if frames.dtype != torch.uint8 or frames.dim() != 5 or frames.shape[-1] != 3:
        raise RuntimeError("mi355x hot path: augment_preprocess_clips expects (B, T, H, W, 3) uint8 frames")
    frames = frames.contiguous()
    B, T, Hr, Wr, _ = frames.shape
    if len(params) != B or any(len(p) != 10 for p in params) or len({p[5] for p in params}) != 1:
        raise RuntimeError("mi355x hot path: one 10-integer augmentation record per clip, one blur kernel size per batch")
    S = int(crop_size)
    out = torch.empty((B, T, S, S, 4) if channels_last else (B, 3, T, S, S), device=frames.device, dtype=torch.float32)
    par = torch.tensor(params, dtype=torch.int32).to(frames.device)
    gk = torch.tensor(gaussian_kernel(int(params[0][5])), dtype=torch.float32).to(frames.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    N.check(N.lib().md_clip_augment_preprocess(ops._p(frames), B, T, Hr, Wr, S, m, 1 if channels_last else 0, ops._p(par), ops._p(gk),
                                               ops._p(out), ops._stream()), "md_clip_augment_preprocess")
    return out
