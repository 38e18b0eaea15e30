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
