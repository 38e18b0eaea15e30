"""Oracle (test infrastructure only): the six video augmentations of the reference's DatasetForVideo (src/dataset.py:129-135,
152-227) restated literally, quirks included, on one cropped clip (T, S, S, 3) float32 holding 0..255 pixel values:

  brightness      (:209-221)  ``bright = int(random.uniform(-val, val))`` is drawn ALWAYS, then ``np.random.random() < p`` decides;
                              bright > 0: clip(frame + bright, 10, 255); otherwise frame - bright (adds |bright|, no clip) AND a
                              horizontal flip
  contrast        (:223-228)  alpha = int(random.uniform(min, max)) -- 1 for the default (1, 1.15) -- and cv2.convertScaleAbs:
                              |alpha x| rounded half to even, saturated to 0..255
  blur            (:196-200)  cv2.GaussianBlur(frame, (k, k), 0) per frame and channel: separable, sigma from the kernel size
                              (k <= 7 with sigma 0: OpenCV's fixed tables), BORDER_REFLECT_101
  randomflip      (:152-159)  flips every frame TWICE: the identity (consumes one random number)
  vertical_shift / horizontal_shift (:161-194)  not a shift: ratio > 0 keeps the first S - to_shift rows (columns) and zeroes the
                              rest -- everything when to_shift == 0, since ``[:-0]`` is empty --, ratio <= 0 zeroes the first
                              |to_shift| rows (columns)

``draw`` reproduces the reference's random-number calls in their order (Python's ``random`` for magnitudes, ``np.random`` for the
coin flips), so seeding both generators as the reference's scripts do gives the same decisions.
Pinned by tests/golden/eval_curve.npz (aug/*: recorded from the reference's own methods with seeded generators) for everything
that does not need OpenCV -- draw order, both brightness branches, both masks; ``cv2.flip`` was stood in for by a NumPy flip.
PARITY UNPINNED for convertScaleAbs and GaussianBlur: OpenCV is not installed, these two follow its documented behaviour."""
import random

import numpy as np

SMALL_GAUSS = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
               7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}     # cv::getGaussianKernel, sigma <= 0


def gaussian_kernel(k: int) -> np.ndarray:
    if k in SMALL_GAUSS:
        return np.array(SMALL_GAUSS[k], dtype=np.float32)
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    x = np.arange(k, dtype=np.float64) - (k - 1) * 0.5
    w = np.exp(-(x * x) / (2 * sigma * sigma))
    return (w / w.sum()).astype(np.float32)


def draw(args: dict, crop_size: int) -> dict:
    """The decisions of one get_video_data call (:129-135), in the reference's order of random-number calls."""
    p = {"mode_b": 0, "bright": 0, "contrast": 0, "alpha": 1, "blur": 0, "ksize": int(args["blur_k"]), "row_lo": 0, "row_hi": crop_size,
         "col_lo": 0, "col_hi": crop_size}
    bright = int(random.uniform(-args["bright_val"], args["bright_val"]))
    if np.random.random() < args["bright_p"]:
        p["mode_b"], p["bright"] = (1, bright) if bright > 0 else (2, -bright)
    if np.random.random() < args["contrast_p"]:
        p["contrast"], p["alpha"] = 1, int(random.uniform(args["contrast_min"], args["contrast_max"]))
    if np.random.random() < args["blur_p"]:
        p["blur"] = 1
    np.random.random() < args["flip_p"]                      # randomflip: one draw, no effect
    for lo, hi, ratio, prob in (("row_lo", "row_hi", "vertical_ratio", "vertical_p"), ("col_lo", "col_hi", "horizontal_ratio", "horizontal_p")):
        if np.random.random() < args[prob]:
            r = random.uniform(-args[ratio], args[ratio])
            to_shift = int(crop_size * r)
            if r > 0:
                p[lo], p[hi] = 0, (crop_size - to_shift if to_shift else 0)
            else:
                p[lo], p[hi] = -to_shift, crop_size
    return p


def _round_half_even_u8(x: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(x), 0, 255).astype(np.float32)


def apply(clip: np.ndarray, p: dict) -> np.ndarray:
    """clip (T, S, S, 3) float32 -> augmented (T, S, S, 3) float32 (before normalisation)."""
    v = clip.astype(np.float32).copy()
    if p["mode_b"] == 1:
        v = np.clip(v + np.float32(p["bright"]), 10, 255).astype(np.float32)
    elif p["mode_b"] == 2:
        v = (v + np.float32(p["bright"]))[:, :, ::-1, :].copy()
    if p["contrast"]:
        v = _round_half_even_u8(np.abs(v * np.float32(p["alpha"])))
    if p["blur"]:
        k = gaussian_kernel(p["ksize"]); h = p["ksize"] // 2
        S = v.shape[1]
        idx = np.arange(-h, S + h)
        idx = np.where(idx < 0, -idx, idx); idx = np.where(idx >= S, 2 * S - 2 - idx, idx)          # BORDER_REFLECT_101
        tmp = np.zeros_like(v)
        for j in range(p["ksize"]):                      # row pass, left to right, float32 multiply then add
            tmp = (tmp + (k[j] * v[:, :, idx[j:j + S], :]).astype(np.float32)).astype(np.float32)
        out = np.zeros_like(v)
        for i in range(p["ksize"]):
            out = (out + (k[i] * tmp[:, idx[i:i + S], :, :]).astype(np.float32)).astype(np.float32)
        v = out
    m = np.zeros(v.shape[1:3], dtype=bool)
    m[p["row_lo"]:p["row_hi"], p["col_lo"]:p["col_hi"]] = True
    # the reference applies the row mask, then the column mask: the intersection survives
    return np.where(m[None, :, :, None], v, np.float32(0)).astype(np.float32)
