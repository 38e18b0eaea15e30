"""Oracle (test infrastructure only): CPU restatement of the reference's inference / evaluation path, SURVEY 8(f) item 4.

  * ``synth_frames``            deterministic uint8 frame stack (integer arithmetic only; stands in for what cv2.imread returns)
  * ``video_windows``           VideoDataset.__init__/load_frames/get_video_data  (src/utils/utility.py:371-473): which frames
                                window ``idx`` reads, centre crop, BGR mean subtraction, (T,H,W,C) -> (C,T,H,W)
  * ``series_windows``          DatasetFor0D (src/utils/utility.py:475-513)
  * ``assemble_video_curve``    generate_prob_curve's post-processing           (src/utils/utility.py:950-961)
  * ``assemble_0D_curve``       generate_prob_curve_from_0D's post-processing   (src/utils/utility.py:1040-1057)
  * ``multi_window_tables``     MultiModalDataset.__init__ index matching       (src/utils/utility.py:579-611), literal
  * ``assemble_multi_curve``    generate_prob_curve_from_multi's post-processing (src/utils/utility.py:1133-1170)
  * ``moving_average``          moving_avarage_smoothing                        (src/utils/utility.py:872-893)
  * ``threshold_predictions``   evaluate's decision rule                        (src/evaluate.py:56-58, 75-76)

Pinned by tests/golden/eval_curve.npz, recorded from the reference itself (tests/golden/make_eval_golden.py).
"""
from __future__ import annotations

import numpy as np

BGR_MEAN = np.array([90.0, 98.0, 102.0], dtype=np.float32)      # utility.py:441
FPS = 210                                                        # utility.py:951, 1041


def synth_frames(n_frames: int, seed: int, size: int = 256) -> np.ndarray:
    """(n_frames, size, size, 3) uint8: a hashed texture plus a slow brightness drift, so that consecutive windows differ."""
    f = np.arange(n_frames, dtype=np.uint64).reshape(-1, 1, 1, 1)
    y = np.arange(size, dtype=np.uint64).reshape(1, -1, 1, 1)
    x = np.arange(size, dtype=np.uint64).reshape(1, 1, -1, 1)
    c = np.arange(3, dtype=np.uint64).reshape(1, 1, 1, -1)
    h = (f * np.uint64(7349) + (y // np.uint64(8)) * np.uint64(977) + (x // np.uint64(8)) * np.uint64(613) + c * np.uint64(101)
         + np.uint64(seed) * np.uint64(7919))
    h = (h * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    tex = ((h >> np.uint64(13)) & np.uint64(0x7F)).astype(np.int64)                  # 0..127
    drift = (64 + 60 * np.sin(np.arange(n_frames) * 0.21 + seed)).astype(np.int64).reshape(-1, 1, 1, 1)
    return np.clip(tex + drift, 0, 255).astype(np.uint8)


def video_window_count(n_frames: int, seq_len: int, dist: int, frame_srt: int, frame_end: int) -> int:
    n_paths = len(range(n_frames)[frame_srt:frame_end + 210])                         # :399
    return max(0, n_paths - seq_len - dist)                                           # :402


def video_windows(frames: np.ndarray, seq_len: int, dist: int, frame_srt: int, frame_end: int, crop: int):
    """Yields the fp32 (3, seq_len, crop, crop) clip of every window in order (:404-440)."""
    paths = frames[frame_srt:frame_end + 210]
    H, W = frames.shape[1:3]
    my, mx, o = H // 2, W // 2, crop // 2
    for idx in range(max(0, len(paths) - seq_len - dist)):
        buf = paths[idx + 1: idx + seq_len + 1].astype(np.float32)                    # :408
        while buf.shape[0] < seq_len:                                                 # refill_temporal_slide
            buf = np.concatenate((buf, buf[-1:]))
        buf = buf[:, my - o:my + o, mx - o:mx + o, :] - BGR_MEAN.reshape(1, 1, 1, 3)
        yield np.ascontiguousarray(buf.transpose(3, 0, 1, 2))


def series_windows(values: np.ndarray, seq_len: int, dist: int):
    """values (n_rows, n_cols), already scaled: window idx = rows idx+1 .. idx+seq_len (:507-511)."""
    for idx in range(max(0, values.shape[0] - seq_len - dist)):
        yield values[idx + 1: idx + seq_len + 1]


def _startup_correction(p, limit):                                                    # :954-958, :1045-1048
    p = list(p)
    for i, v in enumerate(p):
        if i < limit and v >= 0.5:
            p[i] = 0
    return p


def assemble_video_curve(probs, clip_len: int, frame_srt: int):
    p = [0] * (clip_len + frame_srt) + list(probs)[1:-1]                              # :953
    p = _startup_correction(p, FPS * 1)
    return np.arange(0, len(p)) * (1 / FPS) * 1, p                                    # :961


def moving_average(X: np.ndarray, k: int, method: str = "backward") -> np.ndarray:
    S = np.zeros(X.shape[0])
    hw = k // 2
    for t in range(X.shape[0]):
        if method == "backward":
            S[t] = np.mean(X[:t + 1]) if t < k else np.sum(X[t - k:t]) / k
        elif t < hw:
            S[t] = np.mean(X[:t + 1])
        elif t < X.shape[0] - hw:
            S[t] = np.mean(X[t - hw:t + hw])
        else:
            S[t] = np.mean(X[t - hw:])
    return np.clip(S, 0, 1)


def assemble_0D_curve(probs, seq_len: int, t_start: float):
    interval = 4                                                                      # :1040
    frame_srt = int(t_start * FPS / interval)
    p = [0] * (frame_srt + seq_len) + list(probs)[1:] + [0] * seq_len                 # :1043
    p = _startup_correction(p, FPS * 1)
    n = len(p)
    px = np.linspace(0, n, num=n, endpoint=True) * (interval / FPS)                   # :1050
    q = np.interp(np.linspace(0, n * interval, num=n * interval, endpoint=True) * (1 / FPS), px, np.array(p, dtype=np.float64))
    q = moving_average(q, 12)                                                         # :1054
    return np.arange(0, len(q)) * (1 / FPS), q


def threshold_predictions(p_disrupt: np.ndarray, threshold: float) -> np.ndarray:
    """evaluate.py:57-58: label 1 (normal) unless softmax[:,0] > threshold."""
    return np.logical_not(p_disrupt > np.float32(threshold)).astype(np.int64)


def multi_window_tables(n_frames: int, ts_time: np.ndarray, frame_srt: int, frame_end: int, t_srt: float, t_end: float,
                        vis_seq_len: int, ts_seq_len: int, dt: float, tau: int):
    """(frame index list per window, last 0D row per window) exactly as MultiModalDataset builds them (:579-611), including
    the second length match, which truncates by the length of the FIRST-stage list (``ts_indices``, :606), not by the filtered one."""
    video_indices = [i for i in reversed(range(frame_end, frame_srt, -tau))]                            # :580
    ts_idx_end = len(ts_time) - int(np.sum(ts_time > t_end))                                            # :583
    ts_idx_start = int(t_srt / dt)                                                                      # :584
    ts_indices = [i for i in reversed(range(ts_idx_end, ts_idx_start, -tau))]                           # :586
    if len(video_indices) > len(ts_indices):
        video_indices = video_indices[-len(ts_indices):]
    elif len(video_indices) < len(ts_indices):
        ts_indices = ts_indices[-len(video_indices):]
    paths = list(range(n_frames))
    video_frames = [paths[idx + 1: idx - tau * vis_seq_len + 1: -tau][::-1] for idx in video_indices if idx > vis_seq_len * tau]   # :596-598
    ts_sel = [idx for idx in ts_indices if idx > ts_seq_len * tau]                                       # :600-602
    if len(video_frames) > len(ts_sel):
        video_frames = video_frames[-len(ts_indices):]                                                   # :605 (sic)
    elif len(video_frames) < len(ts_sel):
        ts_sel = ts_sel[-len(video_frames):]
    return video_frames, ts_sel


def multi_ts_window(values: np.ndarray, idx_end: int, ts_seq_len: int, tau: int) -> np.ndarray:       # :680-684
    idx_srt = idx_end - ts_seq_len * tau
    return values[idx_srt + 1: idx_end + 1][::tau, :]


def _interp_extrap(x, xp, fp):
    """scipy interp1d(kind='linear', fill_value='extrapolate') on increasing xp."""
    x, xp, fp = np.asarray(x, np.float64), np.asarray(xp, np.float64), np.asarray(fp, np.float64)
    i = np.clip(np.searchsorted(xp, x, side="right") - 1, 0, len(xp) - 2)
    slope = (fp[i + 1] - fp[i]) / (xp[i + 1] - xp[i])
    return fp[i] + slope * (x - xp[i])


def assemble_multi_curve(probs, t_srt: float, t_end: float, tau: int):
    """t_srt / t_end: times of the first / last selected 0D row (:1133-1134).  Returns (time_x, smoothed curve); the reference
    function itself returns (time_x, the RAW per-window probabilities)."""
    dt_end, interval = 1.0, tau
    n0, n1 = int(t_srt * FPS / interval), int(dt_end * FPS / interval)
    total = [0] * n0 + list(probs)[1:] + [0] * n1                                                        # :1140
    total = _startup_correction(total, FPS * 1.0 / interval)
    x_srt = [i * interval / FPS for i in range(0, n0)]
    x_prob = [x_srt[-1] + (i + 1) * 1 / FPS * interval for i in range(0, len(list(probs)[1:]) + n1)]
    q = _interp_extrap(np.linspace(0, t_end + dt_end, num=len(total) * interval, endpoint=True), np.array(x_srt + x_prob), np.array(total, dtype=np.float64))
    q = moving_average(q, 16, "center")                                                                  # :1160
    return np.linspace(0, t_end + dt_end, num=len(q), endpoint=True), q
