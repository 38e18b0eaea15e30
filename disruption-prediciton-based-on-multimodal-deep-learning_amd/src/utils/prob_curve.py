"""Inference path of the reference's ``src/utils/utility.py``: sliding-window disruption-probability curves for one shot
(``generate_prob_curve`` :896-977, ``generate_prob_curve_from_0D`` :979-1066, ``generate_prob_curve_from_multi`` :1068-1178),
``moving_avarage_smoothing`` (:872-893) and
``measure_computation_time`` (:1201-1230), MI355X-first.  SURVEY 8(f) item 4.

The reference builds every window on the host (cv2.imread of seq_len files per window, fp32, crop, normalise, transpose),
copies it over and synchronises on ``.cpu()`` after each B=1 forward.  Here the shot's frames are put into HBM ONCE as uint8
(F, 256, 256, 3); a window is a view of seq_len consecutive frames that ``md_clip_preprocess`` crops, mean-subtracts and
transposes in one launch; softmax column 0 and the arg-max are written into per-shot device buffers and read back once at
the end.  ``windows_per_launch=1`` is the reference's B=1 streaming order; larger values put that many overlapping windows
into one forward (evaluation-mode BatchNorm uses running statistics, so windows do not interact).

The post-processing (zero padding of the start-up phase, the ``p >= 0.5`` start-up correction, the time axis, interpolation and
smoothing of the 0D curve) is host arithmetic on a few hundred numbers and follows the reference line by line.  The two
matplotlib figures (plot_exp_prob_type_1/2) are presentation and are not produced; ``save_dir`` is accepted and ignored.
"""
from __future__ import annotations

import time
from typing import List, Optional, Tuple

import numpy as np
import torch

from .. import ops
import os

from .clip_preprocess import preprocess_clips
from .graphed import graphed_forward

_GRAPH = os.environ.get("MD_GRAPH_STEP") == "1"

FPS = 210


def moving_avarage_smoothing(X: np.ndarray, k: int, method: str = "backward") -> np.ndarray:
    """utility.py:872-893 (name as in the reference).  Prefix sums instead of a mean per element."""
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    c = np.concatenate(([0.0], np.cumsum(X)))
    t = np.arange(n)
    if method == "backward":
        lo, hi, den = np.where(t < k, 0, t - k), np.where(t < k, t + 1, t), np.where(t < k, t + 1, k)
    else:
        hw = k // 2
        head, tail = t < hw, t >= n - hw
        lo = np.where(head, 0, t - hw)
        hi = np.where(head, t + 1, np.where(tail, n, t + hw))
        den = hi - lo
    with np.errstate(invalid="ignore", divide="ignore"):
        S = (c[hi] - c[lo]) / den
    return np.clip(S, 0, 1)


def video_window_count(n_frames: int, seq_len: int, dist: int, frame_srt: int, frame_end: int) -> int:
    """len(VideoDataset) (utility.py:399-402)."""
    return max(0, len(range(n_frames)[frame_srt:frame_end + 210]) - seq_len - dist)


def _softmax_columns(output: torch.Tensor, p0: torch.Tensor, cls: torch.Tensor, at: int):
    n = output.shape[0]
    sm = torch.softmax(output.float(), dim=1)
    p0[at:at + n] = sm[:, 0]
    cls[at:at + n] = sm.max(1)[1]


def video_window_probabilities(model: torch.nn.Module, frames: torch.Tensor, seq_len: int, dist: int, frame_srt: int = 0,
                               frame_end: int = -1, crop_size: int = 128, windows_per_launch: int = 1
                               ) -> Tuple[np.ndarray, np.ndarray]:
    """Softmax column 0 ("disruption") and arg-max of every sliding window of a frame stack resident in HBM.
    frames: (F, Hr, Wr, 3) uint8 on the GPU, in file-name order.  Window idx reads frames idx+1 .. idx+seq_len of
    frames[frame_srt : frame_end + 210] (utility.py:399-408; a window that runs off the end repeats the last frame, :425-429)."""
    ops.require_cuda(frames)
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise RuntimeError("mi355x hot path: a frame stack is (F, H, W, 3) uint8 on the GPU")
    sub = frames[frame_srt:frame_end + 210]
    n = max(0, sub.shape[0] - seq_len - dist)
    p0 = torch.zeros(n, device=frames.device, dtype=torch.float32)
    cls = torch.zeros(n, device=frames.device, dtype=torch.int64)
    model.eval()
    W = max(1, int(windows_per_launch))
    last = sub.shape[0] - 1
    steps = torch.arange(1, seq_len + 1, device=frames.device)
    with torch.no_grad():
        for at in range(0, n, W):
            m = min(W, n - at)
            if m == 1 and at + seq_len <= last:
                clip = sub[at + 1:at + 1 + seq_len].unsqueeze(0)                       # a view: no copy of the frames
            else:
                idx = (torch.arange(at, at + m, device=frames.device).view(-1, 1) + steps.view(1, -1)).clamp_(max=last)
                clip = sub[idx]                                                        # (m, seq_len, H, W, 3) uint8 gather
            x = preprocess_clips(clip, crop_size)
            # MD_GRAPH_STEP=1: the forward of every full batch of windows replayed from one HIP graph (src/utils/graphed.py)
            out = graphed_forward(model, [x], "_md_graphed_curve") if (_GRAPH and m == W) else model(x)
            _softmax_columns(out, p0, cls, at)
    return p0.cpu().numpy(), cls.cpu().numpy()


def series_window_probabilities(model: torch.nn.Module, values: torch.Tensor, seq_len: int, dist: int,
                                windows_per_launch: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """values: (n_rows, n_cols) fp32 on the GPU, already scaled.  Window idx = rows idx+1 .. idx+seq_len (utility.py:507-511)."""
    values = values.contiguous()
    ops.require_cuda(values)
    n = max(0, values.shape[0] - seq_len - dist)
    p0 = torch.zeros(n, device=values.device, dtype=torch.float32)
    cls = torch.zeros(n, device=values.device, dtype=torch.int64)
    model.eval()
    W = max(1, int(windows_per_launch))
    windows = values.float().unfold(0, seq_len, 1).permute(0, 2, 1)                    # (rows-seq_len+1, seq_len, cols) view
    with torch.no_grad():
        for at in range(0, n, W):
            m = min(W, n - at)
            x = windows[at + 1:at + 1 + m].contiguous()
            _softmax_columns(graphed_forward(model, [x], "_md_graphed_curve") if (_GRAPH and m == W) else model(x), p0, cls, at)
    return p0.cpu().numpy(), cls.cpu().numpy()


def _startup_correction(p: List, limit: float) -> List:
    return [0 if (i < limit and v >= 0.5) else v for i, v in enumerate(p)]                # utility.py:954-958


def assemble_video_curve(prob_list, clip_len: int, frame_srt: int):
    p = [0] * (clip_len + frame_srt) + list(prob_list)[1:-1]                              # :953
    p = _startup_correction(p, FPS * 1)
    return np.arange(0, len(p)) * (1 / FPS) * 1, p                                        # :961


def assemble_0D_curve(prob_list, seq_len: int, t_start: float):
    interval = 4                                                                          # :1040
    frame_srt = int(t_start * FPS / interval)
    p = [0] * (frame_srt + seq_len) + list(prob_list)[1:] + [0] * seq_len                 # :1043
    p = _startup_correction(p, FPS * 1)
    n = len(p)
    prob_x = np.linspace(0, n, num=n, endpoint=True) * (interval / FPS)                   # :1050
    q = np.interp(np.linspace(0, n * interval, num=n * interval, endpoint=True) * (1 / FPS), prob_x, np.asarray(p, dtype=np.float64))
    q = moving_avarage_smoothing(q, 12)                                                   # :1054
    return np.arange(0, len(q)) * (1 / FPS), q


def _shot_row(shot_list_dir: str, shot_num: int):
    import pandas as pd
    table = pd.read_csv(shot_list_dir, encoding="euc-kr")
    row = table[table.shot == shot_num]
    return {k: row[k].values[0] for k in ("tTQend", "tftsrt", "tipminf", "frame_startup", "frame_cutoff")}


def _shot_series(ts_data_dir: str, ts_cols: List, shot_num: int):
    import pandas as pd
    ts = pd.read_csv(ts_data_dir).reset_index()
    for col in ts_cols:
        ts[col] = ts[col].astype(np.float32)
    return ts[ts["shot"] == shot_num]                    # (the reference's interpolate() call discards its result, :927)


def load_frame_stack(file_path: str, device, height: int = 256, width: int = 256) -> torch.Tensor:
    """All frames of a shot directory, sorted by name, as one uint8 tensor in HBM.  Needs OpenCV, like the reference."""
    import glob
    import os
    try:
        import cv2
    except ImportError as e:
        raise RuntimeError("generate_prob_curve(file_path=...) decodes images with OpenCV, which is not installed; "
                           "pass frames=<(F,256,256,3) uint8 tensor> instead") from e
    paths = sorted(glob.glob(os.path.join(file_path, "*")))
    host = torch.empty((len(paths), height, width, 3), dtype=torch.uint8).pin_memory()
    for i, p in enumerate(paths):
        host[i] = torch.from_numpy(cv2.imread(p))
    return host.to(device, non_blocking=True)


def generate_prob_curve(file_path: Optional[str], model: torch.nn.Module, device: str = "cuda:0", save_dir: Optional[str] = None,
                        shot_list_dir: Optional[str] = "./dataset/KSTAR_Disruption_Shot_List_extend.csv",
                        ts_data_dir: Optional[str] = "./dataset/KSTAR_Disruption_ts_data_extend.csv",
                        ts_cols: Optional[List] = None, shot_num: Optional[int] = None, clip_len: Optional[int] = None,
                        dist_frame: Optional[int] = None, frames: Optional[torch.Tensor] = None, windows_per_launch: int = 1):
    """Reference utility.py:896-977; returns (time_x, prob_list).  ``frames`` (uint8, (F,256,256,3)) replaces ``file_path``."""
    row = _shot_row(shot_list_dir, shot_num)
    frame_srt, frame_end = int(row["frame_startup"]), int(row["frame_cutoff"])
    model.to(device)
    if frames is None:
        frames = load_frame_stack(file_path, device)
    frames = frames.to(device)
    p0, _ = video_window_probabilities(model, frames, clip_len, dist_frame, frame_srt, frame_end, 128, windows_per_launch)
    time_x, prob_list = assemble_video_curve(p0.tolist(), clip_len, frame_srt)
    print("\n(Info) flat-top : {:.3f}(s) | thermal quench : {:.3f}(s) | current quench : {:.3f}(s)\n".format(
        row["tftsrt"], row["tTQend"], row["tipminf"]))
    return time_x, prob_list


def generate_prob_curve_from_0D(model: torch.nn.Module, device: str = "cuda:0", save_dir: Optional[str] = None,
                                ts_data_dir: Optional[str] = "./dataset/KSTAR_Disruption_ts_data_extend.csv",
                                ts_cols: Optional[List] = None,
                                shot_list_dir: Optional[str] = "./dataset/KSTAR_Disruption_Shot_List_extend.csv",
                                shot_num: Optional[int] = None, seq_len: Optional[int] = None, dist: Optional[int] = None,
                                dt: Optional[float] = None, scaler=None, windows_per_launch: int = 1):
    """Reference utility.py:979-1066; returns (time_x, prob_list).  The scaler is fitted on the shot's own rows (:493-499)."""
    row = _shot_row(shot_list_dir, shot_num)
    ts = _shot_series(ts_data_dir, ts_cols, shot_num)
    t_start = ts.time.values[0]
    if scaler is None:
        from sklearn.preprocessing import RobustScaler
        scaler = RobustScaler()
    values = np.ascontiguousarray(scaler.fit_transform(ts[ts_cols].values), dtype=np.float32)
    model.to(device)
    p0, _ = series_window_probabilities(model, torch.from_numpy(values).to(device), seq_len, dist, windows_per_launch)
    time_x, prob_list = assemble_0D_curve(p0.tolist(), seq_len, t_start)
    print("\n(Info) flat-top : {:.3f}(s) | thermal quench : {:.3f}(s) | current quench : {:.3f}(s)\n".format(
        row["tftsrt"], row["tTQend"], row["tipminf"]))
    return time_x, prob_list


def multi_window_tables(n_frames: int, ts_time: np.ndarray, frame_srt: int, frame_end: int, t_srt: float, t_end: float,
                        vis_seq_len: int, ts_seq_len: int, dt: float, tau: int):
    """Frame indices and last 0D row of every window, as MultiModalDataset.__init__ matches them from the END of the shot
    backwards (utility.py:579-611; literal, including :605's use of the first-stage list length)."""
    video_indices = list(reversed(range(frame_end, frame_srt, -tau)))
    ts_idx_end = len(ts_time) - int(np.sum(np.asarray(ts_time) > t_end))
    ts_idx_start = int(t_srt / dt)
    ts_indices = list(reversed(range(ts_idx_end, ts_idx_start, -tau)))
    if len(video_indices) > len(ts_indices):
        video_indices = video_indices[-len(ts_indices):]
    elif len(video_indices) < len(ts_indices):
        ts_indices = ts_indices[-len(video_indices):]
    frames_of = [list(range(n_frames))[idx + 1: idx - tau * vis_seq_len + 1: -tau][::-1] for idx in video_indices if idx > vis_seq_len * tau]
    ts_sel = [idx for idx in ts_indices if idx > ts_seq_len * tau]
    if len(frames_of) > len(ts_sel):
        frames_of = frames_of[-len(ts_indices):]
    elif len(frames_of) < len(ts_sel):
        ts_sel = ts_sel[-len(frames_of):]
    return frames_of, ts_sel


def multi_window_probabilities(model: torch.nn.Module, frames: torch.Tensor, values: torch.Tensor, frames_of, ts_sel,
                               ts_seq_len: int, tau: int, crop_size: int = 128, windows_per_launch: int = 1):
    """frames (F, Hr, Wr, 3) uint8 and scaled 0D rows (n_rows, n_cols) fp32, both on the GPU; one fused-model forward per
    ``windows_per_launch`` windows; softmax column 0 / arg-max collected on the device."""
    ops.require_cuda(frames)
    values = values.contiguous().float()
    n = len(ts_sel)
    dev = frames.device
    p0 = torch.zeros(n, device=dev, dtype=torch.float32)
    cls = torch.zeros(n, device=dev, dtype=torch.int64)
    if n == 0:
        return p0.cpu().numpy(), cls.cpu().numpy()
    if any(len(f) != len(frames_of[0]) for f in frames_of):
        raise RuntimeError("generate_prob_curve_from_multi: windows of unequal length (shot shorter than one clip)")
    fidx = torch.tensor(frames_of, dtype=torch.int64, device=dev)                          # (n, vis_seq_len)
    steps = torch.arange(ts_seq_len, device=dev) * tau
    ridx = (torch.tensor(ts_sel, dtype=torch.int64, device=dev) - ts_seq_len * tau + 1).view(-1, 1) + steps.view(1, -1)   # rows idx_srt+1 ..
    model.eval()
    W = max(1, int(windows_per_launch))
    with torch.no_grad():
        for at in range(0, n, W):
            m = min(W, n - at)
            clip = preprocess_clips(frames[fidx[at:at + m]], crop_size)                   # gather of uint8 frames, then one launch
            xs = [clip, values[ridx[at:at + m]].contiguous()]
            out = graphed_forward(model, xs, "_md_graphed_curve") if (_GRAPH and m == W) else model(*xs)
            _softmax_columns(out, p0, cls, at)
    return p0.cpu().numpy(), cls.cpu().numpy()


def _interp_extrapolate(x, xp, fp):
    """scipy.interpolate.interp1d(kind="linear", fill_value="extrapolate") for increasing xp."""
    x, xp, fp = np.asarray(x, np.float64), np.asarray(xp, np.float64), np.asarray(fp, np.float64)
    i = np.clip(np.searchsorted(xp, x, side="right") - 1, 0, len(xp) - 2)
    return fp[i] + (fp[i + 1] - fp[i]) / (xp[i + 1] - xp[i]) * (x - xp[i])


def assemble_multi_curve(prob_list, t_srt: float, t_end: float, tau: int):
    """utility.py:1133-1170: (time_x, interpolated + smoothed curve).  The reference function returns the RAW list beside time_x."""
    dt_end, interval = 1.0, tau
    n0, n1 = int(t_srt * FPS / interval), int(dt_end * FPS / interval)
    body = list(prob_list)[1:]
    total = _startup_correction([0] * n0 + body + [0] * n1, FPS * 1.0 / interval)
    x_srt = [i * interval / FPS for i in range(0, n0)]
    x_prob = [x_srt[-1] + (i + 1) * 1 / FPS * interval for i in range(0, len(body) + n1)]
    q = _interp_extrapolate(np.linspace(0, t_end + dt_end, num=len(total) * interval, endpoint=True), x_srt + x_prob, total)
    q = moving_avarage_smoothing(q, 16, "center")
    return np.linspace(0, t_end + dt_end, num=len(q), endpoint=True), q


def generate_prob_curve_from_multi(file_path: Optional[str], model: torch.nn.Module, device: str = "cuda:0",
                                   save_dir: Optional[str] = None,
                                   ts_data_dir: Optional[str] = "./dataset/KSTAR_Disruption_ts_data_extend.csv",
                                   ts_cols: Optional[List] = None,
                                   shot_list_dir: Optional[str] = "./dataset/KSTAR_Disruption_Shot_List_extend.csv",
                                   shot_num: Optional[int] = None, vis_seq_len: Optional[int] = None,
                                   ts_seq_len: Optional[int] = None, dist: Optional[int] = None, dt: Optional[float] = None,
                                   scaler=None, tau: int = 1, frames: Optional[torch.Tensor] = None, windows_per_launch: int = 1):
    """Reference utility.py:1068-1178; returns (time_x, prob_list) like the reference: the time axis of the interpolated
    curve and the raw per-window probabilities.  The scaler is fitted on the shot's own rows when none is given (:573-578)."""
    row = _shot_row(shot_list_dir, shot_num)
    ts = _shot_series(ts_data_dir, ts_cols, shot_num)
    if scaler is None:
        from sklearn.preprocessing import RobustScaler
        values = RobustScaler().fit_transform(ts[ts_cols].values)
    else:
        values = scaler.transform(ts[ts_cols].values)
    values = np.ascontiguousarray(values, dtype=np.float32)
    model.to(device)
    if frames is None:
        frames = load_frame_stack(file_path, device)
    frames = frames.to(device)
    frames_of, ts_sel = multi_window_tables(frames.shape[0], ts.time.values, int(row["frame_startup"]), int(row["frame_cutoff"]),
                                            row["tftsrt"], row["tipminf"], vis_seq_len, ts_seq_len, dt, tau)
    p0, _ = multi_window_probabilities(model, frames, torch.from_numpy(values).to(device), frames_of, ts_sel, ts_seq_len, tau, 128,
                                       windows_per_launch)
    prob_list = p0.tolist()
    time_x, _ = assemble_multi_curve(prob_list, ts.time.values[ts_sel[0]], ts.time.values[ts_sel[-1]], tau)
    print("\n(Info) flat-top : {:.3f}(s) | thermal quench : {:.3f}(s) | current quench : {:.3f}(s)\n".format(
        row["tftsrt"], row["tTQend"], row["tipminf"]))
    return time_x, prob_list


def measure_computation_time(model: torch.nn.Module, input_shape: Tuple, n_samples: int = 1, device: str = "cuda:0"):
    """Reference utility.py:1201-1230: (mean, std, list) of the wall time of one forward on a zero input including its
    host-to-device copy -- but WITH a device synchronisation before the clock is read (the reference reads it while the GPU
    is still working), and without the per-iteration cache flush."""
    model.to(device)
    model.eval()
    t_measures = []
    sample = torch.zeros(input_shape).pin_memory()
    for _ in range(n_samples):
        with torch.no_grad():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = model(sample.to(device, non_blocking=True))
            torch.cuda.synchronize()
            t_measures.append(time.perf_counter() - t0)
        del out
    return float(np.mean(t_measures)), float(np.std(t_measures)), t_measures


def measure_computation_time_multi(model: torch.nn.Module, input_shape_vis: Tuple, input_shape_0D: Tuple, n_samples: int = 1,
                                   device: str = "cuda:0"):
    """Reference utility.py:1232-1265 (used by analysis/compute_time_multimodal.py:47-48): as measure_computation_time for a
    two-input model -- one forward on a zero clip and a zero 0D window, host-to-device copies included, WITH a device
    synchronisation before the clock is read.  A model that returns several logit sets (the *_GB variants) is timed as is."""
    model.to(device)
    model.eval()
    t_measures = []
    sample_vis = torch.zeros(input_shape_vis).pin_memory()
    sample_0d = torch.zeros(input_shape_0D).pin_memory()
    for _ in range(n_samples):
        with torch.no_grad():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = model(sample_vis.to(device, non_blocking=True), sample_0d.to(device, non_blocking=True))
            torch.cuda.synchronize()
            t_measures.append(time.perf_counter() - t0)
        del out
    return float(np.mean(t_measures)), float(np.std(t_measures)), t_measures
