"""CPU restatement (test infrastructure only) of the deterministic part of the reference's DatasetForVideo.get_video_data
(src/dataset.py:124-144): load_frames' uint8 -> float32 (:102-107), crop with is_random=False (:241-246), normalize (:203-207),
to_tensor (:229-230), and of the clip table DatasetForVideo.__init__ builds (:80-96).  Pinned by tests/golden/eval_curve.npz
(keys dsv/*: recorded from the reference's DatasetForVideo with augmentation off; cv2.imread / glob2 stand-ins serve
oracle.prob_curve.synth_frames by file name -- tests/golden/make_eval_golden.py)."""
import numpy as np


def video_clip(frames_u8: np.ndarray, crop_size: int) -> np.ndarray:
    """frames (T, Hr, Wr, 3) uint8 -> (3, T, S, S) float32."""
    buf = frames_u8.astype(np.float32)                                                # :105
    Hr, Wr = buf.shape[1], buf.shape[2]
    mid_x, mid_y = Hr // 2, Wr // 2                                                   # :243
    off = crop_size // 2                                                              # :244
    buf = buf[:, mid_x - off:mid_x + off, mid_y - off:mid_y + off, :]                 # :245
    buf = buf - np.array([[[90.0, 98.0, 102.0]]], dtype=np.float32)                   # :205 (float64 constant applied to a float32 frame in place)
    return np.ascontiguousarray(buf.transpose((3, 0, 1, 2)))                          # :230


def clip_table(frame_tipminf: int, frame_startup: int, seq_len: int, dist: int):
    """(first-frame index - 1 of every clip, labels): clip i reads frames idx+1 .. idx+seq_len; the LAST clip of a shot (the one
    ending ``dist`` frames before the current quench) is labelled 0 = disruptive, all earlier ones 1 (:83-94)."""
    dis_frame = frame_tipminf - dist
    indices = [i for i in reversed(range(dis_frame - seq_len, frame_startup, -seq_len))]
    labels = [0 if idx == indices[-1] else 1 for idx in indices]
    return indices, labels
