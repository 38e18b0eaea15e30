"""CPU restatement (test infrastructure only) of the deterministic part of the reference's DatasetForVideo.get_video_data
(src/dataset.py:124-144): load_frames' uint8 -> float32 (:102-107), crop with is_random=False (:241-246), normalize (:203-207),
to_tensor (:229-230).  PARITY UNPINNED: the reference module imports cv2, which is not installed here, and its tests hold no
fixture for this path; the arithmetic (uint8 -> fp32, minus a constant) is exact, so the restatement follows the text."""
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
