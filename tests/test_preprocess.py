"""device-side clip preprocessing (centre crop, BGR mean subtraction, layout) against the NumPy restatement of
src/dataset.py:124-144 - bit-exact (uint8 -> fp32 minus a constant).  Parity unpinned by the reference (its dataset module needs
cv2, absent here; no fixture exists): the oracle follows the source text."""
import numpy as np
import pytest
import torch

from oracle import preprocess as op


def test_oracle_shapes_and_values():
    f = np.arange(2 * 6 * 8 * 3, dtype=np.uint8).reshape(2, 6, 8, 3)
    out = op.video_clip(f, 4)
    assert out.shape == (3, 2, 4, 4) and out.dtype == np.float32
    assert out[0, 0, 0, 0] == float(f[0, 1, 2, 0]) - 90.0 and out[2, 1, 3, 3] == float(f[1, 4, 5, 2]) - 102.0


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,Hr,Wr,S", [(2, 5, 256, 256, 128), (3, 4, 37, 50, 24), (1, 21, 128, 128, 128)])
def test_device_preprocess_is_bit_exact(B, T, Hr, Wr, S):
    from src.utils.clip_preprocess import preprocess_clips
    rng = np.random.default_rng(B * 100 + S)
    frames = rng.integers(0, 256, size=(B, T, Hr, Wr, 3), dtype=np.uint8)
    ref = np.stack([op.video_clip(frames[b], S) for b in range(B)])
    dev = torch.from_numpy(frames).cuda()
    out = preprocess_clips(dev, S)
    assert tuple(out.shape) == (B, 3, T, S, S) and np.array_equal(out.cpu().numpy(), ref)
    cl = preprocess_clips(dev, S, channels_last=True)
    assert tuple(cl.shape) == (B, T, S, S, 4)
    assert np.array_equal(cl[..., :3].permute(0, 4, 1, 2, 3).cpu().numpy(), ref) and float(cl[..., 3].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        preprocess_clips(dev, S + 1)                       # odd crop sizes would lose a row in the reference's slicing
    with pytest.raises(RuntimeError):
        preprocess_clips(dev.float(), S)
