"""Records tests/golden/eval_curve.npz from the REFERENCE's inference / evaluation path.
Run in the build container only (needs /root/reference):  python tests/golden/make_eval_golden.py

The reference functions run unmodified: evaluate (src/evaluate.py:11-134), generate_prob_curve and generate_prob_curve_from_0D
(src/utils/utility.py:896-1066), moving_avarage_smoothing (:872-893).  What is absent here and is not the thing recorded is
stood in for: cv2.imread returns frames of oracle.prob_curve.synth_frames by file name and glob2.glob lists those names (there
are no image files and no OpenCV), seaborn / the two plot_exp_prob_* figures are no-ops (presentation), the shot list and
0D table are small synthetic CSV files written to a temporary directory.  The fixture stores recipes' seeds and the
reference's OUTPUTS only."""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


class _Quiet:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: _Quiet()

    def __iter__(self):
        return iter((_Quiet(), _Quiet()))

    def __getitem__(self, i):
        return _Quiet()


FRAMES = {}
for name, attrs in (("cv2", dict(imread=lambda p: FRAMES[p], flip=lambda a, flipCode=1: np.ascontiguousarray(a[:, ::-1]))),
                    ("glob2", dict(glob=lambda pat: [k for k in FRAMES if os.path.dirname(k) == os.path.dirname(pat)])),
                    ("torchvision", {}), ("torchvision.transforms", {})):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m

import make_golden as MG          # noqa: E402  (sets up the reference import path and the remaining stubs)
import torch                      # noqa: E402
sys.modules["seaborn"].heatmap = lambda *a, **k: _Quiet()

from oracle import prob_curve as pc                 # noqa: E402
from oracle import r2plus1d as orc                  # noqa: E402
from src import evaluate as ref_eval                # noqa: E402  (reference)
from src.utils import utility as ref_util           # noqa: E402  (reference)
from src.loss import FocalLoss                      # noqa: E402  (reference)

ref_util.plot_exp_prob_type_1 = lambda *a, **k: None
ref_util.plot_exp_prob_type_2 = lambda *a, **k: None
ref_eval.plt = _Quiet()

LAYERS, ALPHA, SEED = [1, 1, 1, 1], 0.01, 9
CLIP, DIST, FRAME_SRT, FRAME_END, NFRAMES = 6, 3, 190, 40, 262
TS_COLS = ["\\q95", "\\ipmhd", "\\kappa", "\\tritop", "\\tribot", "\\betap"]
SHOT = 21310


class F32Scaler:
    """A scaler in the reference's sense (fit_transform / transform) that keeps float32, so that the 0D model sees float32."""

    def fit_transform(self, X):
        X = np.asarray(X, dtype=np.float32)
        self.med = np.median(X, axis=0).astype(np.float32)
        self.iqr = (np.percentile(X, 75, axis=0) - np.percentile(X, 25, axis=0)).astype(np.float32)
        return ((X - self.med) / self.iqr).astype(np.float32)


def series_table(seed, n_rows=120, t0=0.9):
    g = np.random.RandomState(seed)
    t = t0 + np.arange(n_rows) * (4.0 / 210)
    vals = np.cumsum(g.standard_normal((n_rows, len(TS_COLS))).astype(np.float32) * 0.3, axis=0) + g.standard_normal(len(TS_COLS)).astype(np.float32)
    return t, vals.astype(np.float32)


def calibrate(model, clips, rec, tag):
    """Random weights on pixel-scale inputs saturate the softmax.  Give the model running statistics that fit the data (one
    training-mode pass with momentum 1) and centre the two logits, so that the curve crosses 0.5; what was changed is stored."""
    model.train()
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.BatchNorm1d)):
            m.momentum = 1.0
    with torch.no_grad():
        model(clips)
        model.eval()
        lg = model(clips)
        d = (lg[:, 0] - lg[:, 1])
        last = [m for m in model.modules() if isinstance(m, torch.nn.Linear)][-1]
        last.weight.mul_(2.0 / float(d.std()))
        lg = model(clips)
        d = (lg[:, 0] - lg[:, 1])
        # centre near the median, at the offset that keeps every calibration sample farthest from p = 0.5 (d = 0)
        cands = [float(d.median()) + o for o in np.linspace(-0.5, 0.5, 41)]
        last.bias[0] -= max(cands, key=lambda c: float((d - c).abs().min()))
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.BatchNorm1d)):
            m.momentum = 0.1
    last_name = [k for k, m in model.named_modules() if m is last][0]
    for k, v in model.state_dict().items():
        if "running" in k or k.startswith(last_name + "."):
            rec[tag + "/state/" + k] = v.numpy().copy()


def main():
    rec = {}
    torch.manual_seed(0)
    # ---- evaluate(): threshold rule on 3 batches -----------------------------------------------------------------------------
    model = MG.load_ref_model(LAYERS, 4, 32, ALPHA, SEED)
    batches = [(orc.synth_clip(6, 4, 32, SEED + i), orc.synth_labels(6, SEED + i, 0.5)) for i in range(3)]
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    calibrate(model, torch.cat([x for x, _ in batches]), rec, "eval")
    with torch.no_grad():
        model.eval()
        p0 = torch.cat([torch.softmax(model(x), 1)[:, 0] for x, _ in batches]).numpy()
    rec["eval/p0"] = p0
    for thr in (0.5, float(np.sort(p0)[len(p0) // 3]) + 1e-4):
        tl, ta, tf = ref_eval.evaluate(batches, model, opt, FocalLoss(weight=torch.tensor([1.0, 2.0]), gamma=2.0), "cpu", None, None,
                                       thr, "single")
        rec["eval/thr%.4f" % thr] = np.array([thr, tl, ta, tf], dtype=np.float64)
        print("evaluate thr", thr, tl, ta, tf)
    # ---- generate_prob_curve: sliding windows over a frame stack ---------------------------------------------------------------
    frames = pc.synth_frames(NFRAMES, SEED)
    for i in range(NFRAMES):
        FRAMES["/frames/%06d.jpg" % i] = frames[i]
    tmp = tempfile.mkdtemp()
    shot_csv, ts_csv = os.path.join(tmp, "shots.csv"), os.path.join(tmp, "ts.csv")
    t, vals = series_table(SEED)
    with open(shot_csv, "w", encoding="euc-kr") as f:
        f.write("shot,tTQend,tftsrt,tipminf,frame_startup,frame_cutoff\n%d,1.05,0.3,1.1,%d,%d\n" % (SHOT, FRAME_SRT, FRAME_END))
    with open(ts_csv, "w") as f:
        f.write("time,shot," + ",".join(TS_COLS) + "\n")
        for other in (SHOT - 1, SHOT):                      # a second shot in the table: the path must select by shot number
            for i in range(len(t)):
                f.write("%.9f,%d," % (t[i], other) + ",".join("%.9g" % (v + (other != SHOT)) for v in vals[i]) + "\n")
    model = MG.load_ref_model(LAYERS, CLIP, 128, ALPHA, SEED)
    cal = [torch.from_numpy(c) for i, c in enumerate(pc.video_windows(frames, CLIP, DIST, FRAME_SRT, FRAME_END, 128)) if i % 4 == 0]
    calibrate(model, torch.stack(cal), rec, "video")
    seen = []
    hook = model.register_forward_hook(lambda m, i, o: seen.append(torch.softmax(o, 1)[0].numpy().copy()))
    time_x, prob = ref_util.generate_prob_curve("/frames", model, "cpu", None, shot_csv, ts_csv, TS_COLS, SHOT, CLIP, DIST)
    hook.remove()
    rec["video/time_x"], rec["video/prob"], rec["video/window_softmax"] = np.asarray(time_x), np.asarray(prob, dtype=np.float64), np.stack(seen)
    rec["video/cfg"] = np.array([CLIP, DIST, FRAME_SRT, FRAME_END, NFRAMES, SEED])
    print("video curve", len(prob), "windows", len(seen), "p range", np.stack(seen)[:, 0].min(), np.stack(seen)[:, 0].max(),
          "zeroed", sum(1 for i, s in enumerate(seen[1:-1]) if s[0] >= 0.5 and CLIP + FRAME_SRT + i < 210))
    # two windows of the reference's own dataset object, to pin the frame indexing / crop / normalisation
    ds = ref_util.VideoDataset("/frames", 256, 256, 128, CLIP, DIST, FRAME_SRT, FRAME_END)
    rec["video/n_windows"] = np.array(len(ds))
    for i in (0, len(ds) - 1):
        rec["video/clip%d" % i] = MG.subsample(ds[i], 96)
    # ---- generate_prob_curve_from_0D -------------------------------------------------------------------------------------------
    from src.models.transformer import Transformer
    torch.manual_seed(73)
    tm = Transformer(n_features=len(TS_COLS), kernel_size=3, feature_dims=16, max_len=CLIP, n_layers=1, n_heads=2, dim_feedforward=24,
                     dropout=0.0, cls_dims=12, n_classes=2)
    tm.encoder.noise.std = 0.0
    for k, v in tm.state_dict().items():
        rec["ts/sd/" + k] = v.numpy().copy()
    seen = []
    hook = tm.register_forward_hook(lambda m, i, o: seen.append(torch.softmax(o, 1)[0].detach().numpy().copy()))
    time_x, prob = ref_util.generate_prob_curve_from_0D(tm, "cpu", None, ts_csv, TS_COLS, shot_csv, SHOT, CLIP, DIST, 4.0 / 210, F32Scaler())
    hook.remove()
    rec["ts/time_x"], rec["ts/prob"], rec["ts/window_softmax"] = np.asarray(time_x), np.asarray(prob, dtype=np.float64), np.stack(seen)
    print("0D curve", len(prob), "windows", len(seen))
    # ---- DatasetForVideo (src/dataset.py:31-144): clip index / label logic and the deterministic preprocessing -----------------
    import pandas as pd
    from src.dataset import DatasetForVideo          # reference
    for i in range(NFRAMES):
        FRAMES["/shots/%d/%06d.jpg" % (SHOT, i)] = frames[i]
    df = pd.DataFrame({"shot": [SHOT], "frame_tipminf": [250], "frame_startup": [FRAME_SRT]})
    dv = DatasetForVideo(["/shots/%d" % SHOT], df, augmentation=False, crop_size=128, seq_len=CLIP, dist=DIST)
    rec["dsv/n"] = np.array(len(dv)); rec["dsv/labels"] = np.asarray(dv.labels)
    rec["dsv/cfg"] = np.array([250, FRAME_SRT, CLIP, DIST, 128])
    for i in (0, len(dv) - 1):
        clip, lab = dv[i]
        rec["dsv/clip%d" % i] = MG.subsample(clip, 96)
    print("DatasetForVideo", len(dv), dv.labels)
    # ---- augmentations (src/dataset.py:129-135, 152-227): everything that does not need OpenCV (contrast / blur probability 0) ------
    import random as pyrandom
    AUG = {"bright_val": 30, "bright_p": 0.7, "contrast_min": 1, "contrast_max": 1.15, "contrast_p": 0.0, "blur_k": 5, "blur_p": 0.0,
           "flip_p": 0.5, "vertical_ratio": 0.2, "vertical_p": 0.7, "horizontal_ratio": 0.2, "horizontal_p": 0.7}
    da = DatasetForVideo(["/shots/%d" % SHOT], df, augmentation=True, augmentation_args=dict(AUG), crop_size=128, seq_len=CLIP, dist=DIST)
    pyrandom.seed(1234); np.random.seed(4321)
    clips = []
    for rep in range(8):
        clips.append(da[rep % len(da)][0].numpy())
    rec["aug/args"] = np.array([AUG[k] for k in sorted(AUG)], dtype=np.float64)
    rec["aug/seeds"] = np.array([1234, 4321])
    for i, c in enumerate(clips):
        rec["aug/clip%d" % i] = MG.subsample(torch.from_numpy(c), 4096)
        rec["aug/sum%d" % i] = np.float64(c.astype(np.float64).sum())
    print("augmentation: 8 clips recorded")
    # ---- generate_prob_curve_from_multi (utility.py:1068-1178): MultiModalDataset index matching + curve assembly --------------
    from src.models.MultiModal import MultiModalModel     # reference
    SHOT2, SRT2, END2 = 21311, 20, 200
    with open(shot_csv, "a", encoding="euc-kr") as f:
        f.write("%d,1.05,0.3,1.1,%d,%d\n" % (SHOT2, SRT2, END2))
    t2 = np.arange(120) * (4.0 / 210)
    with open(ts_csv, "a") as f:
        for i in range(len(t2)):
            f.write("%.9f,%d," % (t2[i], SHOT2) + ",".join("%.9g" % v for v in vals[i]) + "\n")
    torch.manual_seed(77)
    AVm = dict(image_size=128, patch_size=16, n_frames=CLIP, dim=16, depth=1, n_heads=2, in_channels=3, d_head=8, dropout=0.0,
               embedd_dropout=0.0, scale_dim=2, pool="mean")
    A0m = dict(n_features=len(TS_COLS), kernel_size=3, feature_dims=16, max_len=CLIP, n_layers=1, n_heads=2, dim_feedforward=24,
               dropout=0.0)
    mm = MultiModalModel(2, dict(AVm), dict(A0m))
    for mod in mm.modules():
        if type(mod).__name__ == "NoiseLayer":
            mod.std = 0.0
    with torch.no_grad():                                  # pixel-scale inputs: keep the patch embedding O(1)
        mm.encoder_video.to_patch_embedding[1].weight.mul_(0.02)
    for k, v in mm.state_dict().items():
        rec["multi/sd/" + k] = v.numpy().copy()
    seen = []
    hook = mm.register_forward_hook(lambda m, i, o: seen.append(torch.softmax(o, 1)[0].detach().numpy().copy()))
    time_x, prob = ref_util.generate_prob_curve_from_multi("/frames", mm, "cpu", None, ts_csv, TS_COLS, shot_csv, SHOT2, CLIP, CLIP,
                                                           DIST, 4.0 / 210, None, 1)
    hook.remove()
    rec["multi/time_x"], rec["multi/prob"], rec["multi/window_softmax"] = np.asarray(time_x), np.asarray(prob, dtype=np.float64), np.stack(seen)
    rec["multi/cfg"] = np.array([SHOT2, SRT2, END2, CLIP, CLIP, DIST, 1])
    print("multi curve", len(time_x), "windows", len(seen), "p range", np.stack(seen)[:, 0].min(), np.stack(seen)[:, 0].max())
    # ---- smoothing ---------------------------------------------------------------------------------------------------------------
    x = np.random.RandomState(5).rand(64) * 1.4 - 0.2
    rec["smooth/x"] = x
    rec["smooth/backward12"] = ref_util.moving_avarage_smoothing(x, 12)
    rec["smooth/center16"] = ref_util.moving_avarage_smoothing(x, 16, "center")
    np.savez_compressed(os.path.join(HERE, "eval_curve.npz"), **rec)
    print("wrote eval_curve.npz", sum(v.nbytes for v in rec.values()), "bytes")


if __name__ == "__main__":
    main()
