"""Scope row (f)-4: the inference / evaluation path -- evaluate()'s threshold rule (src/evaluate.py:11-134) and the sliding-window
probability curves (src/utils/utility.py:872-1066).  tests/golden/eval_curve.npz was recorded from the reference itself
(tests/golden/make_eval_golden.py: frames from oracle.prob_curve.synth_frames stand in for image files).
CPU: the oracle restatement and the native module's host arithmetic reproduce the fixture.  GPU: the native models driven
through src.utils.prob_curve / src.evaluate reproduce the reference's per-window softmax to 1e-3, hence its curves."""
import os

import numpy as np
import pytest
import torch

from oracle import prob_curve as pc
from oracle import r2plus1d as orc

LAYERS, ALPHA = [1, 1, 1, 1], 0.01
TS_COLS = ["\\q95", "\\ipmhd", "\\kappa", "\\tritop", "\\tribot", "\\betap"]
SHOT = 21310


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "eval_curve.npz"))


def _cfg(gold):
    clip, dist, srt, end, nfr, seed = [int(v) for v in gold["video/cfg"]]
    return clip, dist, srt, end, nfr, seed


def _state(gold, tag, seed):
    params, bufs = orc.synth_state(LAYERS, seed, ALPHA)
    sd = dict(params); sd.update(bufs)
    for k in gold.files:
        if k.startswith(tag + "/state/"):
            sd[k[len(tag) + 7:]] = torch.from_numpy(gold[k])
    return sd


def _sub(t, n=96):
    f = t.reshape(-1)
    return f[::max(1, f.numel() // n)][:n].numpy()


def test_oracle_windows_and_softmax_match_the_reference(gold):
    clip, dist, srt, end, nfr, seed = _cfg(gold)
    frames = pc.synth_frames(nfr, seed)
    wins = list(pc.video_windows(frames, clip, dist, srt, end, 128))
    assert len(wins) == int(gold["video/n_windows"]) == pc.video_window_count(nfr, clip, dist, srt, end)
    for i in (0, len(wins) - 1):
        assert np.array_equal(_sub(torch.from_numpy(wins[i])), gold["video/clip%d" % i])
    sd = _state(gold, "video", seed)
    params = {k: v for k, v in sd.items() if "running" not in k and "num_batches" not in k}
    bufs = {k: v for k, v in sd.items() if k not in params}
    ref = gold["video/window_softmax"]
    for i in (0, 7, len(wins) - 1):
        lg = orc.classifier_forward(torch.from_numpy(wins[i])[None], params, bufs, LAYERS, ALPHA, training=False)
        sm = torch.softmax(lg, 1)[0].numpy()
        assert np.abs(sm - ref[i]).max() <= 2e-5, (i, sm, ref[i])


def test_curve_assembly_matches_the_reference(gold):
    from src.utils import prob_curve as npc
    clip, dist, srt, end, nfr, seed = _cfg(gold)
    p0 = gold["video/window_softmax"][:, 0].tolist()
    for mod in (pc, npc):
        t, p = mod.assemble_video_curve(p0, clip, srt)
        assert np.array_equal(np.asarray(t), gold["video/time_x"]) and np.array_equal(np.asarray(p, dtype=np.float64), gold["video/prob"])
    # the start-up correction was exercised: some window with p >= 0.5 before sample 210 was zeroed, some later one kept
    body = np.asarray(p0[1:-1]); at = clip + srt + np.arange(len(body))
    assert np.any((body >= 0.5) & (at < 210)) and np.any((body >= 0.5) & (at >= 210))
    assert np.abs(body - 0.5).min() > 5e-3              # no window sits on the threshold: 1e-3 agreement decides the same way
    t0 = 0.9
    p0 = gold["ts/window_softmax"][:, 0].tolist()
    for mod in (pc, npc):
        t, p = mod.assemble_0D_curve(p0, clip, t0)
        assert np.allclose(t, gold["ts/time_x"], rtol=0, atol=1e-12) and np.allclose(p, gold["ts/prob"], rtol=0, atol=1e-12)


def test_smoothing_matches_the_reference(gold):
    from src.utils.prob_curve import moving_avarage_smoothing
    x = gold["smooth/x"]
    for f in (pc.moving_average, moving_avarage_smoothing):
        assert np.allclose(f(x, 12), gold["smooth/backward12"], rtol=0, atol=1e-12)
        assert np.allclose(f(x, 16, "center"), gold["smooth/center16"], rtol=0, atol=1e-12)


def test_threshold_rule_matches_the_reference(gold):
    from src.utils.metrics import macro_f1
    p0 = gold["eval/p0"]
    labels = torch.cat([orc.synth_labels(6, 9 + i, 0.5) for i in range(3)]).numpy()
    seen = 0
    for k in gold.files:
        if k.startswith("eval/thr"):
            thr, _, acc, f1 = gold[k]
            pred = pc.threshold_predictions(p0, thr)
            assert abs(float((pred == labels).mean()) - acc) < 1e-12 and abs(macro_f1(labels, pred) - f1) < 1e-12
            seen += 1
    assert seen == 2


def test_dataset_for_video_clip_table_and_preprocessing_match_the_reference(gold):
    """src/dataset.py:80-144 (augmentation off): which frames a training clip reads, its label, crop / mean / transpose."""
    from oracle import preprocess as op
    tip, srt, L, dist, crop = [int(v) for v in gold["dsv/cfg"]]
    idx, lab = op.clip_table(tip, srt, L, dist)
    assert len(idx) == int(gold["dsv/n"]) and np.array_equal(lab, gold["dsv/labels"]) and lab[-1] == 0 and set(lab[:-1]) == {1}
    frames = pc.synth_frames(_cfg(gold)[4], _cfg(gold)[5])
    for i in (0, len(idx) - 1):
        clip = op.video_clip(frames[idx[i] + 1: idx[i] + L + 1], crop)
        assert np.array_equal(_sub(torch.from_numpy(clip)), gold["dsv/clip%d" % i])


MULTI_AV = dict(image_size=128, patch_size=16, dim=16, depth=1, n_heads=2, in_channels=3, d_head=8, dropout=0.0, embedd_dropout=0.0,
                scale_dim=2, pool="mean")
MULTI_A0 = dict(n_features=len(TS_COLS), kernel_size=3, feature_dims=16, n_layers=1, n_heads=2, dim_feedforward=24, dropout=0.0)


def _multi_inputs(gold):
    from sklearn.preprocessing import RobustScaler
    shot, srt, end, vl, tl, dist, tau = [int(v) for v in gold["multi/cfg"]]
    g = np.random.RandomState(_cfg(gold)[5])
    vals = (np.cumsum(g.standard_normal((120, len(TS_COLS))).astype(np.float32) * 0.3, axis=0)
            + g.standard_normal(len(TS_COLS)).astype(np.float32)).astype(np.float32)
    # the table is written with %.9g and read back as float32: identical values
    t2 = np.arange(120) * (4.0 / 210)
    t2 = np.array([float("%.9f" % v) for v in t2])
    scaled = np.ascontiguousarray(RobustScaler().fit_transform(vals), dtype=np.float32)
    return (shot, srt, end, vl, tl, dist, tau), t2, vals, scaled


def test_multi_window_matching_and_curve_assembly_match_the_reference(gold):
    from oracle import multimodal as om
    from src.utils import prob_curve as npc
    (shot, srt, end, vl, tl, dist, tau), t2, vals, scaled = _multi_inputs(gold)
    nfr, seed = _cfg(gold)[4], _cfg(gold)[5]
    ref = gold["multi/window_softmax"]
    for mod in (pc, npc):
        frames_of, ts_sel = mod.multi_window_tables(nfr, t2, srt, end, 0.3, 1.1, vl, tl, 4.0 / 210, tau)
        assert len(frames_of) == len(ts_sel) == len(ref)
        tx, q = mod.assemble_multi_curve(ref[:, 0].tolist(), t2[ts_sel[0]], t2[ts_sel[-1]], tau)
        assert np.allclose(tx, gold["multi/time_x"], rtol=0, atol=1e-12) and len(q) == len(tx) and np.all((q >= 0) & (q <= 1))
    assert np.array_equal(gold["multi/prob"], ref[:, 0].astype(np.float64))          # the reference returns the raw list
    # the oracle's fused model on the oracle's windows reproduces the reference's per-window softmax
    sd = {k[len("multi/sd/"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("multi/sd/")}
    frames = pc.synth_frames(nfr, seed)
    old_v, old_t = dict(om.VIDEO), dict(om.TS)
    om.VIDEO.update(patch_size=16, depth=1, n_heads=2); om.TS.update(n_layers=1, n_heads=2, kernel_size=3)
    try:
        for i in (0, len(ref) - 1):
            clip = pc.video_windows(frames[frames_of[i][0] - 1:], vl, 0, 0, 10 ** 6, 128)        # window 0 of the shifted stack
            x_vis = torch.from_numpy(next(clip))[None]
            x_ts = torch.from_numpy(pc.multi_ts_window(scaled, ts_sel[i], tl, tau))[None]
            sm = torch.softmax(om.multimodal_forward(x_vis, x_ts, {k: v.clone() for k, v in sd.items()}, "mean", training=False), 1)[0]
            assert np.abs(sm.numpy() - ref[i]).max() <= 2e-5, (i, sm, ref[i])
    finally:
        om.VIDEO.clear(); om.VIDEO.update(old_v); om.TS.clear(); om.TS.update(old_t)


# ------------------------------------------------------------------------------------------------------------------------------
def _native_classifier(gold, tag, T, S, seed):
    from src.models.R2Plus1D import R2Plus1DClassifier
    m = R2Plus1DClassifier(input_size=(3, T, S, S), num_classes=2, layer_sizes=LAYERS, alpha=ALPHA)
    m.load_state_dict(_state(gold, tag, seed), strict=True)
    return m.cuda()


def _write_tables(tmp_path, srt, end, seed):
    shot_csv, ts_csv = str(tmp_path / "shots.csv"), str(tmp_path / "ts.csv")
    g = np.random.RandomState(seed)
    t = 0.9 + np.arange(120) * (4.0 / 210)
    vals = (np.cumsum(g.standard_normal((120, len(TS_COLS))).astype(np.float32) * 0.3, axis=0)
            + g.standard_normal(len(TS_COLS)).astype(np.float32)).astype(np.float32)
    with open(shot_csv, "w", encoding="euc-kr") as f:
        f.write("shot,tTQend,tftsrt,tipminf,frame_startup,frame_cutoff\n%d,1.05,0.3,1.1,%d,%d\n" % (SHOT, srt, end))
    with open(ts_csv, "w") as f:
        f.write("time,shot," + ",".join(TS_COLS) + "\n")
        for other in (SHOT - 1, SHOT):
            for i in range(len(t)):
                f.write("%.9f,%d," % (t[i], other) + ",".join("%.9g" % (v + (other != SHOT)) for v in vals[i]) + "\n")
    return shot_csv, ts_csv


class F32Scaler:
    def fit_transform(self, X):
        X = np.asarray(X, dtype=np.float32)
        med = np.median(X, axis=0).astype(np.float32)
        iqr = (np.percentile(X, 75, axis=0) - np.percentile(X, 25, axis=0)).astype(np.float32)
        return ((X - med) / iqr).astype(np.float32)


@pytest.mark.gpu
def test_video_probability_curve_on_gpu_matches_the_reference(gold, tmp_path):
    from src.utils import prob_curve as npc
    clip, dist, srt, end, nfr, seed = _cfg(gold)
    frames = torch.from_numpy(pc.synth_frames(nfr, seed)).cuda()
    model = _native_classifier(gold, "video", clip, 128, seed)
    ref = gold["video/window_softmax"]
    p1, c1 = npc.video_window_probabilities(model, frames, clip, dist, srt, end, 128, windows_per_launch=1)
    p8, c8 = npc.video_window_probabilities(model, frames, clip, dist, srt, end, 128, windows_per_launch=8)
    assert len(p1) == len(ref) == npc.video_window_count(nfr, clip, dist, srt, end)
    assert np.abs(p1 - ref[:, 0]).max() <= 1e-3, np.abs(p1 - ref[:, 0]).max()
    assert np.abs(p8 - p1).max() <= 1e-4 and np.array_equal(c1, ref.argmax(1)) and np.array_equal(c8, c1)
    shot_csv, ts_csv = _write_tables(tmp_path, srt, end, seed)
    for w in (1, 16):
        t, p = npc.generate_prob_curve(None, model, "cuda:0", None, shot_csv, ts_csv, TS_COLS, SHOT, clip, dist, frames=frames,
                                       windows_per_launch=w)
        assert np.array_equal(np.asarray(t), gold["video/time_x"])
        p, g = np.asarray(p, dtype=np.float64), gold["video/prob"]
        assert np.array_equal(p == 0, g == 0) and np.abs(p - g).max() <= 1e-3
    assert not model.training and all(q.grad is None for q in model.parameters())


@pytest.mark.gpu
def test_0D_probability_curve_on_gpu_matches_the_reference(gold, tmp_path):
    from src.models.transformer import Transformer
    from src.utils import prob_curve as npc
    clip, dist, srt, end, nfr, seed = _cfg(gold)
    m = Transformer(n_features=len(TS_COLS), kernel_size=3, feature_dims=16, max_len=clip, n_layers=1, n_heads=2, dim_feedforward=24,
                    dropout=0.0, cls_dims=12, n_classes=2)
    m.load_state_dict({k[len("ts/sd/"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("ts/sd/")}, strict=True)
    m.encoder.noise.std = 0.0
    shot_csv, ts_csv = _write_tables(tmp_path, srt, end, seed)
    for w in (1, 64):
        t, p = npc.generate_prob_curve_from_0D(m, "cuda:0", None, ts_csv, TS_COLS, shot_csv, SHOT, clip, dist, 4.0 / 210, F32Scaler(),
                                               windows_per_launch=w)
        assert np.allclose(t, gold["ts/time_x"], rtol=0, atol=1e-12)
        assert np.abs(np.asarray(p) - gold["ts/prob"]).max() <= 1e-3


@pytest.mark.gpu
def test_evaluate_on_gpu_matches_the_reference(gold, tmp_path):
    from src.evaluate import evaluate
    from src.loss import FocalLoss
    model = _native_classifier(gold, "eval", 4, 32, 9)
    batches = [(orc.synth_clip(6, 4, 32, 9 + i), orc.synth_labels(6, 9 + i, 0.5)) for i in range(3)]
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    seen = 0
    for k in gold.files:
        if k.startswith("eval/thr"):
            thr, tl, ta, tf = [float(v) for v in gold[k]]
            p0 = gold["eval/p0"]
            assert np.abs(p0 - thr).min() > 5e-5
            loss = FocalLoss(weight=torch.tensor([1.0, 2.0], device="cuda:0"), gamma=2.0)
            l, a, f = evaluate(batches, model, opt, loss, "cuda:0", None, str(tmp_path / "r.txt"), thr, "single")
            assert abs(l - tl) <= 1e-3 * max(1.0, abs(tl)) and abs(a - ta) < 1e-12 and abs(f - tf) < 1e-12, (l, tl, a, ta, f, tf)
            assert os.path.isfile(tmp_path / "r.txt")
            seen += 1
    assert seen == 2


@pytest.mark.gpu
def test_measure_computation_time_is_synchronised(gold):
    from src.utils.prob_curve import measure_computation_time
    model = _native_classifier(gold, "eval", 4, 32, 9)
    mean, std, ts = measure_computation_time(model, (1, 3, 4, 32, 32), 5, "cuda:0")
    assert len(ts) == 5 and mean > 0 and all(t > 0 for t in ts)


@pytest.mark.gpu
def test_multi_probability_curve_on_gpu_matches_the_reference(gold, tmp_path):
    from src.models.MultiModal import MultiModalModel
    from src.utils import prob_curve as npc
    (shot, srt, end, vl, tl, dist, tau), t2, vals, scaled = _multi_inputs(gold)
    nfr, seed = _cfg(gold)[4], _cfg(gold)[5]
    m = MultiModalModel(2, dict(MULTI_AV, n_frames=vl), dict(MULTI_A0, max_len=tl))
    m.load_state_dict({k[len("multi/sd/"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("multi/sd/")}, strict=True)
    for mod in m.modules():
        if type(mod).__name__ == "NoiseLayer":
            mod.std = 0.0
    shot_csv, ts_csv = str(tmp_path / "shots.csv"), str(tmp_path / "ts.csv")
    with open(shot_csv, "w", encoding="euc-kr") as f:
        f.write("shot,tTQend,tftsrt,tipminf,frame_startup,frame_cutoff\n%d,1.05,0.3,1.1,%d,%d\n" % (shot, srt, end))
    with open(ts_csv, "w") as f:
        f.write("time,shot," + ",".join(TS_COLS) + "\n")
        for i in range(120):
            f.write("%.9f,%d," % (i * (4.0 / 210), shot) + ",".join("%.9g" % v for v in vals[i]) + "\n")
    frames = torch.from_numpy(pc.synth_frames(nfr, seed)).cuda()
    ref = gold["multi/window_softmax"][:, 0]
    for w in (1, 16):
        t, p = npc.generate_prob_curve_from_multi(None, m, "cuda:0", None, ts_csv, TS_COLS, shot_csv, shot, vl, tl, dist, 4.0 / 210,
                                                  None, tau, frames=frames, windows_per_launch=w)
        assert np.allclose(t, gold["multi/time_x"], rtol=0, atol=1e-12)
        assert len(p) == len(ref) and np.abs(np.asarray(p) - ref).max() <= 1e-3, np.abs(np.asarray(p) - ref).max()
