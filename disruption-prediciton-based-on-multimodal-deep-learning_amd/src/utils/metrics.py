"""Label bookkeeping used by the training loops (bit-exact integer work, host side)."""
import numpy as np


def macro_f1(labels: np.ndarray, preds: np.ndarray) -> float:
    """sklearn.metrics.f1_score(labels, preds, average="macro") as called at reference src/train.py:86
    (label set = union of y_true and y_pred, zero_division -> 0)."""
    labels = np.asarray(labels).reshape(-1)
    preds = np.asarray(preds).reshape(-1)
    scores = []
    for c in np.union1d(labels, preds):
        tp = float(np.sum((preds == c) & (labels == c)))
        fp = float(np.sum((preds == c) & (labels != c)))
        fn = float(np.sum((preds != c) & (labels == c)))
        den = 2.0 * tp + fp + fn
        scores.append(0.0 if den == 0 else 2.0 * tp / den)
    return float(np.mean(scores)) if scores else 0.0
