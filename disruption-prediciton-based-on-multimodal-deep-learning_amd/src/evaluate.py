"""Test-set evaluation -- mirror of the reference's ``src/evaluate.py:11-134`` (same signature and return value).

Decision rule (:56-58): a sample is labelled 1 ("normal") unless softmax(output)[:, 0] > threshold.  Predictions, labels and
the loss stay on the GPU until the loop is over (the reference calls ``.item()`` twice per batch); accuracy, macro-F1, ROC-AUC
and the classification report are then host arithmetic on N integers.  The confusion-matrix / ROC / PR figure is presentation
and is not drawn (``save_conf`` is accepted and ignored); the text report is written to ``save_txt`` as in the reference."""
from typing import Literal, Optional

import numpy as np
import torch
from torch.utils.data import DataLoader

from .train import _forward
from .utils.metrics import macro_f1


def threshold_predictions(output: torch.Tensor, threshold: float) -> torch.Tensor:
    p0 = torch.nn.functional.softmax(output, dim=1)[:, 0]
    return torch.logical_not(p0 > torch.tensor([threshold], dtype=torch.float32, device=output.device))


def _report(labels: np.ndarray, preds: np.ndarray) -> str:
    rows = ["{:>12s} {:>9s} {:>9s} {:>9s} {:>9s}".format("", "precision", "recall", "f1-score", "support"), ""]
    for c in (0, 1):
        tp = float(np.sum((preds == c) & (labels == c))); fp = float(np.sum((preds == c) & (labels != c)))
        fn = float(np.sum((preds != c) & (labels == c)))
        pr = tp / (tp + fp) if tp + fp else 0.0
        rc = tp / (tp + fn) if tp + fn else 0.0
        f1 = 2 * pr * rc / (pr + rc) if pr + rc else 0.0
        rows.append("{:>12d} {:9.2f} {:9.2f} {:9.2f} {:9d}".format(c, pr, rc, f1, int(np.sum(labels == c))))
    return "\n".join(rows) + "\n"


def _binary_auc(labels: np.ndarray, scores: np.ndarray) -> float:
    """roc_auc_score for 0/1 labels (rank statistic with midranks); nan if only one class is present."""
    pos, neg = scores[labels == 1], scores[labels == 0]
    if len(pos) == 0 or len(neg) == 0:
        return float("nan")
    order = np.argsort(np.concatenate([pos, neg]), kind="mergesort")
    s = np.concatenate([pos, neg])[order]
    ranks = np.empty(len(s))
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1
        i = j + 1
    r = np.empty(len(s)); r[order] = ranks
    return float((r[:len(pos)].sum() - len(pos) * (len(pos) + 1) / 2) / (len(pos) * len(neg)))


def evaluate(test_loader: DataLoader, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer],
             loss_fn: Optional[torch.nn.Module] = None, device: Optional[str] = "cuda:0",
             save_conf: Optional[str] = None, save_txt: Optional[str] = None, threshold: float = 0.5,
             model_type: Literal["single", "multi", "multi-GB"] = "single"):
    if device is None:
        device = torch.device("cuda:0")
    model.to(device)
    model.eval()
    loss_sum, correct, n_batches, total_size = None, None, 0, 0
    total_pred, total_label = [], []
    for data, target in test_loader:
        with torch.no_grad():
            if optimizer is not None:
                optimizer.zero_grad()
            output, output_vis, output_ts = _forward(model, data, device, model_type)
            tgt = target.to(device)
            loss = loss_fn(output, output_vis, output_ts, tgt) if model_type == "multi-GB" else loss_fn(output, tgt)
            loss_sum = loss.detach() if loss_sum is None else loss_sum + loss.detach()
            pred = threshold_predictions(output, threshold)
            c = pred.eq(tgt.view_as(pred)).sum()
            correct = c if correct is None else correct + c
            total_size += pred.size(0)
            n_batches += 1
            total_pred.append(pred.view(-1, 1))
            total_label.append(tgt.view(-1, 1))
    preds = torch.concat(total_pred, dim=0).view(-1).cpu().numpy()
    labels = torch.concat(total_label, dim=0).view(-1).cpu().numpy()
    test_loss = float(loss_sum.item()) / n_batches                                       # :77  (mean over batches)
    test_acc = int(correct.item()) / total_size
    preds = np.nan_to_num(preds, copy=True, nan=0, posinf=1.0, neginf=0)                 # :81
    preds = np.where(preds > 1 - threshold, 1, 0)                                        # :83  (a no-op on 0/1 values)
    test_f1 = macro_f1(labels, preds)
    test_auc = _binary_auc(labels, preds.astype(np.float64))
    report = _report(labels, preds)
    print("############### Classification Report ####################")
    print(report)
    print("\n# test acc : {:.2f}, test f1 : {:.2f}, test AUC : {:.2f}, test loss : {:.3f}".format(test_acc, test_f1, test_auc, test_loss))
    if save_txt:
        with open(save_txt, "w") as f:
            f.write(report)
            f.write("\n# test score : {:.2f}, test loss : {:.3f}, test f1 : {:.3f}, test_auc : {:.3f}".format(
                test_acc, test_loss, test_f1, test_auc))
    return test_loss, test_acc, test_f1
