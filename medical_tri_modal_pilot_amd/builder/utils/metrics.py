"""Evaluator of the reference (builder/utils/metrics.py:26-108) without its hard ``.cuda()`` calls and
without the torchmetrics dependency (absent here; unpinned in the reference's requirements).

The reference computes, over every (target, sigmoid(output)) pair fed by the "test" branch of the trainer
(trainer.py:192-240 -> ``logger.evaluator.add_batch``):
  * ``AUROC(task="binary")`` and ``AveragePrecision(task="binary")`` of torchmetrics with ``thresholds=None``,
    i.e. the exact curves over the distinct prediction values (ties share one threshold) -- the same
    definitions as scikit-learn's ``roc_auc_score`` / ``average_precision_score``, which is what the tests
    pin this implementation against;
  * a "best F1 over thresholds 0.01 .. 0.99" loop (metrics.py:76-83) whose ``temp_output = preds.detach()``
    aliases ``preds``: the first pass (threshold 0.01) binarises the predictions IN PLACE, so every later pass
    sees 0/1 values and scores the same.  The value the reference reports is therefore F1 at threshold 0.01
    (``pred >= 0.01``); that is what ``performance_metric`` returns.  ``best_f1_over_thresholds`` is the sweep
    the loop was presumably meant to be.
Everything runs on the device the predictions live on (one sort + prefix sums: rocPRIM on an MI355X).
"""
from typing import List

import numpy as np
import torch


def _clf_curve(preds: torch.Tensor, target: torch.Tensor):
    """True / false positive counts at every distinct prediction value, thresholds descending."""
    preds = preds.reshape(-1).double()
    target = target.reshape(-1).double()
    order = torch.argsort(preds, descending=True)
    p, t = preds[order], target[order]
    tps_all = torch.cumsum(t, 0)
    fps_all = torch.cumsum(1.0 - t, 0)
    if p.numel() == 0:
        return tps_all, fps_all, p
    last = torch.ones_like(p, dtype=torch.bool)
    last[:-1] = p[1:] != p[:-1]                      # last element of every tie group
    return tps_all[last], fps_all[last], p[last]


def binary_auroc(preds: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Area under the ROC curve (trapezoids over the distinct thresholds); 0 when a class is absent, like
    torchmetrics (which warns and returns a zero curve)."""
    tps, fps, _ = _clf_curve(preds, target)
    if tps.numel() == 0 or tps[-1] <= 0 or fps[-1] <= 0:
        return torch.zeros((), dtype=torch.float32, device=preds.device)
    zero = tps.new_zeros(1)
    tpr = torch.cat([zero, tps / tps[-1]])
    fpr = torch.cat([zero, fps / fps[-1]])
    return torch.trapz(tpr, fpr).float()


def binary_average_precision(preds: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """AP = sum_n (R_n - R_{n-1}) P_n over the distinct thresholds; NaN without positives (0/0 recall)."""
    tps, fps, _ = _clf_curve(preds, target)
    if tps.numel() == 0:
        return torch.full((), float("nan"), dtype=torch.float32, device=preds.device)
    precision = tps / (tps + fps)
    recall = tps / tps[-1]
    prev = torch.cat([recall.new_zeros(1), recall[:-1]])
    return ((recall - prev) * precision).sum().float()


def binary_f1(preds: torch.Tensor, target: torch.Tensor, threshold: float) -> torch.Tensor:
    hard = (preds.reshape(-1) >= threshold).double()
    t = target.reshape(-1).double()
    tp = (hard * t).sum()
    denom = hard.sum() + t.sum()
    return torch.where(denom > 0, 2.0 * tp / denom.clamp_min(1.0), torch.zeros_like(tp)).float()


def best_f1_over_thresholds(preds: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """max over thresholds 0.01 .. 0.99 of F1(pred >= threshold): one [99, n] comparison instead of 99 passes."""
    thr = torch.arange(1, 100, device=preds.device, dtype=torch.float64) / 100.0
    p = preds.reshape(1, -1).double()
    t = target.reshape(1, -1).double()
    hard = (p >= thr[:, None]).double()
    tp = (hard * t).sum(1)
    denom = hard.sum(1) + t.sum()
    f1 = torch.where(denom > 0, 2.0 * tp / denom.clamp_min(1.0), torch.zeros_like(tp))
    return f1.max().float()


class Evaluator(object):
    def __init__(self, args):
        self.args = args
        self.n_labels = args.output_dim
        self.confusion_matrix = np.zeros((self.n_labels, self.n_labels))
        self.batch_size = args.batch_size
        if args.model_types == "classification" and args.loss_types == "rmse":
            self.best_auc = float("inf")
        else:
            self.best_auc = 0
        self.labels_list = [i for i in range(self.n_labels)]
        self.y_true_multi: List[torch.Tensor] = []
        self.y_pred_multi: List[torch.Tensor] = []
        self.rmse: List[torch.Tensor] = []

    def add_batch(self, y_true, y_pred_multi, rmse=None):
        self.y_pred_multi.append(y_pred_multi.detach())
        self.y_true_multi.append(y_true.detach())
        if rmse is not None:
            self.rmse.append(rmse.detach())

    def performance_metric(self):
        """[auc, apr, f1(, rmse)] rounded to 4 decimals like metrics.py:84-93.  Batches are concatenated (the
        reference ``torch.stack``s them, which needs equal batch sizes; cat accepts a short last batch too)."""
        trues = torch.cat([t.reshape(-1) for t in self.y_true_multi]).to(torch.uint8)
        preds = torch.nan_to_num(torch.cat([p.reshape(-1).float() for p in self.y_pred_multi]))
        auc = binary_auroc(preds, trues)
        apr = binary_average_precision(preds, trues)
        f1 = binary_f1(preds, trues, 0.01)               # what the aliasing loop of metrics.py:76-83 evaluates to
        vals = [auc.cpu().numpy(), apr.cpu().numpy(), f1.cpu().numpy()]
        if "rmse" in self.args.auxiliary_loss_type:
            vals.append(torch.mean(torch.stack([r.reshape(()).float() for r in self.rmse])).cpu().numpy())
        return list(np.round(np.array(vals, dtype=np.float64), 4))

    def reset(self):
        self.confusion_matrix = np.zeros((self.n_labels, self.n_labels))
        self.y_true_multi = []
        self.y_pred_multi = []
        self.rmse = []
