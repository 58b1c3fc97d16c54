"""Next-row components on CPU: the Evaluator's metrics against scikit-learn (the published
definitions torchmetrics' exact binary AUROC / AP follow) and the checkpoint translation between
torch.optim.AdamW(model.parameters()) and the flat-buffer FusedAdamW."""
import math
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from sklearn.metrics import average_precision_score, f1_score, roc_auc_score

from medical_tri_modal_pilot_amd.builder.utils import checkpoint as C
from medical_tri_modal_pilot_amd.builder.utils import metrics as M
from medical_tri_modal_pilot_amd.optim import FusedAdamW


def _args(**kw):
    d = dict(output_dim=1, batch_size=8, model_types="detection", loss_types="bce", auxiliary_loss_type="None")
    d.update(kw)
    return SimpleNamespace(**d)


@pytest.mark.parametrize("n,ties", [(50, False), (1000, False), (1000, True), (4097, True)])
def test_auroc_ap_match_sklearn(n, ties):
    g = torch.Generator().manual_seed(n + int(ties))
    y = (torch.rand(n, generator=g) < 0.3).to(torch.uint8)
    p = torch.sigmoid(torch.randn(n, generator=g) + 1.5 * y.float())
    if ties:
        p = (p * 20).round() / 20                       # many tied predictions
    assert abs(float(M.binary_auroc(p, y)) - roc_auc_score(y.numpy(), p.numpy())) < 1e-6
    assert abs(float(M.binary_average_precision(p, y)) - average_precision_score(y.numpy(), p.numpy())) < 1e-6


def test_degenerate_targets():
    p = torch.tensor([0.2, 0.7, 0.7, 0.9])
    assert float(M.binary_auroc(p, torch.zeros(4))) == 0.0          # torchmetrics: zero curve when a class is absent
    assert float(M.binary_auroc(p, torch.ones(4))) == 0.0
    assert math.isnan(float(M.binary_average_precision(p, torch.zeros(4))))
    assert float(M.binary_average_precision(p, torch.ones(4))) == pytest.approx(1.0)
    assert float(M.binary_f1(p, torch.zeros(4), 0.95)) == 0.0


def test_f1_is_the_aliased_loop_value_and_sweep_is_bruteforce():
    g = torch.Generator().manual_seed(5)
    y = (torch.rand(600, generator=g) < 0.2).to(torch.uint8)
    p = torch.sigmoid(2 * torch.randn(600, generator=g) + 2.0 * y.float() - 1.0)
    # the reference loop (metrics.py:76-83) with its aliasing, restated literally
    work, best = p.clone(), 0.0
    for i in range(1, 100):
        t = i / 100.0
        tmp = work                                      # preds.detach() shares storage
        tmp[tmp >= t] = 1
        tmp[tmp < t] = 0
        best = max(best, f1_score(y.numpy(), (tmp > t).numpy().astype(int), zero_division=0))
    assert abs(float(M.binary_f1(p, y, 0.01)) - best) < 1e-6
    brute = max(f1_score(y.numpy(), (p >= i / 100.0).numpy().astype(int), zero_division=0) for i in range(1, 100))
    assert abs(float(M.best_f1_over_thresholds(p, y)) - brute) < 1e-6


def test_evaluator_contract():
    ev = M.Evaluator(_args())
    assert ev.best_auc == 0 and M.Evaluator(_args(model_types="classification", loss_types="rmse")).best_auc == float("inf")
    g = torch.Generator().manual_seed(1)
    ys, ps = [], []
    for nb in (8, 8, 5):                                # a short last batch is accepted
        y = (torch.rand(nb, generator=g) < 0.4).float()
        p = torch.sigmoid(torch.randn(nb, generator=g) + y)
        ev.add_batch(y, p)
        ys.append(y)
        ps.append(p)
    ps[0][0] = float("nan")                             # nan_to_num like metrics.py:67
    ev.y_pred_multi[0][0] = float("nan")
    res = ev.performance_metric()
    yy, pp = torch.cat(ys).numpy(), torch.nan_to_num(torch.cat(ps)).numpy()
    assert res[0] == round(roc_auc_score(yy, pp), 4) and res[1] == round(average_precision_score(yy, pp), 4)
    assert len(res) == 3 and all(isinstance(float(v), float) for v in res)
    ev.reset()
    assert ev.y_true_multi == [] and ev.y_pred_multi == []
    ev2 = M.Evaluator(_args(auxiliary_loss_type="rmse"))
    ev2.add_batch(torch.tensor([0., 1.]), torch.tensor([0.2, 0.8]), rmse=torch.tensor(0.5))
    assert ev2.performance_metric()[3] == 0.5


def _tiny():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 4), torch.nn.Linear(4, 1))


def test_checkpoint_optimizer_state_round_trip():
    ref_model = _tiny()
    ref_model[2].weight.requires_grad_(False)           # a frozen tensor in the middle (like the Swin encoder)
    ref_opt = torch.optim.AdamW(ref_model.parameters(), lr=3e-4, weight_decay=1e-6)
    x = torch.randn(5, 6)
    for _ in range(3):
        ref_opt.zero_grad()
        ref_model(x).sum().backward()
        ref_opt.step()
    ckpt = {"model": ref_model.state_dict(), "optimizer": ref_opt.state_dict(), "best_step": 3, "last_step": None,
            "score": 0.71, "epoch": 2}
    model = _tiny()
    model[2].weight.requires_grad_(False)
    hot = [(n, p) for n, p in model.named_parameters() if p.requires_grad][::-1]      # another order than parameters()
    opt = FusedAdamW(hot, lr=1e-3)
    score, epoch = C.load_checkpoint(ckpt, model, opt)
    assert (score, epoch) == (0.71, 2) and opt.step_count == 3 and opt.param_groups[0]["lr"] == 3e-4
    for a, b in zip(model.parameters(), ref_model.parameters()):
        assert torch.equal(a, b)
    back = C.optimizer_state_as_reference(model, opt)
    ref_sd = ref_opt.state_dict()
    assert set(back["state"].keys()) == set(ref_sd["state"].keys())
    for k, st in ref_sd["state"].items():
        assert torch.equal(back["state"][k]["exp_avg"], st["exp_avg"])
        assert torch.equal(back["state"][k]["exp_avg_sq"], st["exp_avg_sq"])
        assert float(back["state"][k]["step"]) == float(st["step"])
    assert back["param_groups"][0]["params"] == list(range(len(list(model.parameters()))))
    # and torch's own AdamW accepts what the fused optimizer exports
    opt3 = torch.optim.AdamW(model.parameters(), lr=1.0)
    opt3.load_state_dict(C.make_checkpoint(model, opt, 3, 2, 0.71)["optimizer"])
    assert opt3.param_groups[0]["lr"] == 3e-4
    assert torch.equal(opt3.state[list(model.parameters())[0]]["exp_avg"], ref_sd["state"][0]["exp_avg"])
