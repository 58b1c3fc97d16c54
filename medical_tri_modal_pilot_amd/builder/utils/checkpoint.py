"""Checkpoints in the reference's format (``Logger.save``, builder/utils/logger.py:166-177; loaders at
2_train.py:86-102 and :304-314):

    {'model': model.state_dict(), 'optimizer': optimizer.state_dict(), 'best_step': step, 'last_step': last,
     'score': best_auc, 'epoch': epoch}

``model`` needs nothing special -- TRI_MBT_VSLTCLS keeps the reference's 313 keys and shapes.  ``optimizer`` is the
state_dict of ``torch.optim.AdamW(model.parameters())`` (2_train.py:110): moments indexed by a parameter's position
in ``model.parameters()``.  ``optim.FusedAdamW`` keeps its moments in flat buffers over ``model.hot_parameters()``
(a subset, in another order), so the two helpers below translate in both directions; a reference-trained checkpoint
resumes on the MI355X path and vice versa.
"""
from typing import Optional

import torch

_ADAMW_GROUP_DEFAULTS = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                             fused=None)


def optimizer_state_as_reference(model: torch.nn.Module, optimizer) -> dict:
    """``optimizer.state_dict()`` in the layout of ``torch.optim.AdamW(model.parameters())``."""
    if not hasattr(optimizer, "flat"):
        return optimizer.state_dict()
    params = list(model.parameters())
    pos = {id(p): i for i, p in enumerate(params)}
    flat, state = optimizer.flat, {}
    if optimizer.step_count > 0:
        for j, p in enumerate(flat.params):
            lo, hi = flat.slice_of(j)
            state[pos[id(p)]] = {"step": torch.tensor(float(optimizer.step_count)),
                                 "exp_avg": optimizer.exp_avg[lo:hi].view_as(p).clone(),
                                 "exp_avg_sq": optimizer.exp_avg_sq[lo:hi].view_as(p).clone()}
    group = {k: v for k, v in optimizer.param_groups[0].items() if k != "params"}
    for k, v in _ADAMW_GROUP_DEFAULTS.items():
        group.setdefault(k, v)
    group["params"] = list(range(len(params)))
    return {"state": state, "param_groups": [group]}


def load_reference_optimizer_state(model: torch.nn.Module, optimizer, state_dict: dict) -> None:
    """Inverse of the above: moments / step / hyper-parameters of a ``torch.optim.AdamW(model.parameters())``
    state_dict into ``optimizer`` (FusedAdamW: into its flat buffers; anything else: ``load_state_dict``)."""
    if not hasattr(optimizer, "flat"):
        optimizer.load_state_dict(state_dict)
        return
    params = list(model.parameters())
    flat, steps = optimizer.flat, set()
    with torch.no_grad():
        optimizer.exp_avg.zero_()
        optimizer.exp_avg_sq.zero_()
        for idx, st in state_dict["state"].items():
            p = params[int(idx)]
            j = flat.index_of.get(id(p))
            if j is None:
                continue                  # a parameter the hot path never trains (frozen Swin, unused heads)
            lo, hi = flat.slice_of(j)
            optimizer.exp_avg[lo:hi].copy_(st["exp_avg"].reshape(-1))
            optimizer.exp_avg_sq[lo:hi].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError(f"parameters were stepped a different number of times ({sorted(steps)}): the fused AdamW "
                         "keeps one step count for its flat buffer")
    optimizer.step_count = steps.pop() if steps else 0
    g = state_dict["param_groups"][0]
    for k in ("lr", "betas", "eps", "weight_decay"):
        if k in g:
            optimizer.param_groups[0][k] = g[k]


def make_checkpoint(model, optimizer, step, epoch, score, last=None) -> dict:
    """The dict ``Logger.save`` writes (logger.py:167)."""
    return {"model": model.state_dict(), "optimizer": optimizer_state_as_reference(model, optimizer), "best_step": step,
            "last_step": last, "score": score, "epoch": epoch}


def load_checkpoint(ckpt, model, optimizer=None, map_location: Optional[str] = None):
    """``ckpt``: path or dict.  Loads the model strictly (2_train.py:97,312-314) and, if given, the optimizer;
    returns (score, epoch) like the resume branch of 2_train.py:99-100."""
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt, map_location=map_location)
    model.load_state_dict({k: v for k, v in ckpt["model"].items()})
    if optimizer is not None and ckpt.get("optimizer") is not None:
        load_reference_optimizer_state(model, optimizer, ckpt["optimizer"])
    gs = getattr(model, "_mtmp_graph_step", None)
    if gs is not None:
        gs.invalidate()                   # captured graphs hold pointers to weight casts made before the load
    return ckpt.get("score"), ckpt.get("epoch")
