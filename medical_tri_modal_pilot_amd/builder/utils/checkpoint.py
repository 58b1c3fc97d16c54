"""Checkpoints in the reference's format (``Logger.save``, builder/utils/logger.py:166-177; loaders at
2_train.py:86-102 and :304-314):

    {'model': model.state_dict(), 'optimizer': optimizer.state_dict(), 'best_step': step, 'last_step': last,
     'score': best_auc, 'epoch': epoch}

``model`` needs nothing special -- TRI_MBT_VSLTCLS keeps the reference's 313 keys and shapes.  ``optimizer`` is the
state_dict of ``torch.optim.AdamW(model.parameters())`` (2_train.py:110): moments indexed by a parameter's position
in ``model.parameters()``.  ``optim.FusedAdamW`` keeps its moments in flat buffers over ``model.hot_parameters()``
(a subset, in another order), so FusedAdamW.state_dict() / load_state_dict() translate in both directions when the optimizer knows
``model.parameters()`` (``reference_params``); the helpers below supply it.  A reference-trained checkpoint resumes on the
MI355X path and vice versa.
"""
from typing import Optional

import torch

def _with_reference_layout(model, optimizer, fn):
    old = optimizer.reference_params
    optimizer.reference_params = list(model.parameters())
    try:
        return fn()
    finally:
        optimizer.reference_params = old


def optimizer_state_as_reference(model: torch.nn.Module, optimizer) -> dict:
    """``optimizer.state_dict()`` in the layout of ``torch.optim.AdamW(model.parameters())``."""
    if not hasattr(optimizer, "flat"):
        return optimizer.state_dict()
    return _with_reference_layout(model, optimizer, optimizer.state_dict)


def load_reference_optimizer_state(model: torch.nn.Module, optimizer, state_dict: dict) -> None:
    """Inverse of the above: moments / step / hyper-parameters of a ``torch.optim.AdamW(model.parameters())``
    state_dict into ``optimizer`` (FusedAdamW: into its flat buffers; anything else: ``load_state_dict``)."""
    if not hasattr(optimizer, "flat"):
        optimizer.load_state_dict(state_dict)
        return
    sd = dict(state_dict)
    sd["mtmp_layout"] = "reference"
    _with_reference_layout(model, optimizer, lambda: optimizer.load_state_dict(sd))


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
