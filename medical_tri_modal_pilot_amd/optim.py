"""Flat parameter / gradient storage and the fused AdamW of the hot path.

``FlatParams`` re-homes a list of parameters into ONE contiguous fp32 buffer (each tensor
16-byte aligned) and gives every parameter a gradient that is a view into ONE contiguous
gradient buffer, so that
  * AdamW (2_train.py:110) is a single launch of mtmp_adamw_step over ~12 M elements instead of
    a foreach over ~90 tensors, and
  * the DDP all-reduce (ddp.GradReducer) works on contiguous slices with no copy-in/copy-out.
Pure torch, device-agnostic (the gloo CPU tests use it); only ``FusedAdamW.step`` calls HIP.
"""
from typing import Iterable, List, Tuple

import torch

ALIGN = 4  # elements (16 bytes of fp32)


class FlatParams:
    def __init__(self, named_params: Iterable[Tuple[str, torch.nn.Parameter]]):
        self.names: List[str] = []
        self.params: List[torch.nn.Parameter] = []
        for n, p in named_params:
            if p.requires_grad:
                self.names.append(n)
                self.params.append(p)
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("FlatParams needs fp32 parameters on one device (call it after model.to(device))")
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.index_of = {id(p): i for i, p in enumerate(self.params)}
        self.written = set()          # parameter indices whose gradient slice was written directly since zero_grad
        self.accumulated = set()      # parameter indices autograd has accumulated into since zero_grad
        self.ready_cb = None          # ddp.GradReducer: called with an index when a slice was written directly
        self.zero_cb = None           # ddp.GradReducer: called by zero_grad() (a new accumulation starts)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.data[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p._mtmp_flat = self
        self.attach_grads()
        # a kernel may only OVERWRITE a gradient slice nothing has been accumulated into since zero_grad() (claim)
        self._acc_hooks = [p.register_post_accumulate_grad_hook(self._note_accumulated(i))
                           for i, p in enumerate(self.params)]

    def _note_accumulated(self, i: int):
        def hook(_param):
            if i not in self.written:      # (the hook also fires with an undefined gradient: ddp.GradReducer._make_hook)
                self.accumulated.add(i)
        return hook

    def attach_grads(self):
        """(Re)point every p.grad at its slice of the flat gradient buffer."""
        for p, o in zip(self.params, self.offsets):
            g = self.grad[o:o + p.numel()].view_as(p)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def zero_grad(self):
        if self.zero_cb is not None:
            self.zero_cb()
        self.grad.zero_()
        self.written.clear()
        self.accumulated.clear()
        self.attach_grads()

    # ---- direct writes by the backward kernels (ops.GradSink) ----
    def claim(self, idx) -> bool:
        """True if none of these gradient slices was written since the last zero_grad() and every p.grad is
        still the flat view (then a kernel may overwrite them); False -> caller returns gradients normally."""
        for i in idx:
            p = self.params[i]
            if (i in self.written or i in self.accumulated or p.grad is None
                    or p.grad.data_ptr() != self.grad.data_ptr() + 4 * self.offsets[i]):
                return False
        self.written.update(idx)
        return True

    def mark_ready(self, idx):
        if self.ready_cb is not None:
            for i in idx:
                self.ready_cb(i)

    def add_grad(self, p: torch.nn.Parameter, g: torch.Tensor):
        """Accumulate ``g`` into ``p.grad`` by hand (gradients obtained with torch.autograd.grad, which runs no
        AccumulateGrad node and no post-accumulate hook), with the bookkeeping those would have done."""
        if p.grad is None:
            p.grad = g.detach().clone()
        else:
            p.grad.add_(g)
        i = self.index_of.get(id(p))
        if i is not None:
            self.accumulated.add(i)
            self.mark_ready([i])

    def slice_of(self, i: int) -> Tuple[int, int]:
        return self.offsets[i], self.offsets[i] + self.params[i].numel()

    # ---- bf16 shadow copy, kept current by the AdamW kernel itself -------------------------------
    def enable_shadow(self):
        """Allocate the bf16 copy of the whole buffer that the MFMA kernels read.  FusedAdamW refreshes it inside
        the optimizer kernel, so a training step launches no weight-cast kernels; span() hands out views."""
        self.shadow = self.data.to(torch.bfloat16)
        self.shadow_versions = [p._version for p in self.params]

    def span(self, idxs, dtype):
        """A 1-D view over the adjacent parameters ``idxs`` in ``dtype`` (fp32: the master buffer; bf16: the
        shadow), or None if they are not adjacent / the shadow is missing or stale for one of them (parameter
        modified by anything but FusedAdamW, e.g. load_state_dict) -- the caller then casts the usual way."""
        lo = self.offsets[idxs[0]]
        end = lo
        for i in idxs:
            if self.offsets[i] != end:
                return None
            end += self.params[i].numel()
            if end % ALIGN and i != idxs[-1]:
                return None
        if dtype == torch.float32:
            for i in idxs:
                if self.params[i].data_ptr() != self.data.data_ptr() + 4 * self.offsets[i]:
                    return None
            return self.data[lo:end]
        shadow = getattr(self, "shadow", None)
        if dtype != torch.bfloat16 or shadow is None:
            return None
        for i in idxs:
            if self.params[i]._version != self.shadow_versions[i]:
                return None
        return shadow[lo:end]


def compute_weight(p: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """``p`` in the compute dtype: a view of the flat master / shadow buffer when there is a current one,
    otherwise a cast."""
    flat = getattr(p, "_mtmp_flat", None)
    if flat is not None:
        v = flat.span([flat.index_of[id(p)]], dtype)
        if v is not None:
            return v.view(p.shape)
    return p.detach().to(dtype)


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, eps outside the sqrt)
    over FlatParams, one HIP launch per step.  ``lr`` is read from ``param_groups[0]['lr']`` so the
    reference's CosineAnnealingWarmupRestarts drives it unchanged."""

    def __init__(self, named_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, reference_params=None):
        """reference_params: ``list(model.parameters())`` -- when given, state_dict() / load_state_dict() speak the
        layout of ``torch.optim.AdamW(model.parameters())`` (2_train.py:110), so the reference's unchanged
        ``Logger.save`` / resume code (logger.py:167, 2_train.py:98) round-trips the moments."""
        named_params = list(named_params)
        if named_params and not isinstance(named_params[0], (tuple, list)):
            named_params = [(f"p{i}", p) for i, p in enumerate(named_params)]
        self.flat = FlatParams(named_params)
        super().__init__(self.flat.params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.exp_avg = torch.zeros_like(self.flat.data)
        self.exp_avg_sq = torch.zeros_like(self.flat.data)
        self.flat.enable_shadow()
        self.step_count = 0
        self.reducer = None            # ddp.GradReducer, attached by the training script
        self.grad_scale = 1.0
        self.reference_params = None if reference_params is None else list(reference_params)

    # ---- checkpoints (logger.py:166-177 calls optimizer.state_dict(); 2_train.py:98 optimizer.load_state_dict()) ----
    def _positions(self, layout: str):
        """(position of every flat parameter in the state_dict's numbering, number of entries)."""
        if layout == "reference":
            if self.reference_params is None:
                raise ValueError("this optimizer state is indexed by model.parameters(): construct FusedAdamW with "
                                 "reference_params=list(model.parameters()) to load it")
            pos = {id(p): i for i, p in enumerate(self.reference_params)}
            return [pos[id(p)] for p in self.flat.params], len(self.reference_params)
        return list(range(len(self.flat.params))), len(self.flat.params)

    def state_dict(self):
        """The moments and the step count in torch.optim.AdamW's state_dict format: indexed like
        ``torch.optim.AdamW(model.parameters())`` when ``reference_params`` was given (parameters the hot path never
        trains have no entry, like parameters that never received a gradient in the reference), else by flat order."""
        layout = "reference" if self.reference_params is not None else "flat"
        posn, n = self._positions(layout)
        state = {}
        if self.step_count > 0:
            for j, p in enumerate(self.flat.params):
                lo, hi = self.flat.slice_of(j)
                state[posn[j]] = {"step": torch.tensor(float(self.step_count)),
                                  "exp_avg": self.exp_avg[lo:hi].view_as(p).clone(),
                                  "exp_avg_sq": self.exp_avg_sq[lo:hi].view_as(p).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        for k, v in dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                         fused=None).items():
            group.setdefault(k, v)
        group["params"] = list(range(n))
        return {"state": state, "param_groups": [group], "mtmp_layout": layout}

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        layout = state_dict.get("mtmp_layout")
        if layout is None:         # written by torch.optim.AdamW itself (a reference-trained checkpoint)
            layout = "reference" if self.reference_params is not None else "flat"
        posn, n = self._positions(layout)
        g = state_dict["param_groups"][0]
        if len(g["params"]) != n:
            raise ValueError(f"optimizer state has {len(g['params'])} parameters, expected {n} ({layout} layout)")
        where = {q: j for j, q in enumerate(posn)}
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps, seen, dropped = set(), set(), 0
        for idx, st in state_dict["state"].items():
            j = where.get(int(idx))
            if j is None:
                dropped += 1           # a parameter the hot path never trains (frozen Swin, unused heads) that WAS trained there
                continue
            lo, hi = self.flat.slice_of(j)
            self.exp_avg[lo:hi].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[lo:hi].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
            seen.add(j)
        if dropped:
            import warnings
            warnings.warn(f"optimizer state: {dropped} entries belong to parameters this path does not train (e.g. the image "
                          "encoder of bi_vsltimg_mbt_v1, which the reference trains) -- their moments are dropped", stacklevel=2)
        if steps and max(steps) > 0 and len(seen) < len(self.flat.params):
            # torch.optim.AdamW creates a parameter's state at its first gradient: a trained parameter WITHOUT an entry would
            # resume at step 0 there, but this optimizer keeps ONE step count for the flat buffer (its bias correction would be
            # the checkpoint's global step, with zero moments) -- refuse instead of resuming differently from the reference
            missing_names = [self.flat.names[j] if hasattr(self.flat, "names") else str(j)
                             for j in range(len(self.flat.params)) if j not in seen]
            raise ValueError(f"optimizer state at step {max(steps)} has no entry for {len(missing_names)} trained parameter(s) "
                             f"(e.g. {missing_names[:3]}): they would resume with another bias correction than in the reference")
        if len(steps) > 1:
            raise ValueError(f"parameters were stepped a different number of times ({sorted(steps)}): the fused AdamW "
                             "keeps one step count for its flat buffer")
        self.step_count = steps.pop() if steps else 0
        for k in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
            if k in g:
                self.param_groups[0][k] = g[k]

    def zero_grad(self, set_to_none: bool = True):
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        from . import ops
        if self.reducer is not None:
            self.reducer.wait()        # all-reduced sums are in flat.grad; the kernel applies 1/world
        g = self.param_groups[0]
        self.flat.attach_grads()
        self.step_count += 1
        fresh = all(p._version == v for p, v in zip(self.flat.params, self.flat.shadow_versions))
        ops.adamw_step(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.flat.shadow, g["lr"],
                       g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_count, self.grad_scale)
        # the kernel wrote through raw pointers: bump the version counters so that cached compute-dtype weight
        # copies are rebuilt on next use; the bf16 shadow was refreshed by the same kernel, so it stays current
        # (unless somebody else had modified a parameter since the last refresh: then re-cast everything once)
        torch._C._increment_version(self.flat.params)
        if not fresh:
            self.flat.shadow.copy_(self.flat.data)
        self.flat.shadow_versions = [p._version for p in self.flat.params]
        return None
