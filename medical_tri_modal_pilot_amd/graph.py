"""hipGraph replay of the training step's device work (zero_grad + forward + BCE + backward).

Why: at config 2 the step launches ~750 kernels; one MI355X finishes them in less time than one
CPU core needs to enqueue them through Python (bench.py: host_enqueue_ms_per_step), so the eager
step is host-bound.  The step's launch sequence is static for a given input shape -- ragged
batches only change *kv_len tensors*, which the kernels read from HBM -- so it is captured once per
shape and replayed with one hipGraphLaunch.

What is inside the graph: the advance of the dropout step word (ops.set_seed_word), the gradient
memset, the forward (HIP kernels on three streams, fork/join captured), the loss, the backward
(direct gradient writes into optim.FlatParams.grad).  What stays outside: host->device copies of
the batch into the static input buffers, missing_to_num (CPU, trainer.py:53-77), the gradient
all-reduce (ddp.GradReducer, world > 1), AdamW (one launch; lr / bias corrections are host
scalars that change every step), the scheduler, and loss.item().

Anything that would make a replay differ from an eager step is keyed or refused:
  * one graph per input signature (names, shapes, dtypes); at most ``max_graphs`` live (LRU);
  * the first ``warmup`` calls of a signature run eagerly (lazy initialisation must not be captured);
  * a failed capture disables graphs for this object and the step runs eagerly (never silently wrong:
    capture either succeeds completely or the eager path is used).
"""
import os
import warnings
from collections import OrderedDict
from typing import Callable, Dict

import torch

from . import ops

_GOLDEN = 0x9E3779B1          # odd increment of the step word
_SYNC_AFTER_REPLAY = os.environ.get("MTMP_GRAPH_SYNC", "") != ""        # diagnostics: host-wait after every replay


class GraphedTrainStep:
    def __init__(self, device: torch.device, max_graphs: int = 8, warmup: int = 1):
        if device.type != "cuda":
            raise RuntimeError("hipGraph capture needs a GPU device")
        self.device = device
        self.max_graphs, self.warmup = max_graphs, warmup
        self.entries: "OrderedDict[tuple, dict]" = OrderedDict()
        self.disabled = False
        self.seed_word = torch.full((1,), torch.initial_seed() & 0x7FFFFFFF, dtype=torch.int64, device=device)
        ops.set_seed_word(self.seed_word)
        # Warm-up steps and the capture run on ONE dedicated stream: autograd's AccumulateGrad nodes remember the
        # stream they were created on, and a node left on the (non-capturing) default stream breaks the capture.
        self.stream = torch.cuda.Stream(device=device)
        self.pool = None
        self.captures = 0
        self.replays = 0

    @staticmethod
    def signature(inputs: Dict[str, torch.Tensor]) -> tuple:
        return tuple((k, tuple(v.shape), v.dtype) for k, v in inputs.items())

    def invalidate(self):
        """Drop every captured graph (call after load_state_dict / any change of module structure or flags)."""
        self.entries.clear()

    def _capture(self, ent: dict, inputs: Dict[str, torch.Tensor], fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor],
                 params):
        static = {k: v.detach().clone() for k, v in inputs.items()}
        ops.bump_fused_epoch()      # per-layer derived weights (W2^T, casts) must be re-made INSIDE the graph
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(self.device)
        # thread_local: other threads (autograd workers, the RCCL watchdog) may keep making HIP calls during capture
        with torch.cuda.graph(g, pool=self.pool, stream=self.stream, capture_error_mode="thread_local"):
            self.seed_word.add_(_GOLDEN)
            loss = fn(static)
        if self.pool is None:
            self.pool = g.pool()
        ent.update(graph=g, static=static, loss=loss, fresh=True)
        self.captures += 1

    def _eager_on_side_stream(self, inputs, fn):
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.seed_word.add_(_GOLDEN)
            loss = fn(inputs)
        cur.wait_stream(self.stream)
        for v in inputs.values():
            v.record_stream(self.stream)
        return loss

    def run(self, inputs: Dict[str, torch.Tensor], fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor],
            params=None) -> torch.Tensor:
        """fn(inputs) -> scalar loss tensor; must do zero_grad + forward + backward with no host sync.
        Returns the loss tensor of this step (a static buffer when replayed)."""
        if self.disabled:
            return fn(inputs)
        key = self.signature(inputs)
        ent = self.entries.get(key)
        if ent is None:
            ent = self.entries[key] = {"seen": 0}
            while len(self.entries) > self.max_graphs:
                self.entries.popitem(last=False)
        self.entries.move_to_end(key)
        if "graph" not in ent:
            if ent["seen"] < self.warmup:
                ent["seen"] += 1
                return self._eager_on_side_stream(inputs, fn)
            try:
                self._capture(ent, inputs, fn, params)
            except Exception as e:                      # noqa: BLE001 -- any capture failure -> eager
                self.disabled = True
                self.entries.clear()
                torch.cuda.synchronize(self.device)
                warnings.warn(f"hipGraph capture of the training step failed ({type(e).__name__}: {e}); "
                              "continuing with eager launches")
                return fn(inputs)
        cur = torch.cuda.current_stream(self.device)
        fresh = ent.pop("fresh", False)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):           # replay on the capture stream, joined to the caller's by events
            if not fresh:
                # one multi-tensor copy for all same-device inputs (a dozen small D2D copy_ calls cost ~120 us of host
                # time per step, all of it GPU-idle time right after the previous step's loss.item())
                dst, src = [], []
                for k, v in inputs.items():
                    d = ent["static"][k]
                    if v.device == d.device and v.dtype == d.dtype and v.numel() > 0:
                        dst.append(d)
                        src.append(v)
                    elif v.numel() > 0:
                        d.copy_(v, non_blocking=True)
                    v.record_stream(self.stream)
                if dst:
                    torch._foreach_copy_(dst, src)
            ent["graph"].replay()
        cur.wait_stream(self.stream)
        # Why not simply graph.replay() on the caller's stream: measured on ROCm 7.2 / torch 2.10 (bench.py, one-stream
        # graph, --warmup 10), a graph launched on the DEFAULT stream let work enqueued on that stream right after
        # hipGraphLaunch (the AdamW kernel) overtake the graph's tail -- the loss went NaN within ~15 steps.  Launched
        # on the capture stream and joined by events as above, 60 replayed steps are bit-identical to eager ones in
        # every stream configuration.  MTMP_GRAPH_SYNC=1 additionally host-waits for each replay (diagnostics; costs
        # ~0.7 ms/step because the next step's host work no longer overlaps the GPU).
        if _SYNC_AFTER_REPLAY:
            self.stream.synchronize()
        self.replays += 1
        return ent["loss"]
