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
  * a failed capture RAISES (the eager step is host-bound: 12 ms of enqueue for 10 ms of GPU work at config 2, so
    silently reporting it as the product would be wrong); ``fallback=True`` (--hip-graph-fallback 1) turns the failure
    into a warning and eager launches, and ``self.disabled`` / ``self.captures`` / ``self.replays`` say which ran.

Staged steps (data-parallel training): ``fn`` may be a LIST of stage callables ``stage(inputs, carry)``.  Each stage is
captured in its own graph (one shared memory pool, always replayed in order); ``carry`` hands tensors from one stage to
the next at capture time (stage 0 = zero_grad + forward + loss + backward of the last layers, stage k = backward of the
next group of layers down from the tensors stage k-1 left gradients on).  After a stage's replay has been enqueued,
``reducer.launch()`` starts the all-reduce of the gradient buckets that stage completed (recorded while the stage was
captured) on RCCL's side stream, beside the next stage's kernels -- no collective inside any graph.
"""
import time
import warnings
from collections import OrderedDict
from typing import Callable, Dict, List, Union

import torch

from . import ops
from .ddp import PlanMismatch

_GOLDEN = 0x9E3779B1          # odd increment of the step word
# Captured graphs are NEVER destroyed while the process lives.  On ROCm 7.2 / torch 2.10, hipGraphExecDestroy of one captured step
# (a dropped GraphedTrainStep, an entry evicted from its LRU) makes a LATER hipGraphLaunch of another one crash on the host in
# hip::Graph::UpdateStreams (rocgdb backtrace of `pytest tests -m gpu -k packed`, round 4: twenty graph tests in a row, each
# dropping its trainer, then a segfault in the replay of the next test's fresh graph; waiting for the device and collecting
# garbage at a quiet point between the tests did not help, holding on to the graphs did).  A training run captures a handful of
# graphs; their memory stays with the step's private pool anyway.
_ALIVE = []
# Because graphs cannot be released, captures are BUDGETED (ADVICE r4 / VERDICT r4): a GraphedTrainStep never evicts a captured
# signature (an evicted one would be captured again on its next visit, leaking a step's worth of pool memory per round trip) --
# past ``max_graphs`` signatures, and past MAX_ALIVE_GRAPHS captured graphs in the whole process (invalidate() parks the old
# ones), further signatures run eagerly (host-bound, correct).  All captures of one GraphedTrainStep share ONE memory pool: the
# blocks a capture frees at its end are reused by the next capture, so N shape buckets cost one step's activations plus N sets
# of static inputs / outputs, not N steps' worth (capture_log records the device memory reserved after every capture).
MAX_ALIVE_GRAPHS = 64


class GraphedTrainStep:
    def __init__(self, device: torch.device, max_graphs: int = 8, warmup: int = 1, fallback: bool = False):
        if device.type != "cuda":
            raise RuntimeError("hipGraph capture needs a GPU device")
        self.device = device
        self.max_graphs, self.warmup, self.fallback = max_graphs, warmup, fallback
        self.entries: "OrderedDict[tuple, dict]" = OrderedDict()
        self.disabled = False
        self.sync_after_replay = False       # diagnostics (tools/dbg): host-wait after every replay
        self.seed_word = torch.full((1,), torch.initial_seed() & 0x7FFFFFFF, dtype=torch.int64, device=device)
        ops.set_seed_word(self.seed_word)
        # Warm-up steps and the capture run on ONE dedicated stream: autograd's AccumulateGrad nodes remember the
        # stream they were created on, and a node left on the (non-capturing) default stream breaks the capture.
        self.stream = torch.cuda.Stream(device=device)
        self.pool = None
        self.captures = 0
        self.replays = 0
        self.eager_over_budget = 0           # steps of signatures past the capture budget (run eagerly)
        self.capture_log: List[dict] = []    # per capture: signature, device memory reserved before / after, graphs alive
        self._warned_budget = False
        # early loss hand-over (publish_loss / wait_loss): [loss bits, sequence number] on the device and in pinned host memory
        self.loss_dev = torch.zeros(2, dtype=torch.int32, device=device)
        self.loss_host = torch.zeros(2, dtype=torch.int32).pin_memory()
        self._loss_np = self.loss_host.numpy()
        self.expected_seq = 0                # publish_loss executions enqueued on the device so far
        self.published = False               # has a step function published (wait_loss is meaningful)?
        self._captured_publish = False

    def publish_loss(self, loss: torch.Tensor):
        """Call inside the step function right after the loss is computed (before backward): the value and a sequence number go
        to pinned host memory at THIS point of the stream -- one tiny kernel and an 8-byte copy, both capturable.  wait_loss()
        then returns the float as soon as the forward pass is done, while the backward and the optimizer step are still running:
        the host prepares and enqueues the next step meanwhile, and the GPU never idles at the step boundary (≈0.3 ms of host
        work per step otherwise).  Everything the host enqueues afterwards is stream-ordered behind the step as before."""
        v = loss.detach().reshape(1)
        if v.dtype != torch.float32:
            v = v.float()
        ops.call("mtmp_publish_scalar", ops._p(v), ops._p(self.loss_dev), ops._stream())
        self.loss_host.copy_(self.loss_dev, non_blocking=True)
        self.published = True
        if torch.cuda.is_current_stream_capturing():
            self._captured_publish = True    # (not executed now: every replay of this graph will)
        else:
            self.expected_seq += 1

    def wait_loss(self, timeout_s: float = 60.0) -> float:
        """The loss published by the step function of the last run() (host poll of the pinned pair; full device sync as a
        fallback after timeout_s).  NOTE for callers: this returns as soon as the FORWARD pass of the step has published
        its loss -- the backward, the gradient all-reduce and the optimizer step may still be running.  Work enqueued on
        the device afterwards is stream-ordered behind them as always; host-side timing or readers that are not stream
        ordered must synchronise themselves.  The pair is (value, sequence): the sequence word is polled first and the
        value is read again after it matched (the 8-byte copy is not assumed to land atomically); the poll yields the GIL
        between probes (loader / pin threads keep running)."""
        if not self.published:
            raise RuntimeError("wait_loss: the step function of the last run() did not call publish_loss")
        want, pair = self.expected_seq & 0xFFFFFFFF, self._loss_np
        t0 = time.perf_counter()
        spins = 0
        while True:
            while (int(pair[1]) & 0xFFFFFFFF) != want:
                spins += 1
                if spins & 63 == 0:
                    time.sleep(0)                      # yield the GIL
                    if time.perf_counter() - t0 > timeout_s:
                        torch.cuda.synchronize(self.device)
                        if (int(pair[1]) & 0xFFFFFFFF) != want:
                            raise RuntimeError(f"wait_loss: sequence number {int(pair[1])} after a device sync, expected {want}")
            v = float(pair[:1].view("float32")[0])
            if (int(pair[1]) & 0xFFFFFFFF) == want and float(pair[:1].view("float32")[0]) == v:
                return v

    @staticmethod
    def signature(inputs: Dict[str, torch.Tensor]) -> tuple:
        return tuple((k, tuple(v.shape), v.dtype) for k, v in inputs.items())

    def invalidate(self):
        """Forget every captured graph (call after load_state_dict / any change of module structure or flags).  The graphs
        themselves stay parked in _ALIVE (see above) and keep counting against MAX_ALIVE_GRAPHS."""
        self.entries.clear()

    def stats(self) -> dict:
        """What bench.py reports: captures, replays, eager steps past the budget, device memory reserved after the last capture."""
        last = self.capture_log[-1] if self.capture_log else {}
        return {"captures": self.captures, "replays": self.replays, "eager_over_budget": self.eager_over_budget,
                "signatures_captured": sum(1 for e in self.entries.values() if "graphs" in e),
                "graphs_alive_in_process": len(_ALIVE), "max_graphs": self.max_graphs,
                "reserved_bytes_after_last_capture": last.get("reserved_after"),
                "reserved_bytes_grown_by_captures": sum(c["reserved_after"] - c["reserved_before"] for c in self.capture_log)}

    def _budget_left(self, n_stages: int) -> bool:
        captured = sum(1 for e in self.entries.values() if "graphs" in e)
        return captured < self.max_graphs and len(_ALIVE) + n_stages <= MAX_ALIVE_GRAPHS

    @staticmethod
    def _stages(fn) -> List[Callable]:
        if isinstance(fn, (list, tuple)):
            return list(fn)
        return [lambda inputs, carry: carry.__setitem__("loss", fn(inputs))]

    def _capture(self, ent: dict, inputs: Dict[str, torch.Tensor], stages: List[Callable], reducer):
        static = {k: v.detach().clone() for k, v in inputs.items()}
        reserved_before = torch.cuda.memory_reserved(self.device)
        self._captured_publish = False
        ops.bump_fused_epoch()      # per-layer derived weights (W2^T, casts) must be re-made INSIDE the graph
        torch.cuda.synchronize(self.device)
        graphs, ready, carry = [], [], {}
        for i, stage in enumerate(stages):
            g = torch.cuda.CUDAGraph()
            _ALIVE.append(g)                     # (before the capture: a failed capture's graph is not destroyed either)
            # thread_local: other threads (autograd workers, the RCCL watchdog) may keep making HIP calls during capture
            with torch.cuda.graph(g, pool=self.pool, stream=self.stream, capture_error_mode="thread_local"):
                if i == 0:
                    self.seed_word.add_(_GOLDEN)
                stage(static, carry)
            if self.pool is None:
                self.pool = g.pool()
            graphs.append(g)
            # which gradient buckets this stage completed (the hooks ran while its Python was captured)
            ready.append(reducer.take_ready() if reducer is not None else [])
        if reducer is not None:
            reducer.assert_same_plan(ready)      # a LOCAL comparison with the plan the ranks agreed on (ddp._check_plan): no collective here
        ent.update(graphs=graphs, ready=ready, static=static, loss=carry["loss"], fresh=True, publishes=self._captured_publish)
        self.captures += 1
        self.capture_log.append({"signature": ent.get("key"), "stages": len(stages), "reserved_before": reserved_before,
                                 "reserved_after": torch.cuda.memory_reserved(self.device), "graphs_alive": len(_ALIVE)})

    def _eager_on_side_stream(self, inputs, stages, reducer):
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        carry = {}
        with torch.cuda.stream(self.stream):
            self.seed_word.add_(_GOLDEN)
            for stage in stages:
                stage(inputs, carry)
                if reducer is not None and reducer.staged:
                    # an eager stage ends with its modality side streams still running (only a CAPTURED stage has joined
                    # them at its end): the collective waits for each producer stream explicitly, not for this stream alone
                    reducer.launch(reducer.take_ready(), after=[self.stream, *reducer.extra_streams])
        cur.wait_stream(self.stream)
        for v in inputs.values():
            v.record_stream(self.stream)
        return carry["loss"]

    def run(self, inputs: Dict[str, torch.Tensor], fn: Union[Callable, List[Callable]], params=None,
            reducer=None, round_fp16=()) -> torch.Tensor:
        """fn(inputs) -> scalar loss tensor (zero_grad + forward + backward with no host sync), or a list of stage
        callables stage(inputs, carry) of which one stores carry["loss"] (see the module docstring).
        reducer: ddp.GradReducer in staged mode, or None.  round_fp16: keys of fp32 inputs the step must see as
        x.half().float() (the reference's fp16 storage of its event / time inputs): the rounding rides on the copy into the
        graph's static buffers instead of costing two eager launches per tensor in front of every step.  Returns the loss
        tensor of this step (a static buffer when replayed)."""
        stages = self._stages(fn)
        round_fp16 = frozenset(round_fp16)

        def rounded():                     # eager paths: the plain tensor ops
            return {k: (v.half().float() if k in round_fp16 else v) for k, v in inputs.items()}
        if self.disabled:
            return self._eager_on_side_stream(rounded(), stages, reducer)
        key = self.signature(inputs) + (len(stages),)
        ent = self.entries.get(key)
        if ent is None:
            ent = self.entries[key] = {"seen": 0, "key": key}
            # un-captured signatures are only counters: forget the oldest of them, never a captured one
            idle = [k for k, e in self.entries.items() if "graphs" not in e]
            while len(idle) > 4 * self.max_graphs:
                self.entries.pop(idle.pop(0))
        if "graphs" not in ent:
            if ent["seen"] < self.warmup:
                ent["seen"] += 1
                return self._eager_on_side_stream(rounded(), stages, reducer)
            if not self._budget_left(len(stages)):
                # the capture budget is spent (graphs cannot be released on this ROCm): this signature stays eager
                self.eager_over_budget += 1
                if not self._warned_budget:
                    self._warned_budget = True
                    warnings.warn(f"hipGraph capture budget spent ({self.max_graphs} signatures per trainer, {MAX_ALIVE_GRAPHS} "
                                  "graphs per process): further input shapes run as eager launches (host-bound); coarser "
                                  "length buckets (trainer.graph_len_bucket) keep a ragged loader inside the budget")
                return self._eager_on_side_stream(rounded(), stages, reducer)
            try:
                self._capture(ent, rounded(), stages, reducer)
            except PlanMismatch:
                raise                                   # never a fallback case: the ranks would pair different all-reduces
            except Exception as e:                      # noqa: BLE001
                self.entries.clear()
                torch.cuda.synchronize(self.device)
                if not self.fallback:
                    raise RuntimeError("hipGraph capture of the training step failed; run with --hip-graph 0 (eager "
                                       "launches, host-bound) or --hip-graph-fallback 1 to continue eagerly") from e
                self.disabled = True
                warnings.warn(f"hipGraph capture of the training step failed ({type(e).__name__}: {e}); "
                              "continuing with eager launches (--hip-graph-fallback 1)")
                return self._eager_on_side_stream(rounded(), stages, reducer)
        cur = torch.cuda.current_stream(self.device)
        fresh = ent.pop("fresh", False)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):           # replay on the capture stream, joined to the caller's by events
            if not fresh:
                # ONE launch for all same-device inputs (ops.copy_batch, fp16 rounding included): a dozen D2D copy_ calls and
                # six cast launches cost ~250 us of host time per step, all of it GPU-idle time right after the previous
                # step's loss.item()
                dst, src, r16 = [], [], []
                for k, v in inputs.items():
                    d, r = ent["static"][k], k in round_fp16
                    if v.numel() > 0 and ops.copy_batch_ok(d, v, r):
                        dst.append(d)
                        src.append(v)
                        r16.append(r)
                    elif v.numel() > 0:
                        d.copy_(v.half().float() if r else v, non_blocking=True)
                    v.record_stream(self.stream)
                for i in range(0, len(dst), ops.COPY_BATCH_MAX):
                    j = i + ops.COPY_BATCH_MAX
                    ops.copy_batch(dst[i:j], src[i:j], r16[i:j])
            if ent.get("publishes"):
                self.expected_seq += 1       # this replay executes the captured publish_loss once
                self.published = True
            for g, ids in zip(ent["graphs"], ent["ready"]):
                g.replay()
                if reducer is not None and ids:
                    reducer.launch(ids, after=self.stream)     # this stage's finished buckets, beside the next stage
        cur.wait_stream(self.stream)
        # Why not simply graph.replay() on the caller's stream: measured on ROCm 7.2 / torch 2.10 (bench.py, one-stream
        # graph, --warmup 10), a graph launched on the DEFAULT stream let work enqueued on that stream right after
        # hipGraphLaunch (the AdamW kernel) overtake the graph's tail -- the loss went NaN within ~15 steps.  Launched
        # on the capture stream and joined by events as above, 60 replayed steps are bit-identical to eager ones in
        # every stream configuration.  sync_after_replay additionally host-waits for each replay (diagnostics; costs
        # ~0.7 ms/step because the next step's host work no longer overlaps the GPU).
        if self.sync_after_replay:
            self.stream.synchronize()
        self.replays += 1
        return ent["loss"]
