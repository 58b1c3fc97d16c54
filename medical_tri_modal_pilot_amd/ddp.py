"""Data-parallel gradient exchange for the tri-modal step (new: the reference has no multi-GPU
path).  One process per GPU; patient batches are sharded across ranks; the only
collective is the gradient all-reduce -- 12,121,601 fp32 values (48.5 MB) at 6 layers.

Design for xGMI (point-to-point links, ring collectives are per-link bound): few large
contiguous buckets cut from the flat gradient buffer of optim.FlatParams in REVERSE parameter
order (head -> last layer ... first layer -> embeddings = the order backward produces them).
A bucket's all-reduce runs on a side HIP stream, beside the rest of the backward; ``wait()``
(called by FusedAdamW.step) joins the side stream.  Sums are left un-averaged: the 1/world factor
is folded into the AdamW kernel's ``grad_scale``.  BatchNorm statistics stay per-rank (standard DDP).

Two ways a bucket gets launched:
  * hook mode (eager steps): as soon as every parameter of the bucket has its gradient
    (post-accumulate-grad hooks, or optim.FlatParams.mark_ready for gradients the backward kernels
    write straight into the flat buffer).  The side stream then waits for EVERYTHING enqueued so far
    on the stream the hook fired on, on the step's main compute stream and on the model's modality
    side streams: a bucket is completed by its last slice, but its other slices may have been
    written from another stream.
  * staged mode (hipGraph replays: no Python runs inside a replay): the step is captured as a few
    graphs cut at layer boundaries (graph.GraphedTrainStep stages); while a stage is captured (or
    run eagerly) the hooks only LOG which buckets became complete, and the trainer calls
    ``launch(ids, after=stream)`` right after enqueuing that stage's replay -- the all-reduce of the
    finished buckets then overlaps the next stage's backward.

Gradient accumulation: wrap every backward but the last of a step in ``no_sync()``; a gradient that
arrives for a bucket already reduced raises instead of being silently left un-reduced.
"""
import contextlib
import traceback
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from .optim import FlatParams


class PlanMismatch(RuntimeError):
    """The ranks of a staged step would issue different collectives (or this rank deviates from the agreed plan): never
    swallowed by GraphedTrainStep's capture fallback -- continuing would pair all-reduces of different ranges."""


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None):
    """Initial replica sync: parameters and buffers (BatchNorm running stats included) from rank `src`."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


class GradReducer:
    def __init__(self, flat: FlatParams, bucket_bytes: int = 8 << 20, group=None, overlap: bool = True):
        self.flat, self.group, self.overlap = flat, group, overlap
        self.world = dist.get_world_size(group)
        self.cuda = flat.grad.is_cuda
        self.side = torch.cuda.Stream(device=flat.grad.device) if self.cuda else None
        self.staged = False                      # staged mode (see module docstring): hooks log, launch() launches
        self.main_stream: Optional["torch.cuda.Stream"] = None     # the step's compute stream (set by begin_step)
        # other streams gradients are produced on (the encoder's modality side streams)
        self.extra_streams: List["torch.cuda.Stream"] = []
        # buckets: contiguous [lo, hi) element ranges, built from the END of the flat buffer
        self.buckets: List[List[int]] = []       # [lo, hi, n_params]
        self.bucket_of = [0] * len(flat.params)
        hi = flat.numel
        cur_lo, cur_n = hi, 0
        for i in reversed(range(len(flat.params))):
            cur_lo = flat.offsets[i]
            cur_n += 1
            self.bucket_of[i] = len(self.buckets)
            if (hi - cur_lo) * 4 >= bucket_bytes or i == 0:
                self.buckets.append([cur_lo, hi, cur_n])
                hi, cur_n = cur_lo, 0
        self.works = []
        self._sync = True
        self.last_issued: List[tuple] = []
        # the collectives of a staged step as ALL ranks agreed on them: set by the first staged step's wait() (see _check_plan)
        self.verified_plan = None
        self._reset()
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(flat.params)]
        flat.ready_cb = self._ready          # gradients written straight into the flat buffer (ops.GradSink)
        flat.zero_cb = self.begin_step       # zero_grad() opens a step

    def _reset(self):
        self.pending = [b[2] for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.ready_log: List[int] = []           # buckets in the order they became complete this step
        self.issued: List[tuple] = []            # [lo, hi) of every all-reduce issued this step, in order
        self.launch_log: List[List[int]] = []    # staged mode: the bucket ids of every launch() of this step, in order
        self.seen = {}                           # parameters that reported their gradient this step (-> who, if debug)

    def begin_step(self):
        """Called by FlatParams.zero_grad(): a new accumulation starts; the stream it is called on is the
        step's main compute stream."""
        if any(self.launched) or self.works:
            raise RuntimeError("GradReducer: zero_grad() while all-reduces of the previous step are in flight "
                               "(optimizer.step() / reducer.wait() was not called)")
        self._reset()
        if self.cuda:
            self.main_stream = torch.cuda.current_stream(self.flat.grad.device)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backwards inside this context neither count towards bucket completion nor
        launch anything (DistributedDataParallel.no_sync semantics); the LAST backward of the step runs outside."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _ready(self, i: int):
        if not self._sync:
            return
        b = self.bucket_of[i]
        if i in self.seen:
            raise RuntimeError(f"GradReducer: '{self.flat.names[i]}' reported its gradient twice in one step (a second "
                               "backward without no_sync(), or a kernel wrote it into the flat buffer AND autograd "
                               "accumulated into it)"
                               + (f"; first report from:\n{self.seen[i]}" if self.seen[i] else ""))
        self.seen[i] = "".join(traceback.format_stack(limit=8)) if getattr(self, "debug", False) else None
        if self.launched[b]:
            raise RuntimeError(f"GradReducer: a gradient for '{self.flat.names[i]}' arrived after its bucket was "
                               "all-reduced -- with gradient accumulation wrap every backward but the last in no_sync()")
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self.ready_log.append(b)
            if self.overlap and not self.staged:
                self._launch(b)

    def _make_hook(self, i: int):
        def hook(_param):
            # autograd runs a leaf's AccumulateGrad node -- and this hook -- even when the producing node returned None
            # for it: that is the case for every gradient a backward kernel wrote straight into the flat buffer, which
            # was reported through FlatParams.mark_ready already (counted twice, a bucket would go out before its last
            # slice is written)
            if i not in self.flat.written:
                self._ready(i)
        return hook

    def take_ready(self) -> List[int]:
        """Buckets that became complete since the last call (staged mode: the trainer asks after each stage)."""
        out, self.ready_log = self.ready_log, []
        return out

    def _join_producers(self, streams):
        """the collective stream waits for everything enqueued so far on `streams` (duplicates dropped)"""
        seen = set()
        for s in streams:
            if s is not None and s.cuda_stream not in seen:
                seen.add(s.cuda_stream)
                self.side.wait_stream(s)

    def _all_reduce(self, lo: int, hi: int, behind=None):
        """`behind`: the stream whose enqueued work the collective is ordered after (the process group's own stream waits for the
        CURRENT stream of the call); default: this reducer's side stream (which _join_producers made wait for the producers)"""
        buf = self.flat.grad[lo:hi]
        self.issued.append((lo, hi))
        if self.cuda:
            with torch.cuda.stream(self.side if behind is None else behind):
                self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _ranges(self, ids):
        """[lo, hi) element ranges of these buckets with neighbours merged: buckets are contiguous slices of the flat buffer, and
        the ones a backward stage completes are neighbours -- one larger collective instead of several (a ring all-reduce pays its
        latency per call and xGMI is per-link bound: fewer, larger messages)."""
        out = []
        for lo, hi, _ in sorted(self.buckets[b] for b in ids):
            if out and out[-1][1] == lo:
                out[-1][1] = hi
            else:
                out.append([lo, hi])
        return out

    def plan(self, ready_per_stage) -> List[List[List[int]]]:
        """The collectives a staged step issues: per stage that completed buckets the merged [lo, hi) ranges of those buckets, in
        issue order; whatever no stage completed goes out last, from wait().  (Stages without a finished bucket issue nothing
        and do not appear: the plan of a captured step, of its replays and of an eager staged step are then the same list.)"""
        done = {b for ids in ready_per_stage for b in ids}
        rest = [b for b in range(len(self.buckets)) if b not in done]
        return [self._ranges(ids) for ids in ready_per_stage if ids] + [self._ranges(rest)]

    def _check_plan(self, mine, where: str):
        """Every rank must issue the same collectives over the same ranges in the same order -- a rank whose stages completed
        other bucket sets would pair its all-reduces with the wrong ones of its peers: wrong sums or a hang.  The plan is a
        function of the model and of the stage cut, not of the batch shape, so it is agreed on ONCE, at a point every rank
        reaches together: the wait() of the first staged step (an eager warm-up step logs its stages exactly like a replay).
        That is the only collective of the check.  Everything later -- every further staged step, every capture, whenever a
        rank happens to capture (ragged batches give the ranks different shape signatures, so they capture on different
        steps: ADVICE r4) -- is compared with the agreed plan LOCALLY."""
        if self.verified_plan is not None:
            if mine != self.verified_plan:
                raise PlanMismatch(f"GradReducer ({where}): this step's collectives {mine} differ from the plan the ranks agreed on "
                                   f"{self.verified_plan}")
            return mine
        if self.world > 1:
            plans = [None] * self.world
            dist.all_gather_object(plans, mine, group=self.group)
            for r, other in enumerate(plans):
                if other != mine:
                    raise PlanMismatch(f"GradReducer ({where}): rank {r} plans other collectives than this rank ({other} vs {mine}): "
                                       "the staged steps of the ranks differ")
        self.verified_plan = mine
        return mine

    def assert_same_plan(self, ready_per_stage):
        """Called by GraphedTrainStep right after a capture with the buckets each captured stage completed: NO collective here
        (capture timing is per rank) -- the captured plan must equal the plan the ranks agreed on in the first staged step; a
        capture in front of that step (warmup = 0) is checked by that step's wait()."""
        mine = self.plan(ready_per_stage)
        if self.verified_plan is not None:
            self._check_plan(mine, "capture")
        return mine

    def _launch(self, b: int, after: Optional[Iterable["torch.cuda.Stream"]] = None):
        """hook mode: one bucket, behind every stream a gradient slice of it may have been written from"""
        if self.launched[b]:
            return
        self.launched[b] = True
        if self.cuda:
            self._join_producers([torch.cuda.current_stream(self.flat.grad.device), self.main_stream]
                                 + list(self.extra_streams) + list(after or []))
        self._all_reduce(self.buckets[b][0], self.buckets[b][1])

    def launch(self, bucket_ids: Iterable[int], after=None):
        """Staged mode: all-reduce these buckets now, behind everything enqueued on the stream(s) ``after`` -- the stream a
        captured stage was replayed on: the end of a captured graph has joined every stream forked inside it, so that ONE
        dependency covers all producers (waiting for the modality side streams as well, per bucket, put ~35 cross-stream waits
        in front of the optimizer kernel: +0.33 ms per step on one rank)."""
        if after is not None and not isinstance(after, (list, tuple)):
            after = [after]
        ids = [b for b in bucket_ids if not self.launched[b]]
        if not ids:
            return
        for b in ids:
            self.launched[b] = True
        if self.staged:
            self.launch_log.append(list(ids))
        behind = None
        if self.cuda and after and len(after) == 1:
            behind = after[0]        # issued "from" the replay stream itself: one cross-stream hop less in front of the optimizer
        elif self.cuda:
            self._join_producers(list(after) if after else [torch.cuda.current_stream(self.flat.grad.device), self.main_stream]
                                 + list(self.extra_streams))
        for lo, hi in self._ranges(ids):
            self._all_reduce(lo, hi, behind)

    def wait(self):
        """Launch whatever has not been launched (parameters without a gradient this step keep their
        zeroed slice), then make the compute stream wait for every bucket."""
        if self.staged and self._sync:
            self._check_plan(self.plan(self.launch_log), "step")      # before the last launch: a mismatch must not hang in it
        staged, self.staged = self.staged, False                       # (the remainder is not a stage of its own in the log)
        self.launch([b for b in range(len(self.buckets)) if not self.launched[b]])
        self.staged = staged
        for w in self.works:
            w.wait()
        if self.cuda:
            torch.cuda.current_stream(self.flat.grad.device).wait_stream(self.side)
        self.works = []
        self.last_issued = list(self.issued)     # the finished step's collectives, in issue order (tests, debugging)
        self._reset()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self.flat.ready_cb = None
        self.flat.zero_cb = None
