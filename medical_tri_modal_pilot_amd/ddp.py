"""Data-parallel gradient exchange for the tri-modal step (new: the reference has no multi-GPU
path).  One process per GPU; patient batches are sharded across ranks; the only
collective is the gradient all-reduce -- 12,121,601 fp32 values (48.5 MB) at 6 layers.

Design for xGMI (point-to-point links, ring collectives are per-link bound): few large
contiguous buckets cut from the flat gradient buffer of optim.FlatParams in REVERSE parameter
order (head -> last layer ... first layer -> embeddings = the order backward produces them).
A bucket's all-reduce is launched as soon as every parameter in it has accumulated its gradient
(post-accumulate-grad hooks), on a side HIP stream that waits on an event recorded on the
compute stream, so RCCL traffic overlaps the rest of backward; ``wait()`` (called by
FusedAdamW.step) joins the side stream.  Sums are left un-averaged: the 1/world factor is folded
into the AdamW kernel's ``grad_scale``.  BatchNorm statistics stay per-rank (standard DDP).
"""
from typing import List, Optional

import torch
import torch.distributed as dist

from .optim import FlatParams


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None):
    """Initial replica sync: parameters and buffers (BatchNorm running stats included) from rank `src`."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)


class GradReducer:
    def __init__(self, flat: FlatParams, bucket_bytes: int = 12 << 20, group=None, overlap: bool = True):
        self.flat, self.group, self.overlap = flat, group, overlap
        self.world = dist.get_world_size(group)
        self.cuda = flat.grad.is_cuda
        self.side = torch.cuda.Stream(device=flat.grad.device) if self.cuda else None
        # other streams gradients are produced on (the encoder's modality side streams): a bucket's all-reduce waits
        # for everything enqueued on them too -- the slice that completed the bucket says nothing about its
        # neighbours written from another stream
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
        self.pending = [b[2] for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.works = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(flat.params)]
        flat.ready_cb = self._ready          # gradients written straight into the flat buffer (ops.GradSink)

    def _ready(self, i: int):
        b = self.bucket_of[i]
        self.pending[b] -= 1
        if self.pending[b] == 0 and self.overlap:
            self._launch(b)

    def _make_hook(self, i: int):
        def hook(_param):
            self._ready(i)
        return hook

    def _launch(self, b: int):
        if self.launched[b]:
            return
        self.launched[b] = True
        lo, hi, _ = self.buckets[b]
        buf = self.flat.grad[lo:hi]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.side.wait_event(ev)
            for s in self.extra_streams:
                self.side.wait_stream(s)
            with torch.cuda.stream(self.side):
                self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        """Launch whatever has not been launched (parameters without a gradient this step keep their
        zeroed slice), then make the compute stream wait for every bucket."""
        for b in range(len(self.buckets)):
            self._launch(b)
        for w in self.works:
            w.wait()
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.side)
        self.works = []
        self.pending = [b[2] for b in self.buckets]
        self.launched = [False] * len(self.buckets)

    def remove(self):
        for h in self._hooks:
            h.remove()
