"""Measurement instrumentation of the hot path -- NOT part of the training computation (kept apart from ops.py).

Stream-ordered time stamps (C entry ``mtmp_timestamp``): off unless ``marks_enable()`` was called.  ``mark(name)`` launches a
one-lane kernel on the current stream that stores the 100 MHz wall clock into the slot of ``name``; inside a captured step the
launches become graph nodes, so every replay refreshes the slots.  Users: ``bench.py`` (the in-step duration of the roofline
kernels -- HIP events cannot be timed inside a replayed hipGraph; ``kernel_marks`` brackets the grouped attention / weight-gradient
launches) and ``tools/dbg/timeline.py`` (un-profiled per-stream timeline).  With marks disabled (every training run) ``mark()``
returns at once and the step contains no extra launch.
"""
import ctypes
from typing import Optional

import torch

from ._lib import call

_marks: Optional[torch.Tensor] = None
_mark_slots: dict = {}
_mark_only: Optional[tuple] = None
_mark_seq: dict = {}


def marks_enable(device, n: int = 1024, only: Optional[tuple] = None):
    """only: name prefixes to record (None = every mark)."""
    global _marks, _mark_only
    _marks = torch.zeros(n, dtype=torch.int64, device=device)
    _mark_only = only
    _mark_slots.clear()
    _mark_seq.clear()


def marks_disable():
    global _marks
    _marks = None


def marks_new_step():
    """restart the per-step launch counters of kernel_marks (call before every eager step / before a capture)"""
    _mark_seq.clear()


def mark(name: str):
    if _marks is None or (_mark_only is not None and not name.startswith(_mark_only)):
        return
    i = _mark_slots.setdefault(name, len(_mark_slots))
    call("mtmp_timestamp", ctypes.c_void_p(_marks.data_ptr() + 8 * i), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))


class kernel_marks:
    """with kernel_marks("attn_fwd", N): <launch>  -- stamps "k.attn_fwd.N<N>.<i>.s / .e" around the i-th such launch of a step"""

    def __init__(self, kind: str, n_rows: int):
        self.name = None
        if _marks is not None:
            key = f"k.{kind}.N{n_rows}"
            i = _mark_seq.get(key, 0)
            _mark_seq[key] = i + 1
            self.name = f"{key}.{i}"

    def __enter__(self):
        if self.name is not None:
            mark(self.name + ".s")

    def __exit__(self, *exc):
        if self.name is not None:
            mark(self.name + ".e")
        return False


def marks_read() -> dict:
    """{name: microseconds} of the last pass over every mark (100 MHz counter)."""
    v = _marks.cpu().tolist()
    return {k: v[i] / 100.0 for k, i in _mark_slots.items()}
