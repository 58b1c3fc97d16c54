"""TIMING PROBE, results are wrong on purpose: the step without the attention launches of the image + text group (their outputs
stay uninitialised) -- an upper bound on what those launches cost the vital-sign stream's kernels they run beside.
    python tools/dbg/skip_small_attn_probe.py [0|1|2] [bench flags]      (1: skip forward + backward, 2: skip the group's row kernels too)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from medical_tri_modal_pilot_amd import ops
mode = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] in ("0", "1", "2") else 0
if len(sys.argv) > 1 and sys.argv[1] in ("0", "1", "2"):
    del sys.argv[1]
_f, _b = ops.attn_fwd_grouped, ops.attn_bwd_grouped


def fwd(qkvs, kv_lens, ress, knorms, packs=None):
    if qkvs[0].shape[1] > 200:
        return _f(qkvs, kv_lens, ress, knorms, packs)
    B, dt, dev = qkvs[0].shape[0], qkvs[0].dtype, qkvs[0].device
    Ns = [q.shape[1] for q in qkvs]
    return ([torch.zeros(B, N, 256, dtype=dt, device=dev) for N in Ns], [r.clone() for r in ress],
            [torch.zeros(B, 4, N, dtype=torch.float32, device=dev) for N in Ns])


def bwd(qkvs, os_, d_os, lses, kv_lens, packs=None):
    if qkvs[0].shape[1] > 200:
        return _b(qkvs, os_, d_os, lses, kv_lens, packs)
    return [torch.zeros_like(q) for q in qkvs]


if mode >= 1:
    ops.attn_fwd_grouped, ops.attn_bwd_grouped = fwd, bwd
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "30", "--warmup", "8", "--probe-launches", "0"] + sys.argv[1:]
import runpy
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
