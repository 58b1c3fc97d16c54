"""TIMING PROBE, results are wrong on purpose: what would the step cost if the fusion layers' weight-gradient products
(gemm_tn_grouped + their slab reductions) did not run inside the backward chain but at the HEAD of the next step, on the main
stream beside the frozen image encoder?  The backward's launches are skipped, the same launches run at the head of the step on
stand-in operands of the same shapes.

    python tools/dbg/late_tn_probe.py [0|1] [bench.py flags]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from medical_tri_modal_pilot_amd import ops, optim as O  # noqa: E402

ON = len(sys.argv) > 1 and sys.argv[1] == "1"
if len(sys.argv) > 1 and sys.argv[1] in ("0", "1"):
    del sys.argv[1]

slots, state = [], {"recorded": False, "calls": 0}
_tn, _zero = ops.gemm_tn_grouped, O.FusedAdamW.zero_grad


def tn(dys, xs, outs, defers, packs=None):
    if not state["recorded"]:
        slots.append(([torch.empty_like(d) for d in dys], [torch.empty_like(x) for x in xs], list(outs),
                      None if packs is None else [None if p is None else p.clone() for p in packs]))
        return _tn(dys, xs, outs, defers, packs)
    state["calls"] += 1
    return [o if o is not None else (None, None) for o in outs]          # skipped: the head of the step ran its stand-in


def zero_grad(self, set_to_none=True):
    if slots and not state["recorded"]:
        state["recorded"] = True
        print(f"late_tn_probe: {len(slots)} weight-gradient launches move to the head of the step", file=sys.stderr)
    if state["recorded"]:
        reds = []
        for dys, xs, outs, packs in slots:
            d = [[] for _ in dys]
            _tn(dys, xs, outs, d, packs)
            reds += [e for r in d for e in r]
        ops.reduce_batch(reds)
    return _zero(self, set_to_none)


if ON:
    ops.gemm_tn_grouped, O.FusedAdamW.zero_grad = tn, zero_grad
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "30", "--warmup", "8", "--probe-launches", "0", "--instep-steps", "0"] + sys.argv[1:]
import runpy  # noqa: E402
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
