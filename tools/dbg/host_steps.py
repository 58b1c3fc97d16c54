"""Host wall time of the pieces of one replayed training step (bench.py, graph mode): per-line timers around the trainer's
prologue, GraphedTrainStep.run (static-input copy, replay), optimizer.step and loss.item(), medians over the timed steps."""
import atexit, os, statistics as st, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from medical_tri_modal_pilot_amd import graph as G, optim as O
import medical_tri_modal_pilot_amd.builder.trainer.trainer as TR

rec = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        rec.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    return w

TR.missing_trainer = timed("missing_trainer (whole step)", TR.missing_trainer)
TR.missing_to_num = timed("  missing_to_num", TR.missing_to_num)
G.GraphedTrainStep.run = timed("  GraphedTrainStep.run", G.GraphedTrainStep.run)
torch.cuda.CUDAGraph.replay = timed("    CUDAGraph.replay", torch.cuda.CUDAGraph.replay)
torch._foreach_copy_ = timed("    _foreach_copy_", torch._foreach_copy_)
O.FusedAdamW.step = timed("  FusedAdamW.step", O.FusedAdamW.step)
torch.Tensor.item = timed("  Tensor.item", torch.Tensor.item)
torch.Tensor.half = timed("  Tensor.half", torch.Tensor.half)

@atexit.register
def report():
    for k, v in rec.items():
        if len(v) > 12:
            n = len(v) // max(1, len(rec["missing_trainer (whole step)"]))
            print("%-34s median %8.1f us  x%d per step" % (k, 1e6 * st.median(v[10 * n:]), n), file=sys.stderr)

sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "30", "--warmup", "10", "--probe-launches", "0"]
import runpy
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
