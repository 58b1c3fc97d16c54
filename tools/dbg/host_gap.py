"""Host time between two replayed steps: loss.item() return -> GraphedTrainStep.run entry -> CUDAGraph.replay entry/exit
(the GPU idles from the previous step's AdamW until the replay's first kernel)."""
import atexit, os, statistics as st, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from medical_tri_modal_pilot_amd import graph as G

T = {"item": 0.0}
rec = {"item->run": [], "run->replay": [], "replay": []}
_item, _run, _replay = torch.Tensor.item, G.GraphedTrainStep.run, torch.cuda.CUDAGraph.replay


def item(self):
    r = _item(self)
    T["item"] = time.perf_counter()
    return r


def run(self, *a, **k):
    T["run"] = time.perf_counter()
    if T["item"]:
        rec["item->run"].append(T["run"] - T["item"])
    return _run(self, *a, **k)


def replay(self):
    t0 = time.perf_counter()
    if "run" in T:
        rec["run->replay"].append(t0 - T["run"])
    r = _replay(self)
    rec["replay"].append(time.perf_counter() - t0)
    return r


torch.Tensor.item, G.GraphedTrainStep.run, torch.cuda.CUDAGraph.replay = item, run, replay


@atexit.register
def report():
    for k, v in rec.items():
        if len(v) > 10:
            print("%-12s median %.1f us (n=%d)" % (k, 1e6 * st.median(v[10:]), len(v) - 10), file=sys.stderr)


sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "30", "--warmup", "10", "--probe-launches", "0"]
import runpy
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
