"""cProfile of the host side of the replayed (graph-mode) training step; top functions by cumulative and own time."""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "200", "--warmup", "10", "--probe-launches", "0"]
import runpy, torch
from medical_tri_modal_pilot_amd import graph as G
pr = cProfile.Profile()
_run = G.GraphedTrainStep.run
state = {"n": 0}
def run(self, *a, **k):
    state["n"] += 1
    if state["n"] == 12:
        pr.enable()
    return _run(self, *a, **k)
G.GraphedTrainStep.run = run
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(32)
    print(s.getvalue()[:7000], file=sys.stderr)
