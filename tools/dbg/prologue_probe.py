"""Host time of the trainer's prologue (entry -> graph replay launched) and of the replay launch itself, steady state."""
import os, sys, time, statistics as st
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from medical_tri_modal_pilot_amd import graph as G
from medical_tri_modal_pilot_amd.builder.trainer import trainer as T
rec = {"enter": [], "run_in": [], "run_out": [], "ret": []}
_run, _mt = G.GraphedTrainStep.run, T.missing_trainer
def run(self, *a, **k):
    rec["run_in"].append(time.perf_counter()); r = _run(self, *a, **k); rec["run_out"].append(time.perf_counter()); return r
def mt(*a, **k):
    rec["enter"].append(time.perf_counter()); r = _mt(*a, **k); rec["ret"].append(time.perf_counter()); return r
G.GraphedTrainStep.run = run
T.missing_trainer = mt
import medical_tri_modal_pilot_amd.builder.trainer as TP
TP.missing_trainer = mt
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "30", "--warmup", "8", "--probe-steps", "0"]
import runpy
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
n = min(len(rec["enter"]), len(rec["run_in"]))
pro = [1e6 * (rec["run_in"][-i] - rec["enter"][-i]) for i in range(1, 21)]
rep = [1e6 * (rec["run_out"][-i] - rec["run_in"][-i]) for i in range(1, 21)]
tail = [1e6 * (rec["ret"][-i] - rec["run_out"][-i]) for i in range(1, 21)]
print("prologue us median", st.median(pro), "replay call us", st.median(rep), "after replay -> return us (incl. loss.item wait)", st.median(tail), file=sys.stderr)
