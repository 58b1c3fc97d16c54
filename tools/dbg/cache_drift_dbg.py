"""Losses of the graph-cache test's two runs (replayed with length buckets vs eager with exact trims), step by step."""
import os, sys
ROOT = os.environ.get("DBG_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import test_gpu_parity as T
from medical_tri_modal_pilot_amd import ops
Tn, B, L, CAP = 2000, 4, 12, 5
maxima = [2000, 1700, 1500, 1200, 1000, 880, 760, 630, 500, 380, 250, 100]
order = [m for m in maxima for _ in range(2)] + maxima
lens = [[m, max(3, m // 2), max(3, m // 3), 3] for m in order]
lg, _, gs = T._loop(1, 0.0, "bf16", 10, lens[:10], L=L, B=B, T=Tn, hip_graph_max=CAP)
le, _, _ = T._loop(0, 0.0, "bf16", 10, lens[:10], L=L, B=B, T=Tn)
print("root", ROOT)
for i, (a, b) in enumerate(zip(lg, le)):
    print(i, order[i], f"graph {a:.6f} eager {b:.6f} diff {abs(a - b):.2e}")
