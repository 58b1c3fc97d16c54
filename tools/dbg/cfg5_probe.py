"""BASELINE configs[4]-like stress shape on one GPU: batch 128, TIE-len 2000 (N_v = 2005), 12 layers (single image):
does the path run at that size, and how long is a step?  (bench.py with its three shape constants replaced.)"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
src = open(os.path.join(ROOT, "bench.py")).read().replace("B_PER_GPU, TIE_LEN, LAYERS = 64, 1000, 6", "B_PER_GPU, TIE_LEN, LAYERS = 128, 2000, 12")
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "6", "--warmup", "3", "--probe-steps", "2"]
code = compile(src, os.path.join(ROOT, "bench.py"), "exec")
g = {"__name__": "__main__", "__file__": os.path.join(ROOT, "bench.py")}
exec(code, g)
