"""In-kernel s_memtime stamps of swin_mlp_kernel (lab build with -DMTMP_LAB_CLOCK: tools/lab_lib.py run clock tools/dbg/swin_mlp_clock.py):
where a workgroup's lifetime goes -- prologue, the six hidden panels (compute | commit + fetch | barrier), epilogue."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from medical_tri_modal_pilot_amd import ops, _lib

dev = torch.device("cuda", 0)
n, hw, C = 64, 3136, 96
M = n * hw
torch.manual_seed(0)
x = torch.randn(M, C, device=dev).to(torch.bfloat16)
w1 = (torch.randn(4 * C, C, device=dev) * 0.05).to(torch.bfloat16)
w2 = (torch.randn(C, 4 * C, device=dev) * 0.05).to(torch.bfloat16)
b1, b2 = torch.randn(4 * C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
sc = torch.ones(n, device=dev)
for _ in range(5):
    ops.swin_mlp(x, g, b, 1e-5, w1, b1, w2, b2, sc, hw)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
nst = 256 * 4 * 32
buf = (ctypes.c_longlong * nst)()
assert lib.mtmp_dbg_read_swin_stamps(buf, nst) == 0
s = np.frombuffer(buf, dtype=np.int64).reshape(256, 4, 32)[:196]          # workgroups 0, 8, ..., 1560
t0 = s[:, :, 0].min()
life = s[:, :, 23] - s[:, :, 0]
print("launch span cycles:", int(s[:, :, 23].max() - t0), " wave lifetime: median", int(np.median(life)), "min", int(life.min()), "max", int(life.max()))
start = (s[:, 0, 0] - t0)
print("start times of the sampled workgroups (cycles, sorted, every 16th):", [int(v) for v in np.sort(start)[::16]])
med = lambda a: int(np.median(a))
print("prologue: fetch+x+LN (0->1)", med(s[:, :, 1] - s[:, :, 0]), " commit+fetch+barrier (1->2)", med(s[:, :, 2] - s[:, :, 1]))
prev = 2
for j in range(6):
    a, bq, c = 3 + 3 * j, 4 + 3 * j, 5 + 3 * j
    print(f"panel {j}: compute {med(s[:, :, a] - s[:, :, prev])}  commit+fetch {med(s[:, :, bq] - s[:, :, a])}  barrier {med(s[:, :, c] - s[:, :, bq])}")
    prev = c
print("epilogue: loads landed (21->22)", med(s[:, :, 22] - s[:, :, 21]), " stage + stores (22->23)", med(s[:, :, 23] - s[:, :, 22]))
first = s[start < np.median(start)]
late = s[start >= np.median(start)]
for nm, q in (("early workgroups", first), ("late workgroups", late)):
    print(nm, "lifetime", med(q[:, :, 23] - q[:, :, 0]), " loop", med(q[:, :, 21] - q[:, :, 2]))
