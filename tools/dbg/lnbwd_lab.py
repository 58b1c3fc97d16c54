"""gemm_lnbwd (K = 1024 and 768, M = 64,320) on a tools/lab_lib.py library variant: python tools/dbg/lnbwd_lab.py NAME [NAME ...]
(separate processes per variant are the caller's business; one variant per call here)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from medical_tri_modal_pilot_amd import _lib
name = sys.argv[1]
if name != "product":
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "dbg", "_lab", name, "libmtmp_hip.so")
from medical_tri_modal_pilot_amd import ops
DEV, BF = "cuda:0", torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g, device=DEV).to(BF)
M = 64 * 1005
z, dres = R(M, 256), R(M, 256)
gm = torch.ones(256, device=DEV)
st = torch.stack([z.float().mean(-1), 1 / (z.float().std(-1) + 1e-6)], 1).contiguous()
def timeit(fn, rounds=9, inner=5):
    for _ in range(3): fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner * 1e3)
    return sorted(ts)[len(ts) // 2]
out = []
for K in (1024, 768):
    dy, wt = R(M, K), R(256, K) * 0.05
    out.append(f"K={K}: {timeit(lambda: ops.gemm_lnbwd_grouped([dy], [wt], [z], [st], [gm], [dres], [None], [[]])):.1f} us")
print(name, "  ".join(out))
