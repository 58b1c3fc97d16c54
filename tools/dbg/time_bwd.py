import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medical_tri_modal_pilot_amd import ops
B, N = 64, 1005
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 768, device="cuda", generator=g).bfloat16()
do = torch.randn(B, N, 256, device="cuda", generator=g).bfloat16()
kv = torch.full((B,), N, dtype=torch.int32, device="cuda")
o, _, lse = ops.attn_fwd(qkv, kv)
def t(fn, n=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[0], ts[len(ts) // 2]
print(os.environ.get("MTMP_LIB", "shipped").split("_")[-1], "bwd us (min, median):", t(lambda: ops.attn_bwd(qkv, o, do, lse, kv)), "fwd:", t(lambda: ops.attn_fwd(qkv, kv)))
