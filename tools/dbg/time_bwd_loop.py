"""Attention backward (dQ + dK/dV launches) at the config-2 shape: 30 back-to-back calls between one pair of events, 5 repeats.
MTMP_LIB selects an ablation build (diagnostics only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medical_tri_modal_pilot_amd import ops
B, N = 64, int(os.environ.get("N", 1005))
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 768, device="cuda", generator=g).bfloat16()
do = torch.randn(B, N, 256, device="cuda", generator=g).bfloat16()
kv = torch.full((B,), N, dtype=torch.int32, device="cuda")
o, _, lse = ops.attn_fwd(qkv, kv, knorm=ops.key_norms(qkv))


def t(fn, n=30, rep=5):
    for _ in range(5):
        fn()
    out = []
    for _ in range(rep):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / n)
    out.sort()
    return round(out[0], 1), round(out[len(out) // 2], 1)


kn = ops.key_norms(qkv)
print(os.path.basename(os.environ.get("MTMP_LIB", "shipped")), "bwd us (min, median):", t(lambda: ops.attn_bwd(qkv, o, do, lse, kv)),
      "fwd:", t(lambda: ops.attn_fwd(qkv, kv, knorm=kn)))
