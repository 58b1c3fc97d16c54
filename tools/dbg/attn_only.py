"""Runs the attention forward/backward a few times at the config-2 shape (for rocprofv3 --pmc passes).
    python tools/dbg/attn_only.py [n] [online|bounded] [bwd]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medical_tri_modal_pilot_amd import ops
B, N = 64, 1005
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 768, device="cuda", generator=g).bfloat16()
res = torch.randn(B, N, 256, device="cuda", generator=g).bfloat16()
do = torch.randn(B, N, 256, device="cuda", generator=g).bfloat16()
kv = torch.full((B,), N, dtype=torch.int32, device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mode = sys.argv[2] if len(sys.argv) > 2 else "bounded"
kn = ops.key_norms(qkv) if mode == "bounded" else None
for _ in range(n):
    o, _, lse = ops.attn_fwd(qkv, kv, res=res, knorm=kn)
    if len(sys.argv) > 3:
        ops.attn_bwd(qkv, o, do, lse, kv)
torch.cuda.synchronize()
