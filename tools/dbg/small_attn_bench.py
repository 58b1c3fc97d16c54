"""Stand-alone time of the attention launches of the image + text group (B 64, N 54 / 133, H 4): forward and backward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from medical_tri_modal_pilot_amd import ops
DEV, BF = "cuda:0", torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(0)
B, Ns = 64, [54, 133]
qkv = [torch.randn(B, N, 768, generator=g, device=DEV).to(BF) for N in Ns]
res = [torch.randn(B, N, 256, generator=g, device=DEV).to(BF) for N in Ns]
do = [torch.randn(B, N, 256, generator=g, device=DEV).to(BF) for N in Ns]
kv = [None, torch.randint(5, 134, (B,), generator=g, device=DEV).to(torch.int32)]


def timed(fn, reps=20, replays=10):
    """20 launches captured in a hipGraph, replayed: device time per launch (eagerly, the host cannot issue these fast enough)"""
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=torch.cuda.Stream(device=DEV)):
        for _ in range(reps):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        gr.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * replays) * 1e3


kn = [ops.key_norms(q) for q in qkv]
o, o_res, lse = ops.attn_fwd_grouped(qkv, kv, res, kn)
print(f"fwd (image + text group): {timed(lambda: ops.attn_fwd_grouped(qkv, kv, res, kn)):.1f} us")
print(f"bwd (image + text group): {timed(lambda: ops.attn_bwd_grouped(qkv, o, do, lse, kv)):.1f} us")
