"""Stand-alone time of the attention half of a Swin block: mtmp_swin_attn_block against the chain it replaces (idle device).
    python tools/dbg/swin_blk_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from medical_tri_modal_pilot_amd import ops
from medical_tri_modal_pilot_amd.builder.models.src import swin_transformer as ST

dev = torch.device("cuda", 0)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for n, H, C, heads, shift in [(64, 56, 96, 3, 0), (64, 56, 96, 3, 3), (64, 28, 192, 6, 0), (64, 28, 192, 6, 3), (38, 56, 96, 3, 3)]:
    blk = ST.SwinTransformerBlock(C, heads, [7, 7], [shift, shift], 0.1).to(dev).eval()
    x = torch.randn(n, H, H, C, device=dev).to(torch.bfloat16)
    sc = torch.ones(n, device=dev)
    at = blk.attn
    dt = x.dtype

    def fused():
        return ops.swin_attn_block(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, ST._w(at.qkv.weight, dt), at.qkv.bias,
                                   at.additive_table(shift, dt, dev, acc_order=True), heads, shift, ST._w(at.proj.weight, dt),
                                   at.proj.bias, sc)

    def chain():
        if C in ops.SWIN_LN_LINEAR_WIDTHS:
            a = at(x, norm=blk.norm1)
        else:
            a = at(ops.layernorm_rows(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps))
        return ops.gemm_nt(a.view(-1, C), ST._w(at.proj.weight, dt), at.proj.bias, res2d=x.view(-1, C), row_scale=sc, rows_per_scale=H * H)

    with torch.no_grad():
        print(f"n={n} H={H} C={C} shift={shift}: fused {timed(fused):7.1f} us   chain {timed(chain):7.1f} us   "
              f"(tokens r+w {2 * x.numel() * 2 / 1e6:.1f} MB)")
