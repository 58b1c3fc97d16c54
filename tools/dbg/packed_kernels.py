"""Stand-alone times of the vital-sign stream's kernels on a PACKED stream at config-2 size (B 64, N 1005) for three length
distributions: full, all samples half length, ragged U{3..1000} -- how much of the padded time do half the rows cost?
    python tools/dbg/packed_kernels.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from medical_tri_modal_pilot_amd import ops
DEV, BF = "cuda:0", torch.bfloat16
B, N, D = 64, 1005, 256
g = torch.Generator(device=DEV).manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g, device=DEV).to(BF)

def timeit(fn, rounds=7, inner=5):
    for _ in range(2): fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner * 1e3)
    return sorted(ts)[len(ts) // 2]

M = B * N
z, dy = R(M, D), R(M, D)
gm, bt = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
wqkv, w1, w2 = R(768, D) * 0.05, R(1024, D) * 0.05, R(D, 1024) * 0.05
bq, b1, b2 = torch.zeros(768, device=DEV), torch.zeros(1024, device=DEV), torch.zeros(D, device=DEV)
w2t, wqkvt, w1t = w2.t().contiguous(), wqkv.t().contiguous(), w1.t().contiguous()
gl = torch.Generator().manual_seed(1)
dists = {"full": torch.full((B,), N), "half": torch.full((B,), N // 2), "ragged": torch.randint(8, N + 1, (B,), generator=gl)}
for name, lens in dists.items():
    kv = lens.to(torch.int32).to(DEV)
    pk = [ops.row_starts(kv, N)]
    live = int(lens.sum())
    res = {}
    qkv, xn1, st1, kn = ops.ln_gemm_qkv_grouped([z], [gm], [bt], [wqkv], [bq], pk)
    res["ln_gemm_qkv"] = timeit(lambda: ops.ln_gemm_qkv_grouped([z], [gm], [bt], [wqkv], [bq], pk))
    q3 = [qkv[0].view(B, N, 768)]
    o, r1, lse = ops.attn_fwd_grouped(q3, [kv], [z.view(B, N, D)], kn, pk)
    res["attn_fwd"] = timeit(lambda: ops.attn_fwd_grouped(q3, [kv], [z.view(B, N, D)], kn, pk))
    h, xn2, st2, sg = ops.ln_gemm_signs_grouped([z], [gm], [bt], [w1], [b1], 1024, 0.1, [3], pk)
    res["ln_gemm_ffn1"] = timeit(lambda: ops.ln_gemm_signs_grouped([z], [gm], [bt], [w1], [b1], 1024, 0.1, [3], pk))
    res["gemm_nt_ffn2"] = timeit(lambda: ops.gemm_nt_grouped(h, [w2], [b2], [z], 0.1, [5], pk))
    res["dH_signs_drop"] = timeit(lambda: ops.gemm_nt_signs_drop_grouped([dy], [w2t], sg, 1 / 0.9, 0.1, [5], pk))
    res["gemm_tn_1024"] = timeit(lambda: ops.gemm_tn_grouped(h, xn2, [None], [[]], pk))
    res["gemm_lnbwd_1024"] = timeit(lambda: ops.gemm_lnbwd_grouped(h, [w1t], [z], st2, [gm], [dy], [None], [[]], pk))
    res["attn_bwd"] = timeit(lambda: ops.attn_bwd_grouped(q3, o, [dy.view(B, N, D)], lse, [kv], pk))
    print(f"{name:7s} live rows {live:6d} ({100.0 * live / M:5.1f} %)  " + "  ".join(f"{k} {v:6.1f}" for k, v in res.items()), flush=True)
