"""Per-kernel micro-benchmarks at the config-2 shapes (B=64, N_v=1005): TF/s or GB/s of every
libmtmp_hip.so kernel next to the library GEMM (torch.matmul -> hipBLASLt) on the same random
data, measured with HIP events on the launch stream (interleaved rounds, median).

    python tools/bench_kernels.py [--rounds 7] [--only gemm_nt,attn]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from medical_tri_modal_pilot_amd import ops  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def timeit(fn, rounds, inner=5):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    only = [s for s in a.only.split(",") if s]
    g = torch.Generator(device=DEV).manual_seed(0)
    R = lambda *s: torch.randn(*s, generator=g, device=DEV, dtype=torch.float32).to(BF)
    M = 64 * 1005
    out = {}

    def rec(name, ms, flops=None, bytes_=None):
        d = {"ms": round(ms, 4)}
        if flops:
            d["TFLOPs"] = round(flops / ms / 1e9, 1)
        if bytes_:
            d["GBs"] = round(bytes_ / ms / 1e6, 1)
        out[name] = d
        print(name, d, flush=True)

    def want(k):
        return not only or any(k.startswith(o) or o.startswith(k) for o in only)

    # ---- NT GEMMs
    for name, (m, n, k) in {"gemm_nt.ffn2[M,256,1024]": (M, 256, 1024), "gemm_nt.dH[M,1024,256]": (M, 1024, 256),
                            "gemm_nt.dxn2[M,256,1024]": (M, 256, 1024), "gemm_nt.swin_qkv1[200704,288,96]": (200704, 288, 96),
                            "gemm_nt.swin_fc1_1[200704,384,96]": (200704, 384, 96),
                            "gemm_nt.swin_fc2_1[200704,96,384]": (200704, 96, 384),
                            "gemm_nt.swin_fc1_3[12544,1536,384]": (12544, 1536, 384),
                            "gemm_nt.swin_fc2_3[12544,384,1536]": (12544, 384, 1536)}.items():
        if not want("gemm_nt"):
            break
        x, w = R(m, k), R(n, k) * 0.05
        b = torch.zeros(n, device=DEV)
        r = R(m, n)
        rec(name, timeit(lambda: ops.gemm_nt(x, w, b, res2d=r), a.rounds), 2.0 * m * n * k)
        rec(name + ".blas", timeit(lambda: torch.addmm(r, x, w.t()), a.rounds), 2.0 * m * n * k)
    if want("gemm_nt"):             # dH = dY W2 gated by the saved hidden activation (FFN backward)
        dy, w2t, h = R(M, 256), R(1024, 256) * 0.05, R(M, 1024)
        rec("gemm_nt.dH_gated[M,1024,256]", timeit(lambda: ops.gemm_nt(dy, w2t, gate=h, gate_scale=1.0 / 0.9), a.rounds), 2.0 * M * 1024 * 256)
    if want("gemm_nt"):             # the same product gated by the forward's sign bits
        x0, w1 = R(M, 256), R(1024, 256) * 0.05
        _, _, _, sg = ops.ln_gemm(x0, torch.ones(256, device=DEV), torch.zeros(256, device=DEV), w1, None, 1024, relu=True, drop_p=0.1,
                                  seed=3, want_signs=True)
        rec("gemm_nt.dH_signs[M,1024,256]", timeit(lambda: ops.gemm_nt_signs(dy, w2t, sg, 1.0 / 0.9), a.rounds), 2.0 * M * 1024 * 256,
            2.0 * M * (256 + 1024) + M * 128)
    # ---- LN-fused GEMMs
    if want("ln_gemm"):
        x = R(M, 256)
        gm, bt = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
        for n, relu in ((768, False), (1024, True)):
            w = R(n, 256) * 0.05
            b = torch.zeros(n, device=DEV)
            rec(f"ln_gemm[M,{n},256]", timeit(lambda: ops.ln_gemm(x, gm, bt, w, b, n, relu=relu), a.rounds), 2.0 * M * n * 256,
                2.0 * M * (256 + n + 256))
            if relu:
                rec(f"ln_gemm[M,{n},256].drop.signs", timeit(lambda: ops.ln_gemm(x, gm, bt, w, b, n, relu=True, drop_p=0.1, seed=7,
                                                                                   want_signs=True), a.rounds),
                    2.0 * M * n * 256, 2.0 * M * (256 + n + 256))
                rec(f"ln_gemm[M,{n},256].drop", timeit(lambda: ops.ln_gemm(x, gm, bt, w, b, n, relu=True, drop_p=0.1, seed=7), a.rounds),
                    2.0 * M * n * 256, 2.0 * M * (256 + n + 256))
            rec(f"ln_gemm[M,{n},256].blas_nolN", timeit(lambda: torch.addmm(b.to(BF), x, w.t()), a.rounds), 2.0 * M * n * 256)
    # ---- TN (weight gradient) GEMMs
    if want("gemm_tn"):
        for n, k in ((768, 256), (1024, 256), (256, 1024)):
            dy, x = R(M, n), R(M, k)
            rec(f"gemm_tn[{n},{k},M]", timeit(lambda: ops.gemm_tn(dy, x), a.rounds), 2.0 * M * n * k)
            rec(f"gemm_tn[{n},{k},M].blas", timeit(lambda: dy.t() @ x, a.rounds), 2.0 * M * n * k)
    # ---- attention
    if want("attn"):
        B, N = 64, 1005
        qkv = R(B, N, 768)
        res, do = R(B, N, 256), R(B, N, 256)
        kv = torch.full((B,), N, dtype=torch.int32, device=DEV)
        f = 4.0 * B * 4 * N * N * 64
        rec("attn_fwd[64,1005].online", timeit(lambda: ops.attn_fwd(qkv, kv, res=res), a.rounds), f)
        kn = ops.key_norms(qkv)
        rec("key_norms[64,1005]", timeit(lambda: ops.key_norms(qkv), a.rounds), bytes_=2.0 * B * N * 256)
        rec("attn_fwd[64,1005].bounded", timeit(lambda: ops.attn_fwd(qkv, kv, res=res, knorm=kn), a.rounds), f)
        qs = qkv * 0.35                # scores of the size a LayerNorm-fed projection produces
        kn2 = ops.key_norms(qs)
        rec("attn_fwd[64,1005].bounded.smallscores", timeit(lambda: ops.attn_fwd(qs, kv, res=res, knorm=kn2), a.rounds), f)
        o, _, lse = ops.attn_fwd(qkv, kv, res=res)
        rec("attn_bwd[64,1005]", timeit(lambda: ops.attn_bwd(qkv, o, do, lse, kv), a.rounds), 2.5 * f)
    # ---- streaming kernels
    if want("stream"):
        z, dy = R(M, 256), R(M, 256)
        st = torch.stack([z.float().mean(-1), 1 / (z.float().std(-1) + 1e-6)], 1).contiguous()
        gm = torch.ones(256, device=DEV)
        rec("ln_bwd[M,256]", timeit(lambda: ops.ln_bwd(z, st, gm, dy, dy), a.rounds), bytes_=4.0 * M * 256 * 2)
        x = R(64, 56, 56, 96)
        w, b = torch.ones(96, device=DEV), torch.zeros(96, device=DEV)
        rec("ln_rows[200704,96]", timeit(lambda: ops.layernorm_rows(x, w, b), a.rounds), bytes_=2.0 * x.numel() * 2)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_kernels.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
