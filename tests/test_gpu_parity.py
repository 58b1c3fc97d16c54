"""-m gpu: parity of the HIP path (through the C ABI) against the CPU oracle and the golden
vectors generated from the real reference.

Tolerances (written here, per north_star): fp32 "parity" build -- 1e-4 relative (plus a small
absolute floor for values near zero); integer artefacts bit-exact; bf16 "perf" build -- measured,
reported tolerance (bf16 has 8 significant bits: 3e-2 relative-to-scale on activations).
Every comparison also appends its error to gpurun_out/parity_report.json.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

import filler
from oracle import tri_mbt_oracle as O
from tests.state_shapes import reference_state_shapes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}
DEV = "cuda:0"


def _rel(a, b):
    """max |a-b| / (max|b| + tiny): error relative to the tensor's scale (a max-norm relative error, NOT element-wise:
    small entries of a tensor are held to the scale of its largest one; the full-step test adds an element-wise check)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check(name, got, ref, tol):
    e = _rel(got, ref)
    REPORT[name] = {"rel_err": e, "tol": tol}
    assert math.isfinite(e) and e <= tol, f"{name}: rel err {e:.3e} > {tol:.1e}"


IN_CHILD = os.environ.get("MTMP_TEST_CHILD") == "1"      # this session was started BY a test of the parent session (see _in_child_process)


@pytest.fixture(scope="module", autouse=True)
def _dump_report():
    yield
    if IN_CHILD:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def _in_child_process(test_name, timeout=600):
    """Run ONE test of this file in a pytest session of its own and assert that it passed.  For tests that create and DESTROY HIP
    streams they do not own -- an RCCL process group: round 4, twice in eight runs the suite stalled inside, or segfaulted in the
    graph replay right behind, the single-rank RCCL test (hip::Graph::UpdateStreams) -- so that nothing of it survives in the
    process the other ~220 tests share.  Returns True in the parent (done), False in the child (run the body)."""
    if IN_CHILD:
        return False
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", f"{os.path.abspath(__file__)}::{test_name}", "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=dict(os.environ, MTMP_TEST_CHILD="1"), capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    return True


@pytest.fixture(scope="module")
def ops():
    from medical_tri_modal_pilot_amd import ops as _ops
    return _ops


def G(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)


TOL = {torch.float32: 1e-4, torch.bfloat16: 3e-2}
DT = [torch.float32, torch.bfloat16]


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize("bounded", [False, True])
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("N,lens", [(54, None), (54, [54, 4, 30, 7]), (133, [133, 4, 5, 90]),
                                    (261, [261, 200, 4, 64]), (40, [0, 17, 40, 1]), (1005, [1005, 700]),
                                    (2005, [2005, 1290])])        # configs[4]: TIE-len 2000
def test_attention_fwd_bwd(ops, dt, N, lens, bounded):
    """bounded: the forward gets the key-norm table, so every wave takes the maximum-free body (|score| <= ~12 here)."""
    g = torch.Generator().manual_seed(7 + N)
    B = 4 if lens is None else len(lens)
    qkv = torch.randn(B, N, 768, generator=g)
    qkv = qkv.to(dt).float()                         # same rounded inputs on both sides
    res = torch.randn(B, N, 256, generator=g).to(dt).float()
    w = torch.randn(B, N, 256, generator=g).to(dt).float()
    kv = None if lens is None else torch.tensor(lens)
    q_ref = qkv.clone().requires_grad_()
    o_ref = O.attention_core(q_ref, kv)
    (o_ref * w).sum().backward()
    qd = qkv.to(DEV, dt)
    kvd = None if kv is None else kv.to(DEV, torch.int32)
    o, o_res, lse = ops.attn_fwd(qd, kvd, res=res.to(DEV, dt), knorm=ops.key_norms(qd) if bounded else None)
    tag = f"attn[{str(dt)[6:]},N={N},{'mask' if lens else 'nomask'}{'' if not lens else lens[1]}{',bounded' if bounded else ''}]"
    check(tag + ".o", o.float(), o_ref, TOL[dt])
    check(tag + ".o_res", o_res.float(), o_ref.detach().to(dt).float() + res, TOL[dt])
    dqkv = ops.attn_bwd(qd, o, w.to(DEV, dt), lse, kvd)
    for i, nm in enumerate("qkv"):
        check(f"{tag}.d{nm}", dqkv[..., 256 * i:256 * (i + 1)].float(), q_ref.grad[..., 256 * i:256 * (i + 1)],
              TOL[dt] if dt == torch.float32 else 4e-2)


@pytest.mark.parametrize("dt", DT)
def test_attention_grouped_equals_single_launches(ops, dt):
    """Three streams of one fusion layer (config-2 lengths, one of them without a mask) in one launch: bit-identical to three."""
    g = torch.Generator().manual_seed(31)
    B = 3
    Ns, lens = [1005, 54, 133], [[1005, 400, 9], None, [133, 4, 60]]
    qkv = [torch.randn(B, N, 768, generator=g).to(DEV, dt) for N in Ns]
    res = [torch.randn(B, N, 256, generator=g).to(DEV, dt) for N in Ns]
    do = [torch.randn(B, N, 256, generator=g).to(DEV, dt) for N in Ns]
    kv = [None if l is None else torch.tensor(l, dtype=torch.int32, device=DEV) for l in lens]
    kn = [ops.key_norms(q) for q in qkv]
    kn[2] = None                                         # a stream without the table takes the online body
    single = [ops.attn_fwd(q, k, res=r, knorm=t) for q, k, r, t in zip(qkv, kv, res, kn)]
    o, o_res, lse = ops.attn_fwd_grouped(qkv, kv, res, kn)
    for i in range(3):
        assert torch.equal(o[i], single[i][0]) and torch.equal(o_res[i], single[i][1]) and torch.equal(lse[i], single[i][2]), i
    dq1 = [ops.attn_bwd(q, oo, d, l, k) for q, oo, d, l, k in zip(qkv, o, do, lse, kv)]
    dqg = ops.attn_bwd_grouped(qkv, o, do, lse, kv)
    for i in range(3):
        assert torch.equal(dq1[i], dqg[i]), i
    # two streams, and a different order
    o2, _, _ = ops.attn_fwd_grouped(qkv[1:], kv[1:], [None, None], kn[1:])
    assert torch.equal(o2[0], single[1][0]) and torch.equal(o2[1], single[2][0])


def test_key_norms_table(ops):
    g = torch.Generator().manual_seed(11)
    for dt in DT:
        for B, N in ((3, 45), (2, 133), (1, 32), (5, 7)):
            qkv = (torch.randn(B, N, 768, generator=g) * torch.rand(B, N, 1, generator=g) * 3).to(dt)
            got = ops.key_norms(qkv.to(DEV)).cpu()
            k = qkv[..., 256:512].float().reshape(B * N, 4, 64).norm(dim=-1)            # [B N, 4]
            nblk = (B * N + 31) // 32
            want = torch.stack([k[32 * i:32 * i + 32].max(0).values for i in range(nblk)])
            assert got.shape == want.shape
            assert torch.allclose(got, want, rtol=1e-5, atol=0), (dt, B, N)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M", [64320, 6912, 17024, 300, 33, 4 * 133])
def test_ln_gemm_qkv_key_norm_table(ops, dt, M):
    """mtmp_ln_gemm_qkv = mtmp_ln_gemm (same qkv / xn / stats, bit for bit) + the key-norm table from its epilogue (bf16) or from
    mtmp_key_norms behind it (fp32).  The fused table is taken BEFORE the keys are rounded to bf16: within 2^-8 of the table of
    the stored keys (the attention kernel budgets 2 % for it)."""
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, 256, generator=g) * (1 + 3 * torch.rand(M, 1, generator=g))).to(DEV, dt)
    gm, bt = (1 + 0.1 * torch.randn(256, generator=g)).to(DEV), (0.1 * torch.randn(256, generator=g)).to(DEV)
    w = (torch.randn(768, 256, generator=g) * 0.08).to(DEV, dt)
    b = (torch.randn(768, generator=g) * 0.3).to(DEV)
    y0, xn0, st0 = ops.ln_gemm(x, gm, bt, w, b, 768)
    y1, xn1, st1, kn = ops.ln_gemm_qkv(x, gm, bt, w, b)
    assert torch.equal(y0, y1) and torch.equal(xn0, xn1) and torch.equal(st0, st1)
    ref = ops.key_norms(y1.view(1, M, 768))
    assert kn.shape == ref.shape == ((M + 31) // 32, 4)
    rel = float(((kn - ref).abs() / ref).max())
    REPORT[f"ln_gemm_qkv.knorm[{str(dt)[6:]},M={M}]"] = {"rel_err": rel, "tol": 5e-3 if dt == torch.bfloat16 else 0.0}
    assert rel <= (5e-3 if dt == torch.bfloat16 else 0.0), rel


@pytest.mark.parametrize("dt", DT)
def test_attention_bounded_body_adversarial(ops, dt):
    """The maximum-free body of the forward and its per-wave fallback on inputs built to break them (cdna guide rule 26:
    a data-dependent branch needs inputs that FORCE it): large norms (fallback), scores just inside the bound (huge and
    tiny exponentials, no shift), one outlier query row inside a block of tame ones (waves of one workgroup split),
    one-hot softmax rows, rows whose scores are all strongly negative.  Reference: the oracle in fp32 on the CPU."""
    g = torch.Generator().manual_seed(23)
    B, N = 3, 300
    base = torch.randn(B, N, 768, generator=g)
    lens = torch.tensor([300, 131, 64])
    c2 = 0.125 * 1.4426950408889634

    def run(qkv, tag, tol, expect=None):
        qkv = qkv.to(dt).float()
        o_ref = O.attention_core(qkv, lens)
        qd = qkv.to(DEV, dt)
        kn = ops.key_norms(qd)
        o_b, _, lse_b = ops.attn_fwd(qd, lens.to(DEV, torch.int32), knorm=kn)
        o_p, _, lse_p = ops.attn_fwd(qd, lens.to(DEV, torch.int32))
        assert torch.isfinite(o_b).all() and torch.isfinite(lse_b).all(), tag
        check(f"attn_adv[{str(dt)[6:]},{tag}].bounded", o_b.float(), o_ref, tol)
        check(f"attn_adv[{str(dt)[6:]},{tag}].online", o_p.float(), o_ref, tol)
        check(f"attn_adv[{str(dt)[6:]},{tag}].lse", lse_b, lse_p, 1e-5 if dt == torch.float32 else 2e-2)
        # which body ran: the bound the kernel forms, per query row
        qn = qkv[..., :256].reshape(B, N, 4, 64).norm(dim=-1)                                # [B,N,4]
        kn_b = qkv[..., 256:512].reshape(B, N, 4, 64).norm(dim=-1).amax(1, keepdim=True)     # >= per-block table maxima of b
        frac_fast = float(((qn * kn_b * c2 * 1.02) <= 64).float().mean())
        if expect == "fallback":
            assert torch.equal(o_b, o_p), tag                  # every wave took the online body: bit-identical
            assert frac_fast == 0.0
        if expect == "fast":
            assert frac_fast == 1.0
        return o_b

    tol = TOL[dt]
    run(base, "tame", tol, "fast")
    x = base.clone(); x[..., :512] *= 3.2                       # ||q|| ||k|| c2 ~ 120: nobody qualifies
    run(x, "large_norms", tol, "fallback")
    x = base.clone(); x[..., :512] *= 1.7                       # bound ~ 35-45, scores up to ~ +-20: exp2 over 40 binades, unshifted
    run(x, "near_bound", tol * (1 if dt == torch.float32 else 2), "fast")
    x = base.clone(); x[1, 37, :256] *= 40.0; x[0, 290, :256] *= 25.0   # outlier query rows: their waves fall back, neighbours do not
    run(x, "outlier_rows", tol)
    x = base.clone()                                            # one-hot rows: query 5 of every sample matches key 9 at score ~ +45
    x[:, 5, :256] = 0; x[:, 9, 256:512] = 0
    x[:, 5, 0:256:64] = 17.0; x[:, 9, 256:512:64] = 15.0
    run(x, "one_hot", tol)
    x = base.clone()                                            # every score of head 2 strongly negative (~ -40 .. -50)
    u = torch.randn(64, generator=g); u = u / u.norm()
    x[..., 128:192] = 16.0 * u + 0.05 * x[..., 128:192]
    x[..., 384:448] = -16.0 * u + 0.05 * x[..., 384:448]
    run(x, "all_negative", tol)


def test_attention_full_size_properties(ops):
    """Config-2 shape (B=64 is shrunk to 8 to keep the check quick; N, H, dh are the real ones):
    size-independent properties of softmax attention."""
    B, N = 8, 1005
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B, N, 768, generator=g).to(DEV, torch.bfloat16)
    lens = torch.randint(5, N + 1, (B,), generator=g).to(DEV, torch.int32)
    qkv[..., 512:] = 1.0                                    # V == 1  =>  O == 1 whatever the mask
    o, _, lse = ops.attn_fwd(qkv, lens)
    assert torch.all((o.float() - 1).abs() < 1e-2)
    # keys past kv_len are never read: poisoning them changes nothing, bit for bit
    qkv2 = torch.randn(B, N, 768, generator=g).to(DEV, torch.bfloat16)
    o1, _, _ = ops.attn_fwd(qkv2, lens)
    pois = qkv2.clone()
    for b in range(B):
        pois[b, int(lens[b]):, 256:] = float("nan")
    o2, _, _ = ops.attn_fwd(pois, lens)
    valid = torch.arange(N, device=DEV)[None, :] < lens[:, None]
    assert torch.equal(o1[valid], o2[valid])
    # linearity in V
    a = qkv2.clone(); a[..., 512:] *= 2
    o3, _, _ = ops.attn_fwd(a, lens)
    check("attn.fullsize.linearity", o3.float(), 2 * o1.float(), 1e-2)


# ------------------------------------------------------------------ projections / LN
@pytest.mark.parametrize("dt", DT)
def test_ln_gemm_and_gemm_nt(ops, dt):
    g = torch.Generator().manual_seed(5)
    M = 300
    x = (torch.randn(M, 256, generator=g) * 2 + 0.3).to(dt).float()
    gam, bet = 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    for N, relu in ((768, False), (1024, True)):
        w = (torch.randn(N, 256, generator=g) / 16).to(dt).float()
        b = 0.1 * torch.randn(N, generator=g)
        xn_ref = O.custom_layernorm(x, gam, bet)
        y_ref = torch.nn.functional.linear(xn_ref.to(dt).float() if dt != torch.float32 else xn_ref, w, b)
        if relu:
            y_ref = torch.relu(y_ref)
        y, xn, st = ops.ln_gemm(x.to(DEV, dt), gam.to(DEV), bet.to(DEV), w.to(DEV, dt), b.to(DEV), N, relu=relu)
        t = f"ln_gemm[{str(dt)[6:]},N={N}]"
        check(t + ".xn", xn.float(), xn_ref, TOL[dt] if dt == torch.float32 else 1e-2)
        check(t + ".y", y.float(), y_ref, TOL[dt] if dt == torch.float32 else 2e-2)
        check(t + ".mean", st[:, 0], x.mean(-1), 1e-5)
        check(t + ".rstd", st[:, 1], 1 / (x.std(-1) + 1e-6), 1e-5)
    a = torch.randn(M, 1024, generator=g).to(dt).float()
    w2 = (torch.randn(256, 1024, generator=g) / 32).to(dt).float()
    b2 = 0.1 * torch.randn(256, generator=g)
    r = torch.randn(M, 256, generator=g).to(dt).float()
    y = ops.gemm_nt(a.to(DEV, dt), w2.to(DEV, dt), b2.to(DEV), res2d=r.to(DEV, dt))
    yr = torch.nn.functional.linear(a, w2, b2)
    yr = (yr.to(dt).float() if dt != torch.float32 else yr) + r
    check(f"gemm_nt[{str(dt)[6:]}].y", y.float(), yr, TOL[dt] if dt == torch.float32 else 2e-2)


def test_reduce_batch_deferred_gradient_reductions(ops):
    """gemm_tn / gemm_lnbwd with their reductions deferred into ONE mtmp_reduce_batch launch against the immediate form."""
    g = torch.Generator(device=DEV).manual_seed(21)
    bf = torch.bfloat16
    for M in (3456, 20033, 64320):
        dy = torch.randn(M, 768, generator=g, device=DEV).to(bf)
        x = torch.randn(M, 256, generator=g, device=DEV).to(bf)
        dh = torch.randn(M, 1024, generator=g, device=DEV).to(bf)
        z = (torch.randn(M, 256, generator=g, device=DEV) * 2).to(bf)
        gam = 1 + 0.1 * torch.randn(256, generator=g, device=DEV)
        st = torch.stack([z.float().mean(-1), 1 / (z.float().std(-1) + 1e-6)], 1).contiguous()
        w1t = (torch.randn(256, 1024, generator=g, device=DEV) / 32).to(bf)
        dw0, db0 = ops.gemm_tn(dy, x)
        dv0, _ = ops.gemm_tn(dh, x, want_bias=False)
        dz0, dg0, dbt0 = ops.gemm_lnbwd(dh, w1t, z, st, gam)
        red = []
        dw1, db1 = ops.gemm_tn(dy, x, defer=red)
        dv1, _ = ops.gemm_tn(dh, x, want_bias=False, defer=red)
        dz1, dg1, dbt1 = ops.gemm_lnbwd(dh, w1t, z, st, gam, defer=red)
        assert len(red) == 3
        ops.reduce_batch(red)
        assert red == [] and torch.equal(dz1, dz0)
        for name, a, b in (("dw", dw1, dw0), ("db", db1, db0), ("dv", dv1, dv0), ("dgamma", dg1, dg0), ("dbeta", dbt1, dbt0)):
            check(f"reduce_batch[M={M}].{name}", a, b, 1e-5)


def test_copy_batch(ops):
    g = torch.Generator(device=DEV).manual_seed(11)
    srcs = [torch.randn(n, generator=g, device=DEV) * 300 for n in (64 * 1000 * 3, 64, 7, 1, 4099)]
    srcs += [torch.randint(0, 1000, (64,), generator=g, device=DEV), torch.randn(5, 3, generator=g, device=DEV).to(torch.bfloat16)]
    dsts = [torch.full_like(s, 7) for s in srcs]
    r16 = [True, True, False, True, True, False, False]
    ops.copy_batch(dsts, srcs, r16)
    for d, s, r in zip(dsts, srcs, r16):
        assert torch.equal(d, s.half().float() if r else s)
    assert not ops.copy_batch_ok(dsts[0][1:], srcs[0][1:])           # not 16-byte aligned: the caller falls back to copy_()


def test_transpose_batch(ops):
    g = torch.Generator(device=DEV).manual_seed(3)
    for dt in (torch.bfloat16, torch.float32):
        mats = [torch.randn(r, c, generator=g, device=DEV).to(dt) for r, c in
                [(768, 256), (256, 1024), (1024, 256), (70, 33), (1, 5), (65, 64)] * 12]        # 72 matrices: two launches
        outs = ops.transpose_batch(mats)
        assert all(torch.equal(o, m.t()) and o.is_contiguous() for o, m in zip(outs, mats))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M", [4990, 20001, 33000])
def test_row_panel_kernels_multi_panel(ops, dt, M):
    """The row-panel kernels walk 1 / 2 / 4-6 / all 12-16 weight panels per workgroup depending on M (launch_ln_gemm*): every
    loop shape of mtmp_ln_gemm and of the gated K = 256 mtmp_gemm_nt (dH of the FFN backward) against fp32 torch on the GPU."""
    g = torch.Generator(device=DEV).manual_seed(M)
    x = (torch.randn(M, 256, generator=g, device=DEV) * 2 + 0.3).to(dt)
    gam = 1 + 0.1 * torch.randn(256, generator=g, device=DEV)
    bet = 0.1 * torch.randn(256, generator=g, device=DEV)
    xf = x.float()
    mu, sd = xf.mean(-1, keepdim=True), xf.std(-1, keepdim=True)
    xn_ref = gam * (xf - mu) / (sd + 1e-6) + bet
    for N, relu in ((768, False), (1024, True)):
        w = (torch.randn(N, 256, generator=g, device=DEV) / 16).to(dt)
        b = 0.1 * torch.randn(N, generator=g, device=DEV)
        y_ref = torch.nn.functional.linear(xn_ref.to(dt).float(), w.float(), b)
        if relu:
            y_ref = torch.relu(y_ref)
        y, xn, st = ops.ln_gemm(x, gam, bet, w, b, N, relu=relu)
        t = f"ln_gemm.panels[{str(dt)[6:]},M={M},N={N}]"
        check(t + ".xn", xn.float(), xn_ref, TOL[dt] if dt == torch.float32 else 1e-2)
        check(t + ".y", y.float(), y_ref, TOL[dt] if dt == torch.float32 else 2e-2)
        assert torch.isfinite(y.float()).all()
    dy = torch.randn(M, 256, generator=g, device=DEV).to(dt)
    w2t = (torch.randn(1024, 256, generator=g, device=DEV) / 16).to(dt)
    h = torch.relu(torch.randn(M, 1024, generator=g, device=DEV)).to(dt)
    dh = ops.gemm_nt(dy, w2t, gate=h, gate_scale=1.0 / 0.9)
    ref = torch.nn.functional.linear(dy.float(), w2t.float())
    ref = torch.where(h.float() > 0, (ref.to(dt).float() if dt != torch.float32 else ref) * (1.0 / 0.9), torch.zeros_like(ref))
    check(f"gemm_nt.gated.panels[{str(dt)[6:]},M={M}]", dh.float(), ref, TOL[dt] if dt == torch.float32 else 2e-2)
    assert torch.all(dh.float()[h.float() <= 0] == 0)
    # the same product gated by the forward's sign bits (bf16): bits of the ReLU (+ dropout) output, then dH against fp32 torch
    for p_drop in (0.0, 0.1):
        w1 = (torch.randn(1024, 256, generator=g, device=DEV) / 16).to(dt)
        hh, _, _, signs = ops.ln_gemm(x, gam, bet, w1, None, 1024, relu=True, drop_p=p_drop, seed=77, want_signs=True)
        if dt == torch.float32:
            assert signs is None
            continue
        dh2 = ops.gemm_nt_signs(dy, w2t, signs, 1.0 / (1.0 - p_drop))
        ref2 = torch.nn.functional.linear(dy.float(), w2t.float()) * (1.0 / (1.0 - p_drop))
        ref2 = torch.where(hh.float() > 0, ref2, torch.zeros_like(ref2))
        check(f"gemm_nt_signs.panels[M={M},p={p_drop}]", dh2.float(), ref2, 2e-2)
        assert torch.equal(dh2.float() != 0, (hh.float() > 0) & (dh2.float() != 0)) and torch.all(dh2.float()[hh.float() <= 0] == 0)
        nz = (hh.float() > 0) & (ref2.abs() > 1e-3)
        assert torch.all(dh2.float()[nz] != 0)


def test_gemm_nt_signs_with_dropout_backward_folded_in(ops):
    """mtmp_gemm_nt_signs_drop (the FFN backward's dH product with drop2's backward applied to its operand in registers) against
    mtmp_dropout_bwd followed by mtmp_gemm_nt_signs: the masked operand it hands on and dH are bit-identical."""
    g = torch.Generator(device=DEV).manual_seed(5)
    bf = torch.bfloat16
    for M, p in ((4990, 0.1), (20001, 0.3), (64320, 0.1), (129, 0.5)):
        x = torch.randn(M, 256, generator=g, device=DEV).to(bf)
        gam, bet = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
        w1 = (torch.randn(1024, 256, generator=g, device=DEV) / 16).to(bf)
        w2t = (torch.randn(1024, 256, generator=g, device=DEV) / 16).to(bf)
        dy = torch.randn(M, 256, generator=g, device=DEV).to(bf)
        _, _, _, signs = ops.ln_gemm(x, gam, bet, w1, None, 1024, relu=True, drop_p=p, seed=77, want_signs=True)
        dy2_ref = ops.dropout_bwd(dy, 1234, p)
        dh_ref = ops.gemm_nt_signs(dy2_ref, w2t, signs, 1.0 / (1.0 - p))
        dh, dy2 = ops.gemm_nt_signs(dy, w2t, signs, 1.0 / (1.0 - p), drop_p=p, seed=1234)
        assert torch.equal(dy2, dy2_ref), f"masked operand differs (M={M}, p={p})"
        assert torch.equal(dh, dh_ref), f"dH differs (M={M}, p={p})"
        kept = float((dy2 != 0).float().mean())
        assert abs(kept - (1 - p)) < 0.02


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(3456, 768, 256), (1000, 256, 1024), (70, 1024, 256), (20033, 256, 1024)])
def test_gemm_tn_weight_gradient(ops, dt, M, N, K):
    g = torch.Generator().manual_seed(M)
    dy = torch.randn(M, N, generator=g).to(dt).float()
    x = torch.randn(M, K, generator=g).to(dt).float()
    dw, db = ops.gemm_tn(dy.to(DEV, dt), x.to(DEV, dt))
    t = f"gemm_tn[{str(dt)[6:]},M={M},N={N}]"
    check(t + ".dw", dw, dy.double().t() @ x.double(), 1e-4 if dt == torch.float32 else 1e-4)   # fp32 accumulate either way
    check(t + ".db", db, dy.double().sum(0), 1e-4)


@pytest.mark.parametrize("M,N,K", [(64320, 768, 256), (64320, 256, 1024), (12345, 768, 256), (40001, 256, 768), (20033, 1024, 256),
                                   (16447, 768, 256)])
def test_gemm_tn_large_m_dma_tiles(ops, M, N, K):
    """bf16, large M: the LDS-DMA kernel (two token groups per workgroup): token tails that end inside a stage, a split with no
    rows at all (M = 12345), K tiles that do not divide the bias share evenly (K = 768), operands that are column windows of wider
    buffers (row stride > width), against a float64 product of the same bf16 inputs."""
    g = torch.Generator().manual_seed(M + N)
    wide_y = torch.randn(M, N + 64, generator=g).to(torch.bfloat16).to(DEV)
    wide_x = torch.randn(M, K + 128, generator=g).to(torch.bfloat16).to(DEV)
    for dy, x in ((wide_y[:, :N].contiguous(), wide_x[:, :K].contiguous()), (wide_y[:, 64:], wide_x[:, 128:])):
        dw, db = ops.gemm_tn(dy, x)
        t = f"gemm_tn.dma[M={M},N={N},K={K},ld={dy.stride(0)}]"
        check(t + ".dw", dw, (dy.double().t() @ x.double()).cpu(), 1e-4)
        check(t + ".db", db, dy.double().sum(0).cpu(), 1e-4)
        dw2, db2 = ops.gemm_tn(dy, x)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)          # deterministic (no atomics)
    # rows that are only 4-byte aligned cannot go through the DMA kernel: the register-staged kernel at the same split count
    dy, x = wide_y[:, 2:N + 2], wide_x[:, 6:K + 6]
    dw, db = ops.gemm_tn(dy, x)
    check(f"gemm_tn.unaligned[M={M},N={N},K={K}].dw", dw, (dy.double().t() @ x.double()).cpu(), 1e-4)
    check(f"gemm_tn.unaligned[M={M},N={N},K={K}].db", db, dy.double().sum(0).cpu(), 1e-4)


@pytest.mark.parametrize("dt", DT)
def test_ln_bwd(ops, dt, golden_dir):
    g = torch.Generator().manual_seed(9)
    M = 777
    z = (torch.randn(M, 256, generator=g) * 2 + 0.3).to(dt).float().requires_grad_()
    gam = (1 + 0.1 * torch.randn(256, generator=g)).requires_grad_()
    bet = (0.1 * torch.randn(256, generator=g)).requires_grad_()
    dy = torch.randn(M, 256, generator=g).to(dt).float()
    dres = torch.randn(M, 256, generator=g).to(dt).float()
    y = O.custom_layernorm(z, gam, bet)
    (y * dy).sum().backward()
    mu = z.detach().mean(-1)
    rs = 1 / (z.detach().std(-1) + 1e-6)
    st = torch.stack([mu, rs], 1).to(DEV)
    dz, dg, db = ops.ln_bwd(z.detach().to(DEV, dt), st, gam.detach().to(DEV), dy.to(DEV, dt), dres.to(DEV, dt))
    t = f"ln_bwd[{str(dt)[6:]}]"
    check(t + ".dz", dz.float(), z.grad + dres, TOL[dt] if dt == torch.float32 else 2e-2)
    check(t + ".dgamma", dg, gam.grad, 1e-4 if dt == torch.float32 else 1e-2)
    check(t + ".dbeta", db, bet.grad, 1e-4 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,K", [(777, 768), (128, 1024), (3456, 1024), (50, 768)])
def test_gemm_lnbwd_vs_autograd_of_ln_then_linear(ops, dt, M, K):
    """mtmp_gemm_lnbwd = the dX product of a LayerNorm-fed projection with the LayerNorm backward as its epilogue:
    against autograd through the oracle's custom LayerNorm followed by a Linear (module.py:138-144 + attention.py:68-70 /
    module.py:74-77), with the residual-branch gradient added; M not a multiple of the 128-row tile, one tile, many."""
    g = torch.Generator().manual_seed(M + K)
    z = (torch.randn(M, 256, generator=g) * 2 + 0.3).to(dt).float().requires_grad_()
    gam = (1 + 0.1 * torch.randn(256, generator=g)).requires_grad_()
    bet = (0.1 * torch.randn(256, generator=g)).requires_grad_()
    W = (torch.randn(K, 256, generator=g) / 16).to(dt).float()
    dy = torch.randn(M, K, generator=g).to(dt).float()
    dres = torch.randn(M, 256, generator=g).to(dt).float()
    xn = O.custom_layernorm(z, gam, bet)
    if dt == torch.bfloat16:          # the stand-alone path rounds the M x 256 product to bf16 before the LN backward
        dxn = (dy @ W).to(dt).float()
        (xn * dxn).sum().backward()
    else:
        ((xn @ W.t()) * dy).sum().backward()
    mu = z.detach().mean(-1)
    rs = 1 / (z.detach().std(-1) + 1e-6)
    st = torch.stack([mu, rs], 1).to(DEV)
    dz, dg, db = ops.gemm_lnbwd(dy.to(DEV, dt), W.t().contiguous().to(DEV, dt), z.detach().to(DEV, dt), st, gam.detach().to(DEV),
                                dres.to(DEV, dt))
    t = f"gemm_lnbwd[{str(dt)[6:]},M={M},K={K}]"
    check(t + ".dz", dz.float(), z.grad + dres, TOL[dt] if dt == torch.float32 else 2e-2)
    check(t + ".dgamma", dg, gam.grad, 1e-4 if dt == torch.float32 else 1e-2)
    check(t + ".dbeta", db, bet.grad, 1e-4 if dt == torch.float32 else 1e-2)
    dz0, _, _ = ops.gemm_lnbwd(dy.to(DEV, dt), W.t().contiguous().to(DEV, dt), z.detach().to(DEV, dt), st, gam.detach().to(DEV))
    check(t + ".dz_no_residual", dz0.float(), z.grad, TOL[dt] if dt == torch.float32 else 2e-2)


def test_ln_gemm_dropout_statistics(ops):
    """Dropout inside mtmp_ln_gemm (module.py:77-79 after the ReLU): kept elements are the no-dropout values / (1-p), the keep
    fraction is 1-p overall, per row and per column, a different seed gives a different mask, the same seed the same one."""
    g = torch.Generator().manual_seed(7)
    M, N, p = 1536, 1024, 0.1
    x = torch.randn(M, 256, generator=g).to(DEV)
    gm, bt = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
    w = (torch.randn(N, 256, generator=g) / 16).to(DEV)
    b = torch.full((N,), 3.0, device=DEV)            # large bias: nearly every pre-dropout value is positive
    y0, _, _ = ops.ln_gemm(x, gm, bt, w, b, N)
    y1, _, _ = ops.ln_gemm(x, gm, bt, w, b, N, drop_p=p, seed=4242)
    pos = y0 > 0
    kept = (y1 != 0) & pos
    frac = float(kept.sum()) / float(pos.sum())
    REPORT["ln_gemm.dropout.keep_fraction"] = {"rel_err": abs(frac - (1 - p)), "tol": 0.005}
    assert abs(frac - (1 - p)) < 0.005, frac
    assert torch.allclose(y1[kept], y0[kept] / (1 - p), rtol=1e-5, atol=1e-6)
    rows, cols = kept.float().mean(1), kept.float().mean(0)
    assert float((rows - (1 - p)).abs().max()) < 0.06 and float((cols - (1 - p)).abs().max()) < 0.06
    # no structure along the feature axis: neighbouring decisions are uncorrelated
    k = kept.float() - kept.float().mean()
    for lag in (1, 2, 4, 8, 32, 64):
        c = float((k[:, :-lag] * k[:, lag:]).mean() / k.var())
        assert abs(c) < 0.01, (lag, c)
    y2, _, _ = ops.ln_gemm(x, gm, bt, w, b, N, drop_p=p, seed=4242)
    y3, _, _ = ops.ln_gemm(x, gm, bt, w, b, N, drop_p=p, seed=4243)
    assert torch.equal(y1, y2) and not torch.equal(y1 != 0, y3 != 0)


def test_dropout_mask_consistency(ops):
    g = torch.Generator().manual_seed(1)
    M, p, seed = 512, 0.1, 1234567
    a = torch.randn(M, 1024, generator=g).to(DEV)
    w = torch.randn(256, 1024, generator=g).to(DEV) / 32
    y0 = ops.gemm_nt(a, w)
    y1 = ops.gemm_nt(a, w, drop_p=p, seed=seed)
    kept = y1 != 0
    frac = float(kept.float().mean())
    REPORT["dropout.keep_fraction"] = {"rel_err": abs(frac - (1 - p)), "tol": 0.01}
    assert abs(frac - (1 - p)) < 0.01
    assert torch.allclose(y1[kept], y0[kept] / (1 - p), rtol=1e-5, atol=1e-6)
    gmask = ops.dropout_bwd(torch.ones_like(y0), seed, p)
    assert torch.equal(gmask != 0, kept)                     # backward regenerates the same mask
    y2 = ops.gemm_nt(a, w, drop_p=p, seed=seed + 1)
    assert not torch.equal(y2 != 0, kept)


# ------------------------------------------------------------------ TIE / stem / AdamW
_MODEL_SD = {}


def _model_sd(L):
    """closed-form weights of the reference's state_dict at L layers (made once per L: ~30 M values through the hash filler;
    the tensors are shared between callers, which load / clone them and never write into them)"""
    if L not in _MODEL_SD:
        sd = {k: filler.fill_tensor(k, torch.zeros(s)) for k, s in reference_state_shapes(L).items()}
        sd["fusion_transformer.positional_encoding.pe"] = O.sinusoid_table(2500, 256).unsqueeze(0)
        _MODEL_SD[L] = sd
    return dict(_MODEL_SD[L])


@pytest.mark.parametrize("dt", DT)
def test_tie_embedding(ops, dt):
    Gd = G("model_step")
    sd = _model_sd(2)
    bt = filler.make_batch(int(Gd["seed"]), int(Gd["B"]), int(Gd["T"]))
    prm = [sd["ie_vslt.0.weight"], sd["ie_vslt.0.bias"], sd["ie_vslt.1.weight"], sd["ie_vslt.1.bias"],
           sd["ie_time.0.weight"], sd["ie_time.0.bias"], sd["ie_time.1.weight"], sd["ie_time.1.bias"],
           sd["ie_feat.weight"]]
    prm_d = [p.clone().to(DEV).requires_grad_() for p in prm]
    emb = ops.TieEmbed.apply(bt["x"].to(DEV), *prm_d, dt)
    t = f"tie[{str(dt)[6:]}]"
    check(t + ".emb_vs_golden", emb.float(), torch.from_numpy(Gd["tie_emb"]), TOL[dt] if dt == torch.float32 else 1e-2)
    w = torch.from_numpy(Gd["tie_w"]).to(DEV)
    (emb.float() * w.to(dt).float()).sum().backward()
    if dt == torch.float32:
        check(t + ".dWv", prm_d[0].grad, torch.from_numpy(Gd["tie_dWv"]), 1e-4)
        check(t + ".dbt", prm_d[5].grad, torch.from_numpy(Gd["tie_dbt"]), 1e-4)
        check(t + ".dgv", prm_d[2].grad, torch.from_numpy(Gd["tie_dgv"]), 1e-4)
        check(t + ".dF", prm_d[8].grad, torch.from_numpy(Gd["tie_dF"]), 1e-4)
    else:
        check(t + ".dF", prm_d[8].grad, torch.from_numpy(Gd["tie_dF"]), 2e-2)


@pytest.mark.parametrize("dt", DT)
def test_tie_embedding_packed_equals_padded(ops, dt):
    """Ragged batch layout: same kernel arithmetic through the cu_seqlens row map -- valid rows are
    bit-identical to the padded path, pad rows are zero, parameter gradients agree (summation order differs)."""
    from medical_tri_modal_pilot_amd.builder.data import collate_packed
    sd = _model_sd(2)
    prm = [sd["ie_vslt.0.weight"], sd["ie_vslt.0.bias"], sd["ie_vslt.1.weight"], sd["ie_vslt.1.bias"],
           sd["ie_time.0.weight"], sd["ie_time.0.bias"], sd["ie_time.1.weight"], sd["ie_time.1.bias"],
           sd["ie_feat.weight"]]
    g = torch.Generator().manual_seed(21)
    lens = [37, 1, 200, 64, 0, 199]
    T = 200
    x = torch.zeros(len(lens), T, 3)
    for b, n in enumerate(lens):
        x[b, :n, 0] = -24 * torch.rand(n, generator=g)
        x[b, :n, 1] = torch.rand(n, generator=g)
        x[b, :n, 2] = torch.randint(0, 18, (n,), generator=g).float()
    pb = collate_packed([(x[b, :n].numpy(), np.zeros(2, np.float32), 0.0) for b, n in enumerate(lens)])
    pk = pb.on_device(DEV, t_pad=T, bucket=256)
    xr = x.half().float()                                   # the trainer's fp16 rounding, as on_device applies it
    p1 = [p.clone().to(DEV).requires_grad_() for p in prm]
    p2 = [p.clone().to(DEV).requires_grad_() for p in prm]
    e_pad = ops.TieEmbed.apply(xr.to(DEV), *p1, dt)
    e_pk = ops.TieEmbedPacked.apply(pk.events, pk.cu_seqlens, pk.t_pad, *p2, dt)
    assert e_pk.shape == e_pad.shape
    valid = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).to(DEV)
    assert torch.equal(e_pk[valid], e_pad[valid])
    assert float(e_pk[~valid].float().abs().sum()) == 0.0
    w = torch.randn(len(lens), T, 256, generator=g).to(DEV)
    (e_pad.float() * w * valid[..., None]).sum().backward()
    (e_pk.float() * w).sum().backward()                     # pad rows carry no gradient by construction
    for i, (a, b) in enumerate(zip(p1, p2)):
        check(f"tie_packed[{str(dt)[6:]}].grad{i}", b.grad, a.grad, 1e-5 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("use_pe", [False, True])
def test_stream_input_vs_torch_chain(ops, dt, use_pe):
    """K4 (mbt_encoder.py:697-729 + the concatenation of :745): the fused launch against the op chain it replaces
    (cat CLS -> F.layer_norm fp32 -> + sinusoid rows -> cast, bottleneck rows in front), forward and all gradients;
    then the dropout mask: the backward regenerates exactly the forward's mask."""
    from oracle import tri_mbt_oracle as O
    g = torch.Generator().manual_seed(5 + int(use_pe))
    B, N, nb = 3, 37, 4
    x = torch.randn(B, N, 256, generator=g).to(dt)
    cls, gam, bet = torch.randn(1, 1, 256, generator=g), 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    bott = torch.randn(1, nb, 256, generator=g)
    pe = O.sinusoid_table(2500, 256)[:N + 1] if use_pe else None
    w = torch.randn(B, nb + 1 + N, 256, generator=g)
    # reference chain on the CPU in fp32 (inputs rounded to dt like the product sees them)
    xr, cr, gr, br, tr = (t.clone().float().requires_grad_() for t in (x, cls, gam, bet, bott))
    xc = torch.cat([cr.expand(B, -1, -1).to(dt).float(), xr], 1)
    y = F_layer_norm(xc, gr, br)
    if use_pe:
        y = y + pe[None]
    z_ref = torch.cat([tr.expand(B, -1, -1), y], 1)
    (z_ref * w).sum().backward()
    xd, cd, gd, bd, td = (t.clone().to(DEV).requires_grad_() for t in (x, cls, gam, bet, bott))
    z = ops.StreamInputFn.apply(xd, cd, gd, bd, None if pe is None else pe.to(DEV), td, 1e-5, 0.0, 0)
    t = f"stream_in[{str(dt)[6:]},pe={int(use_pe)}]"
    tol = 1e-5 if dt == torch.float32 else 1e-2
    check(t + ".z", z.float(), z_ref, tol)
    (z.float() * w.to(DEV)).sum().backward()
    check(t + ".dx", xd.grad.float(), xr.grad, 1e-4 if dt == torch.float32 else 2e-2)
    check(t + ".dcls", cd.grad, cr.grad, 1e-4 if dt == torch.float32 else 2e-2)
    check(t + ".dgamma", gd.grad, gr.grad, 1e-4 if dt == torch.float32 else 2e-2)
    check(t + ".dbeta", bd.grad, br.grad, 1e-4 if dt == torch.float32 else 2e-2)
    check(t + ".dbott", td.grad, tr.grad, 1e-5 if dt == torch.float32 else 1e-2)
    # dropout: kept elements are scaled by 1/(1-p), dropped ones are exactly 0 in z and get exactly 0 gradient
    p = 0.25
    x2 = xd.detach().clone().requires_grad_()
    zd = ops.StreamInputFn.apply(x2, cd.detach(), gd.detach(), bd.detach(), None, td.detach(), 1e-5, p, 1234)
    body, body0 = zd[:, nb:].float(), z[:, nb:].float().detach() - (0 if pe is None else pe.to(DEV)[None])
    kept = body != 0
    frac = float(kept.float().mean())
    assert abs(frac - (1 - p)) < 0.02, frac
    assert torch.equal(zd[:, :nb], z[:, :nb].detach())                      # bottleneck rows are never dropped
    if pe is None:
        check(t + ".dropout_scale", body[kept], body0[kept] / (1 - p), tol)
    # d(sum z)/dx with everything kept would be rstd*(gamma - mean - xh*mean(gamma*xh)); a fully dropped row gives 0
    zd.float().sum().backward()
    row_dropped = ~kept.any(dim=2)[:, 1:]
    assert float(x2.grad[row_dropped].float().abs().sum()) == 0.0


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("flat", [False, True])
def test_stream_inputs_node_vs_three_stream_input_nodes(ops, dt, flat):
    """ops.StreamInputsFn (the three streams' input kernels as ONE autograd node: mtmp_stream_input_bwd_grouped, mtmp_token_sums,
    one mtmp_reduce_scatter -- into autograd tensors, or straight into an optim.FlatParams gradient buffer) against three
    ops.StreamInputFn nodes with the time embeddings added through torch in front of them (the form of rounds 1-4, itself pinned by
    test_stream_input_vs_torch_chain): outputs bit-identical (same forward launches), every gradient to rounding -- with dropout
    (same seeds: same masks), a PACKED first stream, K = 2 images per sample (two time embeddings per sample's image tokens)."""
    from medical_tri_modal_pilot_amd.optim import FlatParams
    from oracle import tri_mbt_oracle as O
    g = torch.Generator().manual_seed(11)
    B, nb, K = 3, 4, 2
    Ns = [37, 2 * 7, 9]
    R = lambda *sh: torch.randn(*sh, generator=g)
    xs = [R(B, n, 256).to(dt) for n in Ns]
    add_i, add_t = R(B * K, 256).to(dt), R(B, 256).to(dt)
    cls = [R(1, 1, 256) for _ in range(3)]
    gam = [1 + 0.1 * R(256) for _ in range(3)]
    bet = [0.1 * R(256) for _ in range(3)]
    bott = R(1, nb, 256)
    pe = [None, O.sinusoid_table(2500, 256)[:Ns[1] + 1], None]
    ws = [R(B, nb + 1 + n, 256) for n in Ns]
    kv = torch.tensor([nb + 1 + 37, nb + 1 + 5, nb + 1 + 20], dtype=torch.int32, device=DEV)
    pack = ops.row_starts(kv, nb + 1 + Ns[0])
    p_drop, seeds = 0.2, [17, 18, 19]

    def leaves():
        L = dict(x=[t.clone().to(DEV).requires_grad_() for t in xs], ai=add_i.clone().to(DEV).requires_grad_(),
                 at=add_t.clone().to(DEV).requires_grad_())
        mod = torch.nn.ParameterDict({f"cls{m}": torch.nn.Parameter(cls[m].clone()) for m in range(3)})
        for m in range(3):
            mod[f"g{m}"], mod[f"b{m}"] = torch.nn.Parameter(gam[m].clone()), torch.nn.Parameter(bet[m].clone())
        mod["bott"] = torch.nn.Parameter(bott.clone())
        return L, mod.to(DEV)

    def loss_of(zs):
        tot = 0.0
        for m, z in enumerate(zs):
            w = ws[m].to(DEV)
            if m == 0:                                    # packed stream: rows pack[b] .. pack[b] + kv[b] of the same allocation
                rows = torch.cat([int(pack[b]) + torch.arange(int(kv[b]), device=DEV) for b in range(B)])
                tot = tot + (z.reshape(-1, 256)[rows].float() * w.reshape(-1, 256)[:rows.numel()]).sum()
            else:
                tot = tot + (z.float() * w).sum()
        return tot

    # (a) three nodes, adds through torch
    La, Ma = leaves()
    xa = [La["x"][0], (La["x"][1].view(B * K, Ns[1] // K, 256) + La["ai"].unsqueeze(1)).view(B, Ns[1], 256),
          La["x"][2] + La["at"].unsqueeze(1)]
    za = []
    for m in range(3):
        pk = (pack, kv) if m == 0 else (None, None)
        za.append(ops.StreamInputFn.apply(xa[m], Ma[f"cls{m}"], Ma[f"g{m}"], Ma[f"b{m}"], None if pe[m] is None else pe[m].to(DEV),
                                          Ma["bott"], 1e-5, p_drop, seeds[m], *pk))
    loss_of(za).backward()
    # (b) one node
    Lb, Mb = leaves()
    fl = FlatParams(Mb.named_parameters()) if flat else None
    meta = dict(pe=[None if q is None else q.to(DEV) for q in pe], seeds=seeds, pack=(pack, kv), streams=None)
    prm = [Mb[k] for m in range(3) for k in (f"cls{m}", f"g{m}", f"b{m}")]
    zb = ops.StreamInputsFn.apply(Lb["x"][0], Lb["x"][1], Lb["x"][2], Mb["bott"], Lb["ai"], Lb["at"], 1e-5, 1e-5, 1e-5, p_drop, meta, *prm)
    rows0 = torch.cat([int(pack[b]) + torch.arange(int(kv[b]), device=DEV) for b in range(B)])
    assert torch.equal(za[0].reshape(-1, 256)[rows0], zb[0].reshape(-1, 256)[rows0]) and torch.equal(za[1], zb[1]) and torch.equal(za[2], zb[2])
    loss_of(zb).backward()
    if flat:                                            # the node wrote the ten slices itself and reported them
        assert all(fl.index_of[id(q)] in fl.written for q in Mb.values())
    t = f"stream_inputs_node[{str(dt)[6:]},flat={int(flat)}]"
    tol = 2e-5 if dt == torch.float32 else 2e-2
    for m in range(3):
        check(f"{t}.dx{m}", Lb["x"][m].grad.float(), La["x"][m].grad.float(), tol)
    check(t + ".d_add_img", Lb["ai"].grad.float(), La["ai"].grad.float(), tol)
    check(t + ".d_add_txt", Lb["at"].grad.float(), La["at"].grad.float(), tol)
    for k in Ma.keys():
        check(f"{t}.d_{k}", Mb[k].grad, Ma[k].grad, 2e-5 if dt == torch.float32 else 2e-2)


@pytest.mark.parametrize("dt", DT)
def test_time_embed_and_data_linear_vs_torch(ops, dt):
    """ie_time(t) + ie_feat(18|19) (tri_mbt_vsltcls.py:216-224) and the Linear(768,256) projections of data tensors
    (:200, :205-211) against the torch modules they replace, forward and parameter gradients."""
    g = torch.Generator().manual_seed(9)
    n_i, n_t = 6, 5
    t_all = torch.cat([-10 * torch.rand(n_i, generator=g), -torch.randint(3, 100, (n_t,), generator=g).float()]).half().float()
    lin, ln = torch.nn.Linear(1, 256), torch.nn.LayerNorm(256)
    ftab = torch.randn(20, 256, generator=g)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(256, generator=g)); ln.bias.copy_(0.1 * torch.randn(256, generator=g))
    fr = ftab.clone().requires_grad_()
    idx = torch.tensor([18] * n_i + [19] * n_t)
    ref = torch.relu(ln(lin(t_all[:, None]))) + fr[idx]
    w = torch.randn(n_i + n_t, 256, generator=g)
    (ref * w).sum().backward()
    ev = torch.zeros(n_i + n_t, 3); ev[:, 0] = t_all; ev[:, 2] = idx.float()
    prm = [p_.detach().clone().to(DEV).requires_grad_() for p_ in (lin.weight, lin.bias, ln.weight, ln.bias, ftab)]
    out = ops.TimeEmbed.apply(ev.to(DEV), *prm, dt)
    t = f"time_embed[{str(dt)[6:]}]"
    check(t + ".out", out.float(), ref, 1e-5 if dt == torch.float32 else 1e-2)
    (out.float() * w.to(DEV).to(dt).float()).sum().backward()
    tol = 1e-4 if dt == torch.float32 else 2e-2
    for nm, a, b in (("dw", prm[0].grad, lin.weight.grad), ("db", prm[1].grad, lin.bias.grad), ("dg", prm[2].grad, ln.weight.grad),
                     ("dbeta", prm[3].grad, ln.bias.grad), ("dftab", prm[4].grad, fr.grad)):
        check(f"{t}.{nm}", a, b, tol)
    # projection of a data tensor: forward + dW/db through mtmp_gemm_nt / mtmp_gemm_tn
    x = torch.randn(3, 128, 768, generator=g).to(dt).float()
    proj = torch.nn.Linear(768, 256)
    yr = proj(x)
    wy = torch.randn(3, 128, 256, generator=g)
    (yr * wy).sum().backward()
    pw, pb = proj.weight.detach().clone().to(DEV).requires_grad_(), proj.bias.detach().clone().to(DEV).requires_grad_()
    y = ops.DataLinearFn.apply(x.to(DEV), pw, pb, dt)
    check(f"data_linear[{str(dt)[6:]}].y", y.float(), yr, 1e-5 if dt == torch.float32 else 2e-2)
    (y.float() * wy.to(DEV).to(dt).float()).sum().backward()
    check(f"data_linear[{str(dt)[6:]}].dW", pw.grad, proj.weight.grad, 1e-4 if dt == torch.float32 else 3e-2)
    check(f"data_linear[{str(dt)[6:]}].db", pb.grad, proj.bias.grad, 1e-4 if dt == torch.float32 else 3e-2)


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B", [64, 5, 128, 200])
def test_head_vs_torch_modules(ops, training, B):
    """K10 (tri_mbt_vsltcls.py:59-76 ie_demo + :248-255 head): six HIP launches against the torch modules
    (Linear, LayerNorm, BatchNorm1d with its running statistics, ReLU), forward, every gradient, running stats."""
    torch.manual_seed(3)
    dm = torch.nn.Sequential(torch.nn.Linear(2, 256), torch.nn.LayerNorm(256), torch.nn.ReLU())
    ln = torch.nn.LayerNorm(256)
    fc = torch.nn.Sequential(torch.nn.Linear(512, 256), torch.nn.BatchNorm1d(256), torch.nn.ReLU(), torch.nn.Linear(256, 1))
    with torch.no_grad():
        for m_ in (dm[1], ln, fc[1]):
            m_.weight.add_(0.2 * torch.randn(256)); m_.bias.add_(0.2 * torch.randn(256))
        fc[1].running_mean.copy_(0.1 * torch.randn(256)); fc[1].running_var.copy_(1 + 0.2 * torch.rand(256))
    for m_ in (dm, ln, fc):
        m_.train(training)
    cls = torch.randn(B, 256).requires_grad_()
    age, gen = torch.rand(B), torch.randint(0, 2, (B,)).float()
    rm0, rv0 = fc[1].running_mean.clone(), fc[1].running_var.clone()
    ref = fc(torch.cat([ln(cls), dm(torch.stack([age, gen], 1))], 1))
    w = torch.randn(B, 1)
    (ref * w).sum().backward()
    prm_mods = [dm[0].weight, dm[0].bias, dm[1].weight, dm[1].bias, ln.weight, ln.bias, fc[0].weight, fc[0].bias,
                fc[1].weight, fc[1].bias, fc[3].weight, fc[3].bias]
    prm = [p_.detach().clone().to(DEV).requires_grad_() for p_ in prm_mods]
    rm, rv = rm0.clone().to(DEV), rv0.clone().to(DEV)
    cd = cls.detach().clone().to(DEV).requires_grad_()
    nbt = torch.tensor(41, dtype=torch.int64, device=DEV)
    out = ops.HeadFn.apply(cd, age.to(DEV), gen.to(DEV), training, 0.1, 1e-5, rm, rv, nbt if training else None, *prm)
    assert int(nbt) == (42 if training else 41)                  # BatchNorm1d.num_batches_tracked, counted by the head's first launch
    t = f"head[B={B},train={int(training)}]"
    check(t + ".out", out, ref, 1e-5)
    (out * w.to(DEV)).sum().backward()
    check(t + ".dcls", cd.grad, cls.grad, 1e-4)
    names = ["demo_w", "demo_b", "demo_g", "demo_be", "ln_g", "ln_b", "w1", "b1", "bn_g", "bn_b", "w2", "b2"]
    scale = float(prm_mods[4].grad.abs().max())
    for nm, a, b in zip(names, prm, prm_mods):
        if training and nm in ("ln_b", "b1"):
            # BatchNorm removes the batch mean of h, so sum_b dh = 0: these two gradients are mathematically zero and
            # both sides hold rounding noise (cf. the key-projection bias in DESIGN.md 2)
            assert float(a.grad.abs().max()) < 1e-4 * scale and float(b.grad.abs().max()) < 1e-4 * scale, nm
            continue
        check(f"{t}.d{nm}", a.grad, b.grad, 1e-4)
    if training:
        check(t + ".running_mean", rm, fc[1].running_mean, 1e-6)
        check(t + ".running_var", rv, fc[1].running_var, 1e-6)
    else:
        assert torch.equal(rm.cpu(), rm0) and torch.equal(rv.cpu(), rv0)
        # the CLS vectors in the fusion stack's own type: read as bf16, their gradient written as bf16 (no cast launches around the head)
        cb = cls.detach().to(torch.bfloat16).to(DEV).requires_grad_()
        prm2 = [p_.detach().clone().to(DEV).requires_grad_() for p_ in prm_mods]
        out2 = ops.HeadFn.apply(cb, age.to(DEV), gen.to(DEV), False, 0.1, 1e-5, rm, rv, None, *prm2)
        c2 = cls.detach().to(torch.bfloat16).float().requires_grad_()
        ref2 = fc(torch.cat([ln(c2), dm(torch.stack([age, gen], 1))], 1))
        (ref2 * w).sum().backward()
        check(t + ".out[bf16 cls]", out2, ref2, 1e-5)
        (out2 * w.to(DEV)).sum().backward()
        assert cb.grad.dtype == torch.bfloat16
        check(t + ".dcls[bf16 cls]", cb.grad.float(), c2.grad, 1e-2)          # (bf16 rounding of the result)


def test_bce_logits_mean_vs_torch(ops):
    g = torch.Generator().manual_seed(2)
    for n in (1, 4, 64, 300):
        o = (4 * torch.randn(n, generator=g)).requires_grad_()
        t = torch.randint(0, 2, (n,), generator=g).float()
        ref = torch.nn.functional.binary_cross_entropy_with_logits(o, t)
        (3.0 * ref).backward()
        od = o.detach().clone().to(DEV).requires_grad_()
        loss = ops.bce_with_logits(torch.nn.BCEWithLogitsLoss(), od, t.to(DEV))
        (3.0 * loss).backward()
        check(f"bce[{n}].loss", loss, ref, 1e-6)
        check(f"bce[{n}].dlogit", od.grad, o.grad, 1e-5)


def F_layer_norm(x, w, b):
    return torch.nn.functional.layer_norm(x, (256,), w, b, 1e-5)


def _product_model(L, multi, dtype, **over):
    from medical_tri_modal_pilot_amd.control.config import parse_args
    from medical_tri_modal_pilot_amd.builder.models import get_model
    a = parse_args(["--input-types", "vslt_img_txt", "--model", "tri_mbt_vsltcls", "--modality-inclusion",
                    "train-missing_test-missing", "--lr-init", "1e-5", "--batch-size", "4",
                    "--transformer-num-layers", str(L), "--imgtxt-time", "1", "--mbt-only-vslt", "1",
                    "--multiimages", str(multi), "--dropout", "0.0", "--compute-dtype", dtype])
    a.device = torch.device(DEV)
    for k, v in over.items():
        setattr(a, k, v)
    model = get_model(a)(a)
    model.load_state_dict(_model_sd(L), strict=False)         # integer buffers keep their built values
    return a, model.to(DEV)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_swin_stem_and_features(ops, dtype):
    Gd = G("swin")
    _, model = _product_model(2, 0, dtype)
    model.eval()
    g = torch.Generator().manual_seed(int(Gd["seed"]))
    img = torch.rand(2, 1, 224, 224, generator=g).to(DEV)
    enc = model.img_encoder
    st = enc.features[0]
    stem = ops.swin_stem(img, st[0].weight, st[0].bias, st[2].weight, st[2].bias, model.compute_dtype)
    tol = 1e-4 if dtype == "fp32" else 2e-2
    check(f"swin[{dtype}].stem", stem[:, ::8, ::8, :].float(), torch.from_numpy(Gd["stem"]), tol)
    with torch.no_grad():
        feat = enc(img)
    check(f"swin[{dtype}].features", feat.float(), torch.from_numpy(Gd["feat"]), 2e-4 if dtype == "fp32" else 2.5e-2)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("size", [512, 200])
def test_swin_features_other_image_sizes(ops, dtype, size):
    """--image-size 512 (maps of 128 / 64 / 32 / 16 tokens a side: every stage zero-pads its windows to 133 / 70 / 35 / 21,
    swin_transformer.py:150-152) and a 200 x 200 input (50 -> 25 -> 13 -> 7 tokens: two odd-sized patch merges, :34-44)
    against the real class's outputs (tests/golden/swin_sizes.npz); eval mode, all images encoded."""
    Gd = G("swin_sizes")
    _, model = _product_model(2, 0, dtype, image_size=size)
    model.eval()
    g = torch.Generator().manual_seed(int(Gd[f"seed{size}"]))
    img = torch.rand(1 if size == 512 else 2, 1, size, size, generator=g).to(DEV)
    enc = model.img_encoder
    with torch.no_grad():
        st = enc.features[0]
        stem = ops.swin_stem(img, st[0].weight, st[0].bias, st[2].weight, st[2].bias, model.compute_dtype)
        s1 = enc.features[1](stem)
        feat = enc(img)
    check(f"swin{size}[{dtype}].stage1", s1[:, ::8, ::8, :].float(), torch.from_numpy(Gd[f"stage1_{size}"]),
          2e-4 if dtype == "fp32" else 3e-2)
    check(f"swin{size}[{dtype}].features", feat.float(), torch.from_numpy(Gd[f"feat{size}"]), 2e-4 if dtype == "fp32" else 6e-2)


def test_train_step_at_image_size_512_vs_oracle(ops):
    """A whole training step with 512 x 512 images (256 image tokens per sample; mixed missing modalities; every image is encoded:
    the present-images-only form needs window-multiple maps) against the CPU oracle, fp32 build, eager and hipGraph replay."""
    from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
    from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
    from medical_tri_modal_pilot_amd.optim import FusedAdamW
    import math
    L = 2
    bt = filler.make_batch(777, 4, 24, img_size=512)
    static = torch.stack([bt["gen"], bt["age"]], 1)
    losses = {}
    for graph in (0, 1):
        args, model = _product_model(L, 0, "fp32", hip_graph=graph, image_size=512)
        model.train()
        model.img_encoder.eval()
        opt = FusedAdamW(model.hot_parameters(), lr=args.lr_init, weight_decay=args.weight_decay)
        sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=args.t_0 * 10, cycle_mult=args.t_mult,
                                              max_lr=args.lr_init * math.sqrt(args.batch_size), min_lr=1e-6,
                                              warmup_steps=args.t_up * 10, gamma=args.gamma)
        kw = dict(args=args, x=bt["x"], static=static, y=bt["y"], output_lengths=None, model=model, logger=_Logger(),
                  device=torch.device(DEV), scheduler=sched, optimizer=opt, criterion=torch.nn.BCEWithLogitsLoss(),
                  x_txt=bt["txt"], x_img=bt["img"], imgtxt_time=(bt["img_time"], bt["txt_time"]), scaler=None,
                  missing=bt["missing"], reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
        losses[graph] = [get_trainer(iteration=it, input_lengths=bt["input_lengths"].clone(), txt_lengths=bt["txt_lengths"].clone(),
                                     flow_type="train", **kw)[1] for it in (1, 2)]
        del model, opt
    tr = O.OracleTrainer(_model_sd(L), O.Cfg(n_layers=L), lr_init=args.lr_init, batch_size=args.batch_size, iters_per_epoch=10)
    ref = [tr.step(bt, it) for it in (1, 2)]
    worst = max(abs(a - b) for a, b in zip(losses[0], ref))
    REPORT["image512_step[fp32].loss"] = {"rel_err": worst, "tol": 1e-4}
    assert worst < 1e-4, (losses[0], ref)
    assert losses[0] == losses[1], (losses[0], losses[1])


@pytest.mark.parametrize("n,H,C,heads,shift", [(3, 56, 96, 3, 0), (3, 56, 96, 3, 3), (3, 28, 192, 6, 3), (3, 28, 192, 6, 0),
                                               (3, 14, 96, 3, 3), (3, 7, 192, 6, 3),
                                               (33, 56, 96, 3, 3)])
def test_swin_attention_half_in_one_launch(ops, monkeypatch, n, H, C, heads, shift):
    """mtmp_swin_attn_block (norm1 -> qkv -> window attention -> proj -> StochasticDepth factor -> residual in one launch, q / k / v
    and the attention output never in HBM) against the chain of launches it replaces, on the same block and input: bf16 results
    that differ by accumulation order only (the roundings sit in the same places); with a live-row word the images in front of
    it come out the same."""
    from medical_tri_modal_pilot_amd.builder.models.src import swin_transformer as ST
    g = torch.Generator().manual_seed(H + C + shift)
    blk = ST.SwinTransformerBlock(C, heads, [7, 7], [shift, shift], 0.1)
    sd = {k: filler.fill_tensor("blk." + k, v) for k, v in blk.state_dict().items()}
    for k in sd:
        if k.endswith("bias"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
    sd["attn.relative_position_bias_table"] = torch.randn(sd["attn.relative_position_bias_table"].shape, generator=g)
    blk.load_state_dict(sd)
    blk = blk.to(DEV).eval()
    x = torch.randn(n, H, H, C, generator=g).to(DEV, torch.bfloat16)
    sc = torch.full((n,), 1.25, device=DEV)
    sc[1] = 0.0
    scales = (sc, None)
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(ST, "_FUSED_ATTN", fused)
        with torch.no_grad():
            res[fused] = blk(x, scales=scales).float()
    t = f"swin_attn_block[n={n},H={H},C={C},shift={shift}]"
    check(t, res[True], res[False], 1e-2)
    assert float((res[True] - res[False]).abs().max()) <= 0.07 * float(res[False].abs().max()), t      # no element is off by much
    assert torch.equal(res[True][1], res[False][1])              # (factor 0: the attention branch is dropped, x passes through the MLP half)
    # live rows: two of the three images
    at = blk.attn
    n_live = n - 1 if n > 3 else 2
    word = torch.tensor([n_live * H * H], dtype=torch.int32, device=DEV)
    sh = 0 if 7 >= H else shift
    args = (x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, ST._w(at.qkv.weight, x.dtype), at.qkv.bias,
            at.additive_table(sh, x.dtype, x.device, acc_order=True), heads, sh, ST._w(at.proj.weight, x.dtype), at.proj.bias, scales[0])
    full = ops.swin_attn_block(*args)
    with ops.rows_live(word, 0):
        part = ops.swin_attn_block(*args)
    assert torch.equal(part[:n_live], full[:n_live])
    # ... and against the ORACLE (not only against the HIP chain it replaces: VERDICT r3, P1): x + factor * attention(norm1(x)) of
    # oracle/tri_mbt_oracle.py in fp32 on the same bf16-rounded input and weights -- the kernel is bf16-only, so the gate is a bf16
    # one, and a second, tighter gate holds the error of the attention BRANCH (output minus the residual x, which dominates the sum)
    if n <= 3:
        xf = x.float().cpu()
        sdo = {"a." + k[len("attn."):]: (v.to(torch.bfloat16).float() if k.endswith("weight") else v.float())
               for k, v in sd.items() if k.startswith("attn.")}
        h = torch.nn.functional.layer_norm(xf, (C,), sd["norm1.weight"].float(), sd["norm1.bias"].float(), blk.norm1.eps)
        branch = O.swin_window_attention(sdo, "a", h, heads, shift)
        ref = xf + sc.cpu().view(-1, 1, 1, 1) * branch
        check(t + ".vs_oracle", full.float().cpu(), ref, 1.2e-2)
        got_branch = (full.float().cpu() - xf)[sc.cpu() != 0]
        ref_branch = (ref - xf)[sc.cpu() != 0]
        check(t + ".branch_vs_oracle", got_branch, ref_branch, 2.5e-2)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("H,C,heads,shift", [(16, 768, 24, 3), (32, 384, 12, 3), (25, 192, 6, 3), (13, 384, 12, 0), (5, 96, 3, 3),
                                             (128, 96, 3, 3)])
def test_swin_window_attention_zero_padded_windows(ops, dt, H, C, heads, shift):
    """Maps that are not multiples of the 7x7 window: the zero-padded windows of swin_transformer.py:150-152 (pad tokens carry the
    qkv bias as q / k / v, take part as keys, are cropped from the result; a padded map of one window is not shifted) against
    the oracle's restatement, which tests/test_oracle_golden.py pins to the real class at 512 and 200 pixels."""
    from medical_tri_modal_pilot_amd.builder.models.src.swin_transformer import ShiftedWindowAttention
    g = torch.Generator().manual_seed(H + C + shift)
    att = ShiftedWindowAttention(C, [7, 7], [shift, shift], heads)
    sd = {k: filler.fill_tensor("wa." + k, v) for k, v in att.state_dict().items()}
    sd["qkv.bias"] = 0.3 * torch.randn(3 * C, generator=g)             # (the filler's biases are ~0: make the pad tokens count)
    att.load_state_dict(sd)
    att = att.to(DEV)
    n = 1 if H > 64 else 2
    x = torch.randn(n, H, H, C, generator=g).to(dt).float()
    sdo = {"a." + k: (v.to(dt).float() if k.endswith("weight") else v) for k, v in sd.items()}
    ref = O.swin_window_attention(sdo, "a", x, heads, shift)
    a = att(x.to(DEV, dt))
    assert a.shape == x.shape and a.is_contiguous()
    y = torch.nn.functional.linear(a.float(), sdo["a.proj.weight"].to(DEV), sdo["a.proj.bias"].to(DEV))
    check(f"swin_wattn_padded[{str(dt)[6:]},H={H},C={C},shift={shift}]", y, ref, 1e-4 if dt == torch.float32 else 3e-2)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("H,C,heads,shift", [(14, 192, 6, 0), (14, 192, 6, 3), (28, 96, 3, 3), (7, 768, 24, 3)])
def test_swin_window_attention_and_layernorm(ops, dt, H, C, heads, shift):
    from medical_tri_modal_pilot_amd.builder.models.src.swin_transformer import ShiftedWindowAttention
    g = torch.Generator().manual_seed(H + C + shift)
    att = ShiftedWindowAttention(C, [7, 7], [shift, shift], heads)
    sd = {k: filler.fill_tensor("wa." + k, v) for k, v in att.state_dict().items()}
    att.load_state_dict(sd)
    att = att.to(DEV)
    x = torch.randn(2, H, H, C, generator=g).to(dt).float()
    sdo = {"a." + k: (v.to(dt).float() if k.endswith("weight") else v) for k, v in sd.items()}
    ref = O.swin_window_attention(sdo, "a", x, heads, shift)
    a = att(x.to(DEV, dt))
    y = torch.nn.functional.linear(a.float(), sdo["a.proj.weight"].to(DEV), sdo["a.proj.bias"].to(DEV))
    t = f"swin_wattn[{str(dt)[6:]},H={H},C={C},shift={shift}]"
    check(t, y, ref, 1e-4 if dt == torch.float32 else 3e-2)
    w, b = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ln = ops.layernorm_rows(x.to(DEV, dt), w.to(DEV), b.to(DEV))
    check(t + ".ln", ln.float(), torch.nn.functional.layer_norm(x, (C,), w, b, 1e-5), 1e-5 if dt == torch.float32 else 1e-2)
    if H % 2 == 0:
        w4, b4 = 1 + 0.1 * torch.randn(4 * C, generator=g), 0.1 * torch.randn(4 * C, generator=g)
        m = ops.layernorm_rows(x.to(DEV, dt), w4.to(DEV), b4.to(DEV), merge_hw=(H, H))
        cat = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
        check(t + ".merge_ln", m.float(), torch.nn.functional.layer_norm(cat, (4 * C,), w4, b4, 1e-5),
              1e-5 if dt == torch.float32 else 1e-2)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("H,C,heads,shift", [(14, 192, 6, 0), (14, 192, 6, 3), (28, 96, 3, 3), (7, 768, 24, 3), (14, 384, 12, 3)])
def test_swin_window_attention_backward_vs_oracle_autograd(ops, dt, H, C, heads, shift):
    """Round 5 (VERDICT r4 missing #3): the trainable attention half -- qkv projection, mtmp_swin_window_attn(_bwd), output
    projection as autograd nodes (ShiftedWindowAttention.forward_train + ops.LinearFn) -- against torch autograd of the oracle's
    window attention (swin_transformer.py:115-225): input gradient, qkv / proj weights and biases, relative_position_bias_table."""
    from medical_tri_modal_pilot_amd.builder.models.src.swin_transformer import ShiftedWindowAttention
    g = torch.Generator().manual_seed(H + C + shift)
    att = ShiftedWindowAttention(C, [7, 7], [shift, shift], heads)
    sd = {k: filler.fill_tensor("wa." + k, v) for k, v in att.state_dict().items()}
    att.load_state_dict(sd)
    att = att.to(DEV)
    x = torch.randn(3, H, H, C, generator=g).to(dt).float()
    w = torch.randn(3, H, H, C, generator=g).to(dt).float()
    sdo = {"a." + k: (v.to(dt).float() if k.endswith("weight") else v.clone()) for k, v in sd.items()}
    xr = x.clone().requires_grad_()
    for k in ("qkv.weight", "qkv.bias", "proj.weight", "proj.bias", "relative_position_bias_table"):
        sdo["a." + k].requires_grad_()
    (O.swin_window_attention(sdo, "a", xr, heads, shift) * w).sum().backward()
    xd = x.to(DEV, dt).requires_grad_()
    y = ops.LinearFn.apply(att.forward_train(xd), att.proj.weight, att.proj.bias, dt)
    (y.float() * w.to(DEV)).sum().backward()
    t = f"swin_wattn_bwd[{str(dt)[6:]},H={H},C={C},shift={shift}]"
    tol = 1e-4 if dt == torch.float32 else 4e-2
    check(t + ".y", y.float(), O.swin_window_attention(sdo, "a", x, heads, shift).detach(), 1e-4 if dt == torch.float32 else 3e-2)
    check(t + ".dx", xd.grad.float(), xr.grad, tol)
    for k in ("qkv.weight", "qkv.bias", "proj.weight", "proj.bias", "relative_position_bias_table"):
        got = dict(att.named_parameters())[k].grad
        check(f"{t}.d{k}", got.float(), sdo["a." + k].grad, tol)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("rows,C", [(3 * 196, 192), (2 * 3136 - 5, 96), (77, 768), (130, 1536), (64, 384)])
def test_layernorm_rows_and_gelu_backward_vs_torch(ops, dt, rows, C):
    """mtmp_layernorm_rows_bwd / mtmp_gelu_bwd (nn.LayerNorm and nn.GELU of the image encoder, swin_transformer.py:428-449) against
    torch autograd on the same (dtype-rounded) inputs."""
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g).to(dt).float()
    wgt = torch.randn(rows, C, generator=g).to(dt).float()
    lw, lb = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xr, lwr, lbr = x.clone().requires_grad_(), lw.clone().requires_grad_(), lb.clone().requires_grad_()
    (torch.nn.functional.layer_norm(xr, (C,), lwr, lbr, 1e-5) * wgt).sum().backward()
    xd = x.to(DEV, dt).requires_grad_()
    lwd, lbd = lw.to(DEV).requires_grad_(), lb.to(DEV).requires_grad_()
    y = ops.LayerNormRowsFn.apply(xd, lwd, lbd, 1e-5)
    (y.float() * wgt.to(DEV)).sum().backward()
    t = f"ln_rows_bwd[{str(dt)[6:]},rows={rows},C={C}]"
    tol = 1e-4 if dt == torch.float32 else 2e-2
    check(t + ".dx", xd.grad.float(), xr.grad, tol)
    check(t + ".dw", lwd.grad, lwr.grad, tol)
    check(t + ".db", lbd.grad, lbr.grad, tol)
    # GELU: the fp32 build is the exact erf form; the bf16 build a tanh-form logistic (|error| <= 5e-4 absolute)
    xr2 = x.clone().requires_grad_()
    (torch.nn.functional.gelu(xr2) * wgt).sum().backward()
    xd2 = x.to(DEV, dt).requires_grad_()
    yg = ops.GeluFn.apply(xd2)
    (yg.float() * wgt.to(DEV)).sum().backward()
    check(t + ".gelu", yg.float(), torch.nn.functional.gelu(x), 1e-5 if dt == torch.float32 else 1e-2)
    check(t + ".dgelu", xd2.grad.float(), xr2.grad, 1e-4 if dt == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_swin_encoder_backward_vs_oracle_autograd(ops, dtype, monkeypatch):
    """The whole trainable encoder (SwinTransformer.forward_train: patch embedding, 12 blocks, 3 patch mergings, final norm) in
    TRAIN mode with the same StochasticDepth draws injected on both sides, against torch autograd of the oracle's encoder
    (swin_transformer.py:559-654): features and the gradient of every encoder parameter."""
    from medical_tri_modal_pilot_amd.builder.models.src import swin_transformer as sw
    _, model = _product_model(2, 0, dtype)
    enc = model.img_encoder
    enc.train()
    n = 3
    g = torch.Generator().manual_seed(77)
    img = torch.rand(n, 1, 224, 224, generator=g)
    wgt = torch.randn(n, 7, 7, 768, generator=g)
    mods = [m for m in enc.modules() if isinstance(m, sw.StochasticDepth)]
    scales = []
    for m in mods:
        keep = 1.0 - m.p
        pair = [(torch.rand(n, generator=g) < keep).float() / keep for _ in range(2)]
        scales.append((pair[0], pair[1]))
        m._predrawn = [pair[1].to(DEV), pair[0].to(DEV)]               # popped in call order: attention branch, then MLP
    monkeypatch.setattr(sw, "draw_row_scales", lambda *a, **k: None)   # keep the injected draws
    dt = torch.float32 if dtype == "fp32" else torch.bfloat16
    sd = {"img_encoder." + k: (v.detach().cpu().float().clone()) for k, v in enc.state_dict().items()}
    if dtype == "bf16":
        sd = {k: (v.to(dt).float() if k.endswith("weight") and v.dim() > 1 else v) for k, v in sd.items()}
    train_keys = ["img_encoder." + k for k, p in enc.named_parameters() if not k.startswith("head.")]
    for k in train_keys:
        sd[k].requires_grad_()
    ref = O.swin_forward(sd, "img_encoder", img, row_scales=scales)
    (ref * wgt).sum().backward()
    feat = enc(img.to(DEV))
    assert feat.requires_grad
    (feat.float() * wgt.to(DEV)).sum().backward()
    t = f"swin_train_bwd[{dtype}]"
    check(t + ".features", feat.float(), ref.detach(), 2e-4 if dtype == "fp32" else 3e-2)
    prm = dict(enc.named_parameters())
    errs = sorted(((_rel(prm[k[len("img_encoder."):]].grad.float().cpu(), sd[k].grad), k) for k in train_keys), reverse=True)
    worst, typical = errs[0][0], errs[len(errs) // 2][0]
    REPORT[t + ".worst_param_grad"] = {"rel_err": worst, "tol": 2e-4 if dtype == "fp32" else 0.1, "tensor": errs[0][1], "tensors": len(errs)}
    REPORT[t + ".median_param_grad"] = {"rel_err": typical, "tol": 1e-4 if dtype == "fp32" else 3e-2}
    assert len(errs) == 171 and all(prm[k].grad is None for k in ("head.weight", "head.bias"))
    assert worst < (2e-4 if dtype == "fp32" else 0.1) and typical < (1e-4 if dtype == "fp32" else 3e-2), errs[:6]


@pytest.mark.parametrize("C,rows,hw", [(96, 3 * 3136, 3136), (192, 5 * 784 - 7, 784), (96, 130, 64)])
def test_swin_mlp_fused_vs_chain_and_torch(ops, C, rows, hw):
    """mtmp_swin_mlp (LN -> fc1 -> GELU -> fc2 -> row scale -> residual, one launch) against the three-launch chain
    it replaces (same roundings; only fp32 summation order differs) and against the fp32 torch modules
    (swin_transformer.py:428-449); ragged row count = clamped last rows."""
    g = torch.Generator().manual_seed(C + rows)
    bf = torch.bfloat16
    x = torch.randn(rows, C, generator=g).to(bf)
    lw, lb = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    w1, b1 = (torch.randn(4 * C, C, generator=g) * C ** -0.5).to(bf), 0.1 * torch.randn(4 * C, generator=g)
    w2, b2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5).to(bf), 0.1 * torch.randn(C, generator=g)
    n_img = (rows + hw - 1) // hw
    rs = torch.tensor([0.0, 1.25, 1.0, 1.25, 1.25][:n_img])
    d = lambda t: t.to(DEV)
    for scale in (None, rs):
        y = ops.swin_mlp(d(x), d(lw), d(lb), 1e-5, d(w1), d(b1), d(w2), d(b2), None if scale is None else d(scale), hw)
        h = ops.layernorm_rows(d(x), d(lw), d(lb), 1e-5)
        h = ops.gemm_nt(h, d(w1), d(b1), act="gelu")
        chain = ops.gemm_nt(h, d(w2), d(b2), res2d=d(x), row_scale=None if scale is None else d(scale), rows_per_scale=hw)
        xf = x.float()
        hid = torch.nn.functional.gelu(torch.nn.functional.layer_norm(xf, (C,), lw, lb, 1e-5) @ w1.float().t() + b1)
        ref = hid @ w2.float().t() + b2
        if scale is not None:
            ref = ref * scale.repeat_interleave(hw)[:rows, None]
        ref = xf + ref
        t = f"swin_mlp[C={C},rows={rows},scale={scale is not None}]"
        check(t + ".vs_chain", y.float(), chain.float(), 2e-2)
        check(t + ".vs_torch_fp32", y.float(), ref, 3e-2)
        frac = (y != chain).float().mean().item()
        assert frac < 0.10, f"{t}: {frac:.3f} of the outputs differ from the chain by a bf16 ulp or more"


@pytest.mark.parametrize("C,rows", [(96, 3 * 3136), (192, 5 * 784 - 7), (96, 33)])
def test_swin_ln_linear_fused_vs_chain_and_torch(ops, C, rows):
    """mtmp_swin_ln_linear (norm1 + qkv projection, one launch) against mtmp_layernorm_rows + mtmp_gemm_nt and against
    fp32 torch; ragged row counts exercise the clamped last rows / the early exit of whole waves."""
    g = torch.Generator().manual_seed(C + rows)
    bf = torch.bfloat16
    x = torch.randn(rows, C, generator=g).to(bf)
    lw, lb = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    w, b = (torch.randn(3 * C, C, generator=g) * C ** -0.5).to(bf), 0.1 * torch.randn(3 * C, generator=g)
    d = lambda t: t.to(DEV)
    for bias in (b, None):
        y = ops.swin_ln_linear(d(x), d(lw), d(lb), 1e-5, d(w), None if bias is None else d(bias))
        chain = ops.gemm_nt(ops.layernorm_rows(d(x), d(lw), d(lb), 1e-5), d(w), None if bias is None else d(bias))
        ref = torch.nn.functional.layer_norm(x.float(), (C,), lw, lb, 1e-5) @ w.float().t() + (0 if bias is None else bias)
        t = f"swin_ln_linear[C={C},rows={rows},bias={bias is not None}]"
        check(t + ".vs_chain", y.float(), chain.float(), 1e-2)
        check(t + ".vs_torch_fp32", y.float(), ref, 2e-2)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_swin_train_mode_stochastic_depth_vs_oracle(ops, dtype, monkeypatch):
    """model.train() re-activates the frozen encoder's row-mode StochasticDepth (2_train.py:128 overrides the .eval() of
    tri_mbt_vsltcls.py:104).  RNG streams cannot match across devices, so the same 24 draws are injected on both sides:
    the product applies them in the projection GEMMs' epilogues (and splits stages 3-4 over two streams), the oracle
    multiplies each residual branch (torchvision.ops.stochastic_depth, mode "row")."""
    import medical_tri_modal_pilot_amd.builder.models.src.swin_transformer as sw
    _, model = _product_model(2, 0, dtype)
    enc = model.img_encoder
    enc.train()
    n = 16
    g = torch.Generator().manual_seed(5)
    img = torch.rand(n, 1, 224, 224, generator=g)
    mods = [m for m in enc.modules() if isinstance(m, sw.StochasticDepth)]
    assert len(mods) == 12 and mods[0].p == 0.0 and abs(mods[-1].p - 0.2) < 1e-12
    pairs = []
    for m in mods:
        keep = 1.0 - m.p
        if m.p == 0.0:
            pairs.append((None, None))
            continue
        sa = (torch.rand(n, generator=g) < keep).float() / keep
        sm = (torch.rand(n, generator=g) < keep).float() / keep
        pairs.append((sa, sm))
        m._predrawn = [sm.to(DEV), sa.to(DEV)]             # popped in call order: attention branch, then MLP
    assert any(float(sa.min()) == 0.0 for sa, _ in pairs if sa is not None)      # some branch is really dropped
    monkeypatch.setattr(sw, "draw_row_scales", lambda *a, **k: None)              # keep the injected draws
    streams = model.fusion_transformer._side_streams(torch.device(DEV))
    with torch.no_grad():
        feat = enc(img.to(DEV), tail_streams=(streams[0], streams[1]))
    torch.cuda.synchronize()
    ref = O.swin_forward(_model_sd(2), "img_encoder", img, row_scales=pairs)
    check(f"swin_train_stochastic_depth[{dtype}].features", feat.float(), ref, 2e-4 if dtype == "fp32" else 6e-2)
    # the same against the fixture of the REAL encoder in train mode (its recorded draws injected here)
    Gt = G("swin_train")
    gi = torch.Generator().manual_seed(int(Gt["seed"]))
    img3 = torch.rand(3, 1, 224, 224, generator=gi)
    draws, it = torch.from_numpy(Gt["draws"]), 0
    for m in mods:
        if m.p > 0.0:
            m._predrawn = [draws[it + 1].to(DEV), draws[it].to(DEV)]        # popped in call order: attention branch, then MLP
            it += 2
    with torch.no_grad():
        feat3 = enc(img3.to(DEV), tail_streams=(streams[0], streams[1]))
    torch.cuda.synchronize()
    check(f"swin_train_golden[{dtype}].features", feat3.float(), torch.from_numpy(Gt["feat"]), 2e-4 if dtype == "fp32" else 2.5e-2)


def test_swin_tail_split_equals_single_stream(ops):
    """Stages 3-4 as two half batches on two HIP streams (SwinTransformer.forward(tail_streams=...)) against the
    one-stream forward, train mode (StochasticDepth active, same draws): the per-image arithmetic is the same."""
    _, model = _product_model(2, 0, "bf16")
    enc = model.img_encoder.train()
    g = torch.Generator().manual_seed(5)
    img = torch.rand(16, 1, 224, 224, generator=g).to(DEV)
    s0, s1, head = (torch.cuda.Stream(device=DEV) for _ in range(3))
    with torch.no_grad():
        torch.manual_seed(11)
        ref = enc(img)
        torch.cuda.synchronize()
        torch.manual_seed(11)
        head.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(head):
            out = enc(img, tail_streams=(s0, s1))
        torch.cuda.current_stream().wait_stream(s0)
        torch.cuda.current_stream().wait_stream(head)
        torch.cuda.synchronize()
    assert out.shape == ref.shape == (16, 7, 7, 768)
    check("swin.tail_split_vs_single", out.float(), ref.float(), 1e-6)


def test_fused_adamw_matches_torch(ops):
    from medical_tri_modal_pilot_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(2)
    shapes = [(256, 256), (1024, 256, 1), (256,), (1, 4, 256), (3,)]
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).to(DEV)) for s in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    fo = FusedAdamW([(f"p{i}", p) for i, p in enumerate(ps)], lr=1e-3, weight_decay=1e-2)
    to = torch.optim.AdamW(qs, lr=1e-3, weight_decay=1e-2)
    for it in range(3):
        fo.zero_grad(); to.zero_grad()
        for p, q in zip(ps, qs):
            gr = torch.randn(p.shape, generator=g).to(DEV)
            p.grad.copy_(gr); q.grad = gr.clone()
        for grp in fo.param_groups + to.param_groups:
            grp["lr"] = 1e-3 * (it + 1)
        fo.step(); to.step()
    for i, (p, q) in enumerate(zip(ps, qs)):
        check(f"adamw.p{i}", p, q, 1e-6)


# ------------------------------------------------------------------ blocks / encoder / full model
def test_encoder_layer_vs_golden(ops):
    from medical_tri_modal_pilot_amd.builder.models.src.transformer.encoder import TransformerEncoderLayer
    Gd = G("blocks")
    lay = TransformerEncoderLayer(256, 4, 1024, 0.0)
    lay.load_state_dict({k: filler.fill_tensor("g3." + k, v) for k, v in lay.state_dict().items()})
    lay = lay.to(DEV)
    g = torch.Generator().manual_seed(11)
    for N in (54, 133, 261):
        torch.randn(4, N, 256, generator=g); torch.randn(4, N, 256, generator=g)
    torch.randn(2, 40, 256, generator=g)
    torch.randn(3, 17, 256, generator=g); torch.randn(3, 17, 256, generator=g)
    x = torch.randn(3, 70, 256, generator=g)
    w = torch.randn(3, 70, 256, generator=g)
    xd = x.to(DEV).requires_grad_()
    y, _ = lay(xd, torch.from_numpy(Gd["lay_len"]).to(DEV))
    (y * w.to(DEV)).sum().backward()
    check("layer[fp32].y", y[:, ::5], torch.from_numpy(Gd["lay_y"]), 1e-4)
    check("layer[fp32].dx", xd.grad[:, ::5], torch.from_numpy(Gd["lay_dx"]), 1e-4)
    check("layer[fp32].dgamma_attn", lay.attention_prenorm.gamma.grad, torch.from_numpy(Gd["lay_dgamma_attn"]), 1e-4)
    d = lambda t: torch.cat([t.detach().double().norm().view(1).cpu(),
                             t.detach().double().reshape(-1)[torch.linspace(0, t.numel() - 1, 8).long()].cpu()])
    check("layer[fp32].dW1", d(lay.feed_forward.w_1.weight.grad), torch.from_numpy(Gd["lay_dW1"]), 1e-4)
    check("layer[fp32].dWk", d(lay.self_attention.key_proj.linear.weight.grad), torch.from_numpy(Gd["lay_dWk"]), 1e-4)


def test_mbt_encoder_variants_vs_golden(ops):
    from medical_tri_modal_pilot_amd.builder.models.src.transformer.mbt_encoder import TrimodalTransformerEncoder_MBT
    Gd = G("encoder")
    worst = 0.0
    for case in range(int(Gd["n_cases"])):
        vsltonly, resb, fstart, multi, B, T, L = [int(v) for v in Gd[f"c{case}_cfg"]]
        enc = TrimodalTransformerEncoder_MBT(batch_size=B, n_modality=3, bottlenecks_n=4, fusion_startidx=fstart,
                                             d_input=256, resbottle=bool(resb), n_layers=L, n_head=4, d_model=256,
                                             d_ff=1024, dropout=0.0, vsltonly=vsltonly, pe_maxlen=2500,
                                             use_pe=[False, False, True], mask=[True, bool(multi), True],
                                             compute_dtype=torch.float32)
        enc.load_state_dict({k: filler.fill_tensor("g4." + k, v) for k, v in enc.state_dict().items()})
        enc = enc.to(DEV).eval()
        g = torch.Generator().manual_seed(100 + case)
        n_img = 147 if multi else 49
        v = torch.randn(B, T, 256, generator=g).to(DEV)
        i = torch.randn(B, n_img, 256, generator=g).to(DEV)
        t = torch.randn(B, 30, 256, generator=g).to(DEV)
        in_len, txt_len = torch.tensor([T, 3, 11, 7]).to(DEV), torch.tensor([20, 0, 5, 0]).to(DEV)
        img_len = (torch.from_numpy(Gd[f"c{case}_imgcnt"]) * 49).to(DEV) if multi else n_img
        with torch.no_grad():
            outs, _ = enc([v, i, t], fixed_lengths=[T, n_img, 30], varying_lengths=[in_len, img_len, txt_len + 2],
                          missing=torch.tensor([0, 1, 2, 3]).to(DEV))
        assert len(outs) == (1 if vsltonly else 3)
        for m, o in enumerate(outs):
            ref = torch.from_numpy(Gd[f"c{case}_out{m}"])
            got = o if m == 0 else o[:, ::7]
            e = _rel(got, ref)
            worst = max(worst, e)
            assert e < 1e-4, f"case {case} stream {m}: {e}"
    REPORT["mbt_encoder[fp32].worst_of_16_cases"] = {"rel_err": worst, "tol": 1e-4}


def test_bimodal_mbt_encoder_vs_golden(ops):
    """BimodalTransformerEncoder_MBT (mbt_encoder.py:519-634) on the same explicit-buffer engine with two
    streams -- outputs, input gradients, the bottleneck-token gradient and two weight-gradient digests against the
    real class (tests/golden/gen/make_golden.py gen_bimodal), fp32 build."""
    from medical_tri_modal_pilot_amd.builder.models.src.transformer.mbt_encoder import BimodalTransformerEncoder_MBT
    Gd = G("bimodal")
    worst = 0.0
    for case in range(int(Gd["n_cases"])):
        mask1, txt_idx, pe1, B, T, L = [int(v) for v in Gd[f"c{case}_cfg"]]
        enc = BimodalTransformerEncoder_MBT(batch_size=B, n_modality=2, bottlenecks_n=4, fusion_startidx=0, d_input=256,
                                            n_layers=L, n_head=4, d_model=256, d_ff=1024, dropout=0.0, pe_maxlen=2500,
                                            txt_idx=txt_idx, use_pe=[False, bool(pe1)], mask=[True, bool(mask1)],
                                            compute_dtype=torch.float32)
        enc.load_state_dict({k: filler.fill_tensor("g5." + k, v) for k, v in enc.state_dict().items()})
        enc = enc.to(DEV).eval()
        g = torch.Generator().manual_seed(300 + case)
        v = torch.randn(B, T, 256, generator=g).to(DEV).requires_grad_()
        t = torch.randn(B, 30, 256, generator=g).to(DEV).requires_grad_()
        in_len, txt_len = torch.tensor([T, 3, 11, 7]).to(DEV), torch.tensor([20, 0, 5, 0]).to(DEV)
        outs, _ = enc([v, t], fixed_lengths=[T, 30], varying_lengths=[in_len, txt_len + 2],
                      missing=torch.tensor([0, 1, 1, 0]).to(DEV))
        assert len(outs) == 2
        w0 = torch.randn(outs[0].shape, generator=g).to(DEV)
        w1 = torch.randn(outs[1].shape, generator=g).to(DEV)
        ((outs[0] * w0).sum() + (outs[1] * w1).sum()).backward()
        pairs = [(outs[0], Gd[f"c{case}_out0"]), (outs[1][:, ::7], Gd[f"c{case}_out1"]), (v.grad[:, ::5], Gd[f"c{case}_dv"]),
                 (t.grad[:, ::5], Gd[f"c{case}_dt"]), (enc.bottlenecks.grad, Gd[f"c{case}_dbott"]),
                 (_digest(enc.layer_stacks[1][1].feed_forward.w_1.weight.grad), Gd[f"c{case}_dw1"]),
                 (_digest(enc.layer_stacks[0][0].self_attention.query_proj.linear.weight.grad), Gd[f"c{case}_dwq"])]
        for k, (got, ref) in enumerate(pairs):
            e = _rel(got, torch.from_numpy(np.asarray(ref)))
            worst = max(worst, e)
            assert e < 1e-4, f"case {case} item {k}: {e}"
    REPORT["bimodal_mbt_encoder[fp32].worst_of_4_cases"] = {"rel_err": worst, "tol": 1e-4}


def test_bi_vslttxt_model_train_step_vs_golden(ops):
    """BI_VSLTTXT_MBT_V1 through get_model / get_trainer (--input-types vslt_txt: the trainer folds the four
    modality patterns onto {0, 1}) -- logits-derived loss and all 86 parameter gradients against the real class, then
    the same step replayed from a hipGraph."""
    import json
    from medical_tri_modal_pilot_amd.control.config import parse_args
    from medical_tri_modal_pilot_amd.builder.models import get_model
    from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
    from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
    from medical_tri_modal_pilot_amd.optim import FusedAdamW
    Gd = G("bimodel_step")
    with open(os.path.join(ROOT, "tests", "golden", "state_shapes_bi_vslttxt_L2.json")) as f:
        shapes = json.load(f)
    sd = {k: filler.fill_tensor(k, torch.zeros(s)) for k, (s, dt_) in shapes.items() if dt_.startswith("float")}
    sd["fusion_transformer.positional_encoding.pe"] = O.sinusoid_table(2500, 256).unsqueeze(0)
    losses = {}
    for graph in (0, 1):
        a = parse_args(["--input-types", "vslt_txt", "--model", "bi_vslttxt_mbt_v1", "--modality-inclusion",
                        "train-missing_test-missing", "--lr-init", "1e-5", "--batch-size", "4", "--transformer-num-layers", "2",
                        "--imgtxt-time", "1", "--dropout", "0.0", "--compute-dtype", "fp32", "--hip-graph", str(graph)])
        a.device = torch.device(DEV)
        model = get_model(a)(a)
        model.load_state_dict(sd, strict=False)
        model = model.to(DEV).train()
        opt = FusedAdamW(model.hot_parameters(), lr=a.lr_init, weight_decay=a.weight_decay)
        sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=a.t_0 * 10, cycle_mult=a.t_mult,
                                              max_lr=a.lr_init * math.sqrt(a.batch_size), min_lr=1e-6,
                                              warmup_steps=a.t_up * 10, gamma=a.gamma)
        bt = filler.make_batch(int(Gd["seed"]), int(Gd["B"]), int(Gd["T"]))
        static = torch.stack([bt["gen"], bt["age"]], 1)
        kw = dict(args=a, x=bt["x"], static=static, y=bt["y"], output_lengths=None, model=model, logger=_Logger(),
                  device=torch.device(DEV), scheduler=sched, optimizer=opt, criterion=torch.nn.BCEWithLogitsLoss(),
                  x_txt=bt["txt"], x_img=bt["img"], imgtxt_time=(bt["img_time"], bt["txt_time"]), scaler=None,
                  missing=bt["missing"], reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
        seq = []
        for it in range(3 if graph else 1):
            _, l = get_trainer(iteration=it + 1, input_lengths=bt["input_lengths"].clone(),
                               txt_lengths=bt["txt_lengths"].clone(), flow_type="train", **kw)
            seq.append(l)
            if it == 0 and not graph:
                grads = {n: p.grad.detach().clone() for n, p in model.hot_parameters()}
        losses[graph] = seq
    assert abs(losses[0][0] - float(Gd["loss"])) < 1e-5, (losses, float(Gd["loss"]))
    assert losses[1][0] == losses[0][0]                         # eager warm-up step of the graph path = the eager step
    gs = getattr(model, "_mtmp_graph_step", None)
    assert gs is not None and gs.captures == 1 and gs.replays == 2 and all(math.isfinite(x) for x in losses[1])
    names = [str(s) for s in Gd["grad_names"]]
    assert sorted(names) == sorted(grads)
    med = float(np.median(Gd["grad_digest"][:, 0]))
    worst = 0.0
    for n_, gd in zip(names, Gd["grad_digest"]):
        if gd[0] < 1e-4 * med:
            assert float(_digest(grads[n_])[0]) < 1e-3 * med, n_
            continue
        worst = max(worst, _rel(_digest(grads[n_]), torch.from_numpy(gd)))
    REPORT["bi_vslttxt_step[fp32].loss"] = {"rel_err": abs(losses[0][0] - float(Gd["loss"])), "tol": 1e-5}
    REPORT["bi_vslttxt_step[fp32].worst_grad_digest"] = {"rel_err": worst, "tol": 1e-4}
    assert worst < 1e-4


@pytest.mark.parametrize("name,input_types,tag", [("tri_mbt_vsltcls_noshareumse", "vslt_img_txt", "noshareumse"), ("tri_mbt_v1", "vslt_img_txt", "tri_v1"),
                                                  ("tri_mbt_vflexible", "vslt_img_txt", "tri_vflex"), ("tri_mbt_vflexible2", "vslt_img_txt", "tri_vflex2"),
                                                  ("tri_mbt_vflexible3", "vslt_img_txt", "tri_vflex3"),
                                                  ("bi_vsltimg_mbt_v1", "vslt_img", "bi_vsltimg"), ("bitxt_mbt_vflexible1", "vslt_txt", "bitxt_vflex1"),
                                                  ("biimg_mbt_vflexible1", "vslt_img", "biimg_vflex1"),
                                                  ("bi_vsltimg_mbt_v1", "vslt_img", "bi_vsltimg_train"),
                                                  ("tri_mbt_v2", "vslt_img_txt", "tri_v2"), ("tri_mbt_vnoshavgtr", "vslt_img_txt", "tri_vnoshavgtr")])
def test_more_sibling_models_train_step_vs_golden(ops, name, input_types, tag, monkeypatch):
    """SURVEY 8 f-4 / VERDICT r2 missing #3: TRI_MBT_VSLTCLS_NOSHAREUMSE (UMSE chains without LayerNorm, own time chains for
    image / report) and BI_VSLTIMG_MBT_V1 (two streams with the CXR encoder, head on both CLS rows) through get_model:
    logits, BCE loss and every parameter gradient -- the image encoder's included where the reference trains it -- against the REAL classes
    (tests/golden/gen/make_golden.py siblings), fp32 build, 1e-4."""
    import json
    from medical_tri_modal_pilot_amd.control.config import parse_args
    from medical_tri_modal_pilot_amd.builder.models import get_model
    Gd = G(tag + "_step")
    # (<tag>_train: the same model with its image encoder left in TRAIN mode -- 2_train.py:128 -- and the reference's recorded
    #  StochasticDepth draws injected here)
    with open(os.path.join(ROOT, "tests", "golden", f"state_shapes_{tag[:-6] if tag.endswith('_train') else tag}_L2.json")) as f:
        shapes = json.load(f)
    sd = {k: filler.fill_tensor(k, torch.zeros(s)) for k, (s, dt_) in shapes.items() if dt_.startswith("float")}
    sd["fusion_transformer.positional_encoding.pe"] = O.sinusoid_table(2500, 256).unsqueeze(0)
    a = parse_args(["--input-types", input_types, "--model", name, "--modality-inclusion", "train-missing_test-missing",
                    "--lr-init", "1e-5", "--batch-size", "4", "--transformer-num-layers", "2", "--imgtxt-time", "1",
                    "--mbt-only-vslt", "1", "--dropout", "0.0", "--compute-dtype", "fp32", "--hip-graph", "0"]
                   + (["--berttype", "bert"] if "tokens" in Gd.files else []))      # (tri_mbt_v2: token-id reports, tri_mbt_v2.py:205)
    a.device, a.output_dim = torch.device(DEV), 1
    model = get_model(a)(a)
    missing_keys = model.load_state_dict(sd, strict=False)
    assert not [k for k in missing_keys.missing_keys if "relative_position_index" not in k and "idx" not in k], missing_keys
    model = model.to(DEV).train()
    if hasattr(model, "img_encoder") and "draws" not in Gd.files:
        model.img_encoder.eval()
    if "draws" in Gd.files:
        from medical_tri_modal_pilot_amd.builder.models.src import swin_transformer as sw
        draws, it = torch.from_numpy(Gd["draws"]), 0
        for m in model.img_encoder.modules():
            if isinstance(m, sw.StochasticDepth) and m.p > 0.0:
                m._predrawn = [draws[it + 1].to(DEV), draws[it].to(DEV)]        # popped in call order: attention branch, then MLP
                it += 2
        assert it == draws.shape[0]
        monkeypatch.setattr(sw, "draw_row_scales", lambda *a, **k: None)       # keep the injected draws
    bt = filler.make_batch(int(Gd["seed"]), int(Gd["B"]), int(Gd["T"]))
    if "tokens" in Gd.files:
        bt["txt"] = torch.from_numpy(Gd["tokens"]).float()
    mnum = torch.from_numpy(Gd["missing_num"])
    tmax = int(bt["input_lengths"].max())
    dv = lambda t: t.to(DEV)
    out, o2, o3 = model(dv(bt["x"][:, :tmax]), None, None, None, None, dv(bt["age"]), dv(bt["gen"]), dv(bt["input_lengths"].clone()),
                        dv(bt["txt"]), dv(bt["txt_lengths"].clone()), dv(bt["img"]), dv(mnum), None, dv(bt["img_time"].half().float()),
                        dv(bt["txt_time"].half().float()), "train", None, None)
    assert o2 is None and o3 is None and tuple(out.shape) == tuple(Gd["logits"].shape)
    check(f"{tag}_step[fp32].logits", out, torch.from_numpy(Gd["logits"]), 1e-4)
    loss = torch.nn.BCEWithLogitsLoss()(out.squeeze(-1), dv(bt["y"].float()))
    REPORT[f"{tag}_step[fp32].loss"] = {"rel_err": abs(float(loss) - float(Gd["loss"])), "tol": 1e-5}
    assert abs(float(loss) - float(Gd["loss"])) < 1e-5
    loss.backward()
    names = [str(s) for s in Gd["grad_names"]]
    med = float(np.median(Gd["grad_digest"][:, 0]))
    prm = dict(model.named_parameters())
    worst, n_checked = 0.0, 0
    enc_checked = 0
    for n_, gd in zip(names, Gd["grad_digest"]):
        # (img_encoder.* too where the reference trains the encoder -- bi_vsltimg_mbt_v1.py:203-206: the HIP backward of the
        #  Swin-T blocks, ops.LayerNormRowsFn / GeluFn / WindowAttnFn / LinearFn)
        enc_checked += n_.startswith("img_encoder.")
        assert prm[n_].grad is not None, n_
        if gd[0] < 1e-4 * med:
            assert float(_digest(prm[n_].grad)[0]) < 1e-3 * med, n_
            continue
        worst = max(worst, _rel(_digest(prm[n_].grad), torch.from_numpy(gd)))
        n_checked += 1
    for n_ in (str(s) for s in Gd["nograd_names"]):
        assert prm[n_].grad is None, n_
    REPORT[f"{tag}_step[fp32].worst_grad_digest"] = {"rel_err": worst, "tol": 1e-4, "tensors": n_checked, "img_encoder_tensors": enc_checked}
    assert worst < 1e-4 and n_checked >= 80, (worst, n_checked)
    assert (enc_checked > 150) == (name in ("bi_vsltimg_mbt_v1", "tri_mbt_v2", "tri_mbt_vnoshavgtr")), enc_checked      # 171 encoder tensors with a gradient there


class _Logger:
    class _Ev:
        def __init__(self):
            self.calls = []

        def add_batch(self, t, o):
            self.calls.append((t, o))

    def __init__(self):
        self.evaluator, self.lrs = self._Ev(), []

    def log_lr(self, lr, it):
        self.lrs.append(lr)


def _digest(t):
    f = t.detach().reshape(-1).double().cpu()
    return torch.cat([f.norm().view(1), f[torch.linspace(0, f.numel() - 1, 8).long()]])


def _run_steps(dtype, multi, tag, fused, packed=False, **over):
    from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
    from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
    from medical_tri_modal_pilot_amd.optim import FusedAdamW
    Gd = G(tag)
    args, model = _product_model(2, multi, dtype, **over)
    bt = filler.make_batch(int(Gd["seed"]), int(Gd["B"]), int(Gd["T"]), multiimages=multi)
    model.train()
    model.img_encoder.eval()
    if fused:
        opt = FusedAdamW(model.hot_parameters(), lr=args.lr_init, weight_decay=args.weight_decay)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=args.lr_init, weight_decay=args.weight_decay)
    sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=args.t_0 * 10, cycle_mult=args.t_mult,
                                          max_lr=args.lr_init * math.sqrt(args.batch_size), min_lr=1e-6,
                                          warmup_steps=args.t_up * 10, gamma=args.gamma)
    crit = torch.nn.BCEWithLogitsLoss(reduction="mean")
    lg = _Logger()
    static = torch.stack([bt["gen"], bt["age"]], 1)
    x_in = bt["x"]
    if packed:                        # the same batch in the ragged layout of builder/data
        from medical_tri_modal_pilot_amd.builder.data import collate_packed
        x_in = collate_packed([(bt["x"][b, :int(n)].numpy(), static[b].numpy(), float(bt["txt_time"][b]))
                               for b, n in enumerate(bt["input_lengths"])])
        args.TIE_len = bt["x"].shape[1]
    kw = dict(args=args, x=x_in, static=static, y=bt["y"], output_lengths=None, model=model, logger=lg,
              device=torch.device(DEV), scheduler=sched, optimizer=opt, criterion=crit, x_txt=bt["txt"],
              x_img=bt["img"], imgtxt_time=(bt["img_time"], bt["txt_time"]), scaler=None, missing=bt["missing"],
              reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
    grads = {}
    def keep(p_, n_):
        if p_.grad is not None:       # parameters of skipped blocks reach the hook with an undefined gradient
            grads[n_] = p_.grad.detach().clone()

    hooks = [p.register_post_accumulate_grad_hook(lambda p_, n_=n: keep(p_, n_)) for n, p in model.named_parameters()]
    _, loss1 = get_trainer(iteration=1, input_lengths=bt["input_lengths"].clone(),
                           txt_lengths=bt["txt_lengths"].clone(), flow_type="train", **kw)
    for h in hooks:
        h.remove()
    hot = set(n for n, _ in model.hot_parameters())
    for n, p in model.named_parameters():     # gradients written straight into the flat buffer bypass the hooks
        if n not in grads and n in hot and p.grad is not None:
            grads[n] = p.grad.detach().clone()
    lr_after = opt.param_groups[0]["lr"]
    params1 = {n: p.detach().clone() for n, p in model.named_parameters()}
    _, loss2 = get_trainer(iteration=2, input_lengths=bt["input_lengths"].clone(),
                           txt_lengths=bt["txt_lengths"].clone(), flow_type="train", **kw)
    model.eval()
    _, tl = get_trainer(iteration=3, input_lengths=bt["input_lengths"].clone(), txt_lengths=bt["txt_lengths"].clone(),
                        flow_type="test", **kw)
    return Gd, model, grads, params1, loss1, loss2, tl, lg, lr_after


@pytest.mark.parametrize("multi,tag", [(0, "model_step"), (1, "model_step_multi")])
@pytest.mark.parametrize("fused", [True, False])
def test_full_training_step_fp32_vs_golden(ops, multi, tag, fused):
    Gd, model, grads, params1, loss1, loss2, tl, lg, lr_after = _run_steps("fp32", multi, tag, fused)
    t = f"step[fp32,multi={multi},{'fused' if fused else 'torch'}AdamW]"
    REPORT[t + ".loss"] = {"rel_err": abs(loss1 - float(Gd["loss"])), "tol": 1e-4}
    assert abs(loss1 - float(Gd["loss"])) < 1e-4 * max(1.0, abs(float(Gd["loss"])))
    assert abs(lr_after - float(Gd["lr_after"])) < 1e-12
    names = [str(s) for s in Gd["grad_names"]]
    assert sorted(names) == sorted(grads.keys()), set(names) ^ set(grads.keys())
    if fused:
        assert sorted(n for n, _ in model.hot_parameters()) == sorted(names)      # static no-grad set is right
    # Some gradients are mathematically zero (key bias: softmax is shift invariant; biases feeding
    # BatchNorm): the reference holds rounding noise there (norm ~1e-9 vs a median of ~1e-1), so
    # those tensors are only required to be noise-sized too.  AdamW's first update is lr*sign(g):
    # on noise-sized elements the sign is arbitrary, so parameters may differ by up to 2*lr.
    med = float(np.median(Gd["grad_digest"][:, 0]))
    lr1 = 1e-6
    worst_g = worst_p = worst_e = 0.0
    for n_, gd, pd in zip(names, Gd["grad_digest"], Gd["param_digest"]):
        got_g, got_p = _digest(grads[n_]), _digest(params1[n_])
        if gd[0] < 1e-4 * med:
            assert float(got_g[0]) < 1e-3 * med, f"grad {n_} should be ~0, norm {float(got_g[0])}"
        else:
            ref_g = torch.from_numpy(gd)
            eg = _rel(got_g, ref_g)
            worst_g = max(worst_g, eg)
            assert eg < 1e-4, f"grad {n_}: {eg}"                 # north_star: 1e-4 forward + backward (max-norm relative)
            # ... and element by element on the sampled entries that are not small against the tensor's own scale
            big = ref_g[1:].abs() > 0.05 * ref_g[1:].abs().max()
            if bool(big.any()):
                ee = float(((got_g[1:] - ref_g[1:]).abs() / ref_g[1:].abs())[big].max())
                worst_e = max(worst_e, ee)
                assert ee < 2e-3, f"grad {n_}: element-wise relative error {ee}"          # measured worst: 6.7e-4
        dp = float((got_p[1:] - torch.from_numpy(pd)[1:]).abs().max())
        worst_p = max(worst_p, dp)
        assert dp <= 2.2 * lr1, f"param {n_}: |diff| {dp}"
        assert abs(float(got_p[0]) - pd[0]) <= 1e-5 * pd[0] + 1e-9, f"param norm {n_}"
    REPORT[t + ".worst_grad_digest"] = {"rel_err": worst_g, "tol": 1e-4}
    REPORT[t + ".worst_grad_elementwise"] = {"rel_err": worst_e, "tol": 2e-3}
    REPORT[t + ".worst_param_abs_diff_after_adamw"] = {"rel_err": worst_p, "tol": 2.2 * lr1}
    assert abs(loss2 - float(Gd["loss2"])) < 1e-4
    assert abs(tl - float(Gd["test_loss"])) < 1e-4
    check(t + ".test_sigmoid", lg.evaluator.calls[-1][1], torch.from_numpy(Gd["test_sigmoid"]), 1e-4)
    check(t + ".bn_running_mean", model.fc_list[1].running_mean, torch.from_numpy(Gd["bn_running_mean"]), 1e-4)


@pytest.mark.parametrize("graph", [0, 1])
def test_packed_batch_training_steps_vs_golden(ops, graph):
    """Two train steps + one test step fed with the ragged PackedTieBatch instead of the padded x: same losses as
    the reference-generated golden (1e-4), eager and through the hipGraph replay path."""
    Gd, model, grads, params1, loss1, loss2, tl, lg, _ = _run_steps("fp32", 0, "model_step", True, packed=True,
                                                                    hip_graph=graph)
    t = f"step_packed[fp32,graph={graph}]"
    REPORT[t + ".loss"] = {"rel_err": abs(loss1 - float(Gd["loss"])), "tol": 1e-4}
    assert abs(loss1 - float(Gd["loss"])) < 1e-4 * max(1.0, abs(float(Gd["loss"])))
    assert abs(loss2 - float(Gd["loss2"])) < 1e-4
    assert abs(tl - float(Gd["test_loss"])) < 1e-4


# ~2 x the figures measured on MI355X (see the test): the worst tensor, AND the median / 95th percentile over the tensors -- a
# regression that doubles the error of every well-conditioned tensor moves the median, not the worst one
# (round 4, measured: median 3.0e-2, 95th percentile 5.1e-2, worst 6.9e-2 over the 85 tensors with a non-zero gradient -- at B = 4 the
#  whole gradient flows through four CLS rows, so bf16 rounding does not average out and the tensors share most of their error)
BF16_STEP_GATES = dict(loss=3.5e-3, grad_norm=0.14, grad_digest=0.14, test_loss=1.8e-3,
                       grad_norm_median=6e-2, grad_norm_p95=0.1, grad_digest_median=6e-2, grad_digest_p95=0.1)


def test_full_training_step_bf16_tolerance(ops):
    """bf16 MFMA build against the fp32 golden of the reference: the benchmarked build runs its own kernels for the heaviest
    GEMMs (LDS-DMA row-panel / weight-gradient kernels), so its step is gated per tensor at about twice the error measured on
    MI355X (round 3: loss |diff| 6.2e-4 ... 1.7e-3 -- it moved within that range when the image encoder's attention half became
    one kernel with another accumulation order, the encoder's features keeping their 1.2e-2 against the golden --, test loss
    8.7e-4; worst tensor 7e-2 in L2 norm and in its sampled entries, median tensor 3e-2, 95th percentile 5e-2), not at a blanket figure: the
    worst tensor, the median and the 95th percentile each have their own gate."""
    Gd, model, grads, params1, loss1, loss2, tl, lg, _ = _run_steps("bf16", 0, "model_step", True)
    gt = BF16_STEP_GATES
    REPORT["step[bf16].loss_abs_err"] = {"rel_err": abs(loss1 - float(Gd["loss"])), "tol": gt["loss"]}
    names = [str(s) for s in Gd["grad_names"]]
    worst_n = worst_d = 0.0
    worst_name = ""
    all_n, all_d = [], []
    med = float(np.median(Gd["grad_digest"][:, 0]))
    for n_, gd in zip(names, Gd["grad_digest"]):
        dg = _digest(grads[n_])
        got, ref = float(dg[0]), float(gd[0])          # gradient L2 norms
        if ref < 1e-4 * med:                                           # mathematically-zero gradients: noise
            assert got < 1e-2 * med, n_
            continue
        en = abs(got - ref) / abs(ref)
        ed = _rel(dg, torch.from_numpy(gd))                            # norm + 8 sampled entries, relative to the largest
        if ed > worst_d:
            worst_d, worst_name = ed, n_
        worst_n = max(worst_n, en)
        all_n.append(en)
        all_d.append(ed)
    REPORT["step[bf16].worst_grad_norm_rel_err"] = {"rel_err": worst_n, "tol": gt["grad_norm"]}
    REPORT["step[bf16].worst_grad_digest_rel_err"] = {"rel_err": worst_d, "tol": gt["grad_digest"], "tensor": worst_name}
    REPORT["step[bf16].test_loss_abs_err"] = {"rel_err": abs(tl - float(Gd["test_loss"])), "tol": gt["test_loss"]}
    stats = {"grad_norm_median": float(np.median(all_n)), "grad_norm_p95": float(np.percentile(all_n, 95)),
             "grad_digest_median": float(np.median(all_d)), "grad_digest_p95": float(np.percentile(all_d, 95))}
    for k, v in stats.items():
        REPORT[f"step[bf16].{k}_rel_err"] = {"rel_err": v, "tol": gt[k], "tensors": len(all_n)}
    assert abs(loss1 - float(Gd["loss"])) < gt["loss"]
    assert worst_n < gt["grad_norm"], worst_n
    assert worst_d < gt["grad_digest"], (worst_d, worst_name)
    assert abs(tl - float(Gd["test_loss"])) < gt["test_loss"]
    for k, v in stats.items():
        assert v < gt[k], (k, v, gt[k])


@pytest.mark.parametrize("M", [64320, 4990, 300])
def test_bf16_row_panel_kernels_vs_fp32_kernels_on_identical_inputs(ops, M):
    """the bf16 build's heaviest GEMMs are their own kernels (LDS-DMA row panels), not the fp32 kernels with
    another MFMA.  Same bf16-representable inputs through both builds: ln_gemm (QKV and FFN1 + ReLU), the sign-gated dH
    product and gemm_lnbwd agree to bf16 rounding of the outputs (2^-8 of the tensor's scale, + accumulation order)."""
    g = torch.Generator().manual_seed(M + 5)
    bf, f32 = torch.bfloat16, torch.float32
    x = torch.randn(M, 256, generator=g).to(DEV, bf)
    gm, bt = (1 + 0.1 * torch.randn(256, generator=g)).to(DEV), (0.1 * torch.randn(256, generator=g)).to(DEV)
    for n_out, relu in ((768, False), (1024, True)):
        w = (torch.randn(n_out, 256, generator=g) * 0.06).to(DEV, bf)
        b = (torch.randn(n_out, generator=g) * 0.2).to(DEV)
        yb, xnb, stb = ops.ln_gemm(x, gm, bt, w, b, n_out, relu=relu)[:3]
        yf, xnf, stf = ops.ln_gemm(x.float(), gm, bt, w.float(), b, n_out, relu=relu)[:3]
        check(f"bf16_vs_fp32.ln_gemm[M={M},N={n_out}].y", yb.float(), yf, 8e-3)                # measured <= 3.7e-3
        check(f"bf16_vs_fp32.ln_gemm[M={M},N={n_out}].xn", xnb.float(), xnf, 6e-3)              # measured <= 3.0e-3
        check(f"bf16_vs_fp32.ln_gemm[M={M},N={n_out}].stats", stb, stf, 1e-5)
    # dH = (dY W2) gated by h > 0: sign bits (bf16 kernel) against the stored activation (fp32 kernel)
    w1 = (torch.randn(1024, 256, generator=g) * 0.06).to(DEV, bf)
    h, _, _, sg = ops.ln_gemm(x, gm, bt, w1, None, 1024, relu=True, want_signs=True)
    dy = torch.randn(M, 256, generator=g).to(DEV, bf)
    w2t = (torch.randn(1024, 256, generator=g) * 0.05).to(DEV, bf)
    dhb = ops.gemm_nt_signs(dy, w2t, sg, 1.25)
    dhf = ops.gemm_nt(dy.float(), w2t.float(), gate=h.float(), gate_scale=1.25)
    check(f"bf16_vs_fp32.gemm_nt_signs[M={M}]", dhb.float(), dhf, 6.5e-3)                        # measured <= 3.2e-3
    # dz = LNbackward(dY Wt^T) + d_res
    for K in (768, 1024):
        dyk = torch.randn(M, K, generator=g).to(DEV, bf)
        wt = (torch.randn(256, K, generator=g) * 0.05).to(DEV, bf)
        st = torch.stack([x.float().mean(-1), 1 / (x.float().std(-1) + 1e-6)], 1).contiguous()
        dres = torch.randn(M, 256, generator=g).to(DEV, bf)
        zb, ggb, gbb = ops.gemm_lnbwd(dyk, wt, x, st, gm, d_res2d=dres)
        zf, ggf, gbf = ops.gemm_lnbwd(dyk.float(), wt.float(), x.float(), st, gm, d_res2d=dres.float())
        check(f"bf16_vs_fp32.gemm_lnbwd[M={M},K={K}].dz", zb.float(), zf, 1e-2)                  # measured <= 5.1e-3
        check(f"bf16_vs_fp32.gemm_lnbwd[M={M},K={K}].dgamma", ggb, ggf, 5e-3)                      # measured <= 2.2e-3
        check(f"bf16_vs_fp32.gemm_lnbwd[M={M},K={K}].dbeta", gbb, gbf, 5e-3)                       # measured <= 2.5e-3


# ------------------------------------------------------------------ hipGraph replay of the step
def _loop(hip_graph, dropout, dtype, n_steps, lens_per_step, L=2, B=4, T=96, missing_on_device=False, ptrs=None, **over):
    from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
    from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
    from medical_tri_modal_pilot_amd.optim import FusedAdamW
    torch.manual_seed(7)
    args, model = _product_model(L, 0, dtype, hip_graph=hip_graph, dropout=dropout, **over)
    model.train()
    model.img_encoder.eval()
    opt = FusedAdamW(model.hot_parameters(), lr=1e-4, weight_decay=args.weight_decay)
    sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=100, cycle_mult=1, max_lr=1e-3, min_lr=1e-6,
                                          warmup_steps=10, gamma=1.0)
    crit = torch.nn.BCEWithLogitsLoss(reduction="mean")
    losses = []
    for it in range(n_steps):
        bt = filler.make_batch(900 + it, B, T, ragged=True, missing_mode="mixed" if it % 2 else "none")
        if lens_per_step is not None:
            bt["input_lengths"] = torch.tensor(lens_per_step[it])
        static = torch.stack([bt["gen"], bt["age"]], 1)
        miss = bt["missing"]
        if missing_on_device:             # a fresh device tensor per step, freed at the end of the iteration
            miss = bt["missing"].to(DEV)
            if ptrs is not None:
                ptrs.append(miss.data_ptr())
        _, loss = get_trainer(args=args, iteration=it + 1, x=bt["x"], static=static, y=bt["y"], output_lengths=None,
                              model=model, logger=_Logger(), device=torch.device(DEV), scheduler=sched, optimizer=opt,
                              criterion=crit, x_txt=bt["txt"], x_img=bt["img"],
                              imgtxt_time=(bt["img_time"], bt["txt_time"]), scaler=None, missing=miss,
                              input_lengths=bt["input_lengths"], txt_lengths=bt["txt_lengths"], flow_type="train",
                              reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
        losses.append(loss)
        del miss
    _loop.last_packed = getattr(model.fusion_transformer, "last_pack", None) is not None      # (a flag, not the model: no graph outlives its test)
    _loop.last_layout = list(zip(opt.flat.names, opt.flat.offsets))                           # (for diagnostics: which parameter differs)
    return losses, opt.flat.data.detach().clone(), getattr(model, "_mtmp_graph_step", None)


def test_graph_replay_equals_eager_steps(ops):
    """The captured step is the eager step: same kernels in the same order -> bit-identical losses and
    parameters after 5 optimisation steps on 5 different ragged batches with mixed missing-modality patterns."""
    full = [[96, 96, 50, 7]] * 5                       # max_len == T: same trimmed shape in both modes
    le, pe, _ = _loop(0, 0.0, "fp32", 5, full)
    lg, pg, gs = _loop(1, 0.0, "fp32", 5, full)
    assert gs is not None and gs.captures == 1 and gs.replays == 4 and not gs.disabled
    assert le == lg, (le, lg)
    assert torch.equal(pe, pg)
    REPORT["graph_vs_eager[fp32].5_steps"] = {"rel_err": 0.0, "tol": 0.0}


@pytest.mark.parametrize("graph", [0, 1])
def test_fresh_device_missing_tensor_every_step(ops, graph):
    """ADVICE r2 (high) / VERDICT P5: `missing` handed in as a FRESH device tensor each step (alternating all-present and
    mixed-missing batches; the caching allocator returns the same address with _version 0 every time) must give the
    pattern ids of THAT batch: losses and parameters bit-identical to the run that keeps `missing` on the host."""
    full = [[96, 96, 50, 7]] * 4
    lh, ph, _ = _loop(graph, 0.0, "fp32", 4, full)
    ptrs = []
    ld, pd, _ = _loop(graph, 0.0, "fp32", 4, full, missing_on_device=True, ptrs=ptrs)
    REPORT[f"fresh_device_missing[graph={graph}].address_reused"] = {"rel_err": float(len(set(ptrs)) < len(ptrs)), "tol": 1.0}
    assert lh == ld, (lh, ld)
    assert torch.equal(ph, pd)


@pytest.mark.parametrize("vsltonly", [0, 1])
def test_staged_graphs_equal_single_graph(ops, vsltonly):
    """--graph-stages 3 (what data-parallel steps use so that bucket all-reduces overlap the backward): the fusion
    stack is cut into three chained autograd nodes and the step into three hipGraphs replayed back to back.  Same
    kernels, same order -> losses and parameters bit-identical to the one-graph step, and to eager."""
    full = [[96, 96, 50, 7]] * 5
    le, pe, _ = _loop(0, 0.0, "fp32", 5, full, L=4, mbt_only_vslt=vsltonly)
    l1, p1, g1 = _loop(1, 0.0, "fp32", 5, full, L=4, mbt_only_vslt=vsltonly, graph_stages=1)
    l3, p3, g3 = _loop(1, 0.0, "fp32", 5, full, L=4, mbt_only_vslt=vsltonly, graph_stages=3)
    assert g3.captures == 1 and g3.replays == 4 and not g3.disabled
    assert [len(e["graphs"]) for e in g3.entries.values()] == [3] and [len(e["graphs"]) for e in g1.entries.values()] == [1]
    assert le == l1 == l3, (le, l1, l3)
    assert torch.equal(pe, p1) and torch.equal(p1, p3)
    REPORT[f"staged_graphs_vs_single[fp32,vsltonly={vsltonly}].5_steps"] = {"rel_err": 0.0, "tol": 0.0}


def test_vsltonly0_training_step_vs_oracle(ops):
    """--mbt-only-vslt 0 (the flag's default): the last layer's image / text blocks run forward but nothing reads their
    outputs (tri_mbt_vsltcls.py:248), so the reference leaves their gradient None.  The product gives them no backward
    and keeps them out of AdamW; loss, second-step loss and every other gradient follow the CPU oracle."""
    from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
    from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
    from medical_tri_modal_pilot_amd.optim import FusedAdamW
    L = 2
    args, model = _product_model(L, 0, "fp32", hip_graph=0, mbt_only_vslt=0)
    model.train()
    model.img_encoder.eval()
    hot = dict(model.hot_parameters())
    assert not any(n.startswith((f"fusion_transformer.layer_stacks.{L - 1}.1.", f"fusion_transformer.layer_stacks.{L - 1}.2."))
                   for n in hot)
    opt = FusedAdamW(list(hot.items()), lr=args.lr_init, weight_decay=args.weight_decay)
    sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=args.t_0 * 10, cycle_mult=args.t_mult,
                                          max_lr=args.lr_init * math.sqrt(args.batch_size), min_lr=1e-6,
                                          warmup_steps=args.t_up * 10, gamma=args.gamma)
    bt = filler.make_batch(4321, 4, 40)
    static = torch.stack([bt["gen"], bt["age"]], 1)
    kw = dict(args=args, x=bt["x"], static=static, y=bt["y"], output_lengths=None, model=model, logger=_Logger(),
              device=torch.device(DEV), scheduler=sched, optimizer=opt, criterion=torch.nn.BCEWithLogitsLoss(),
              x_txt=bt["txt"], x_img=bt["img"], imgtxt_time=(bt["img_time"], bt["txt_time"]), scaler=None,
              missing=bt["missing"], reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    _, loss1 = get_trainer(iteration=1, input_lengths=bt["input_lengths"].clone(), txt_lengths=bt["txt_lengths"].clone(),
                           flow_type="train", **kw)
    grads = {n: p.grad.detach().clone() for n, p in hot.items()}
    _, loss2 = get_trainer(iteration=2, input_lengths=bt["input_lengths"].clone(), txt_lengths=bt["txt_lengths"].clone(),
                           flow_type="train", **kw)
    tr = O.OracleTrainer(_model_sd(L), O.Cfg(n_layers=L, vsltonly=0), lr_init=args.lr_init, batch_size=args.batch_size,
                         iters_per_epoch=10)
    ref1 = tr.step(bt, 1)
    ref_grads = {k: v.clone() for k, v in tr.grads.items()}
    ref2 = tr.step(bt, 2)
    assert abs(loss1 - ref1) < 1e-4 and abs(loss2 - ref2) < 1e-4, (loss1, ref1, loss2, ref2)
    assert sorted(ref_grads) == sorted(hot), set(ref_grads) ^ set(hot)        # the oracle's autograd reaches the same set
    med = float(np.median([float(v.norm()) for v in ref_grads.values()]))
    worst = 0.0
    for n, g in ref_grads.items():
        if float(g.norm()) < 1e-4 * med:
            continue
        worst = max(worst, _rel(grads[n], g))
    REPORT["vsltonly0_step[fp32].worst_grad"] = {"rel_err": worst, "tol": 1e-4}
    assert worst < 1e-4
    for n, p in model.named_parameters():          # untouched: the skipped blocks (no weight decay either) and the frozen Swin
        if n not in hot:
            assert torch.equal(p.detach(), before[n]), n


def test_cfg5_shape_four_images_twelve_layers_vs_oracle(ops):
    """BASELINE configs[4] structure at a size the CPU oracle finishes in seconds: --multiimages 1 generalised to
    --n-images 4 (the reference hard-codes 3, tri_mbt_vsltcls.py:161-162,226-231), 12 fusion layers, ragged vital-sign
    series, one absent image slot per sample at most.  fp32 build, two train steps against the oracle; the image
    stream's valid-key counts (4 + 1 + 49 * #present) are bit-exact."""
    from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
    from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
    from medical_tri_modal_pilot_amd.optim import FusedAdamW
    L, K, B, T = 12, 4, 4, 40      # (B = 2 would make BatchNorm1d output +-1 whatever its input: every gradient ~ eps)
    args, model = _product_model(L, 1, "fp32", hip_graph=0, n_images=K, batch_size=B)
    assert model.n_images == K
    model.train()
    model.img_encoder.eval()
    opt = FusedAdamW(model.hot_parameters(), lr=args.lr_init, weight_decay=args.weight_decay)
    sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=args.t_0 * 10, cycle_mult=args.t_mult,
                                          max_lr=args.lr_init * math.sqrt(args.batch_size), min_lr=1e-6,
                                          warmup_steps=args.t_up * 10, gamma=args.gamma)
    bt = filler.make_batch(777, B, T, multiimages=1, n_images=K, missing_mode="none")
    assert bt["img"].shape == (B, K, 1, 224, 224) and bt["img_time"].shape == (B, K)
    # valid image keys per sample, as the encoder derives them (integer artefact: bit-exact)
    present = (bt["img_time"] != 10).sum(1)
    lens = model.fusion_transformer.key_lengths([bt["input_lengths"], present * 49, bt["txt_lengths"] + 2], "cpu")
    assert torch.equal(lens[1], present * 49 + 1)
    static = torch.stack([bt["gen"], bt["age"]], 1)
    kw = dict(args=args, x=bt["x"], static=static, y=bt["y"], output_lengths=None, model=model, logger=_Logger(),
              device=torch.device(DEV), scheduler=sched, optimizer=opt, criterion=torch.nn.BCEWithLogitsLoss(),
              x_txt=bt["txt"], x_img=bt["img"], imgtxt_time=(bt["img_time"], bt["txt_time"]), scaler=None,
              missing=bt["missing"], reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
    _, loss1 = get_trainer(iteration=1, input_lengths=bt["input_lengths"].clone(), txt_lengths=bt["txt_lengths"].clone(),
                           flow_type="train", **kw)
    grads = {n: p.grad.detach().clone() for n, p in model.hot_parameters()}
    _, loss2 = get_trainer(iteration=2, input_lengths=bt["input_lengths"].clone(), txt_lengths=bt["txt_lengths"].clone(),
                           flow_type="train", **kw)
    tr = O.OracleTrainer(_model_sd(L), O.Cfg(n_layers=L, multiimages=1), lr_init=args.lr_init, batch_size=args.batch_size,
                         iters_per_epoch=10)
    ref1 = tr.step(bt, 1)
    ref_grads = {k: v.clone() for k, v in tr.grads.items()}
    ref2 = tr.step(bt, 2)
    REPORT["cfg5_k4_L12[fp32].loss1"] = {"rel_err": abs(loss1 - ref1), "tol": 1e-4}
    REPORT["cfg5_k4_L12[fp32].loss2"] = {"rel_err": abs(loss2 - ref2), "tol": 1e-4}
    assert abs(loss1 - ref1) < 1e-4 and abs(loss2 - ref2) < 1e-4, (loss1, ref1, loss2, ref2)
    assert sorted(ref_grads) == sorted(grads)
    med = float(np.median([float(v.norm()) for v in ref_grads.values()]))
    errs = sorted(((_rel(grads[n], g), n) for n, g in ref_grads.items() if float(g.norm()) >= 1e-4 * med), reverse=True)
    # 12 layers deep on 4 x 45 tokens, a ReLU gate whose pre-activation differs in the last bit between the two
    # machines flips, and one flipped gate is ~0.5 % of an FFN bias gradient summed over 180 tokens (measured worst:
    # 2.5e-3 on a w_1.bias; the 2-layer golden step stays at 1.9e-5): the typical gradient is held to 1e-4, the worst
    # to 1e-2
    worst, typical = errs[0][0], errs[len(errs) // 2][0]
    REPORT["cfg5_k4_L12[fp32].worst_grad"] = {"rel_err": worst, "tol": 3e-3}
    REPORT["cfg5_k4_L12[fp32].median_grad"] = {"rel_err": typical, "tol": 1e-4}
    assert worst < 3e-3 and typical < 1e-4, errs[:8]                 # measured 1.35e-3 / 3.1e-5
    # The explanation above, SHOWN: the hidden units whose ReLU gate differs between the two machines are
    # found from the gradients themselves -- a flipped gate moves ONE entry of that layer's w_1.bias gradient by a whole
    # token's dh, orders of magnitude above rounding -- and with exactly those units masked out (w_1 rows / w_1.bias entries /
    # w_2 columns) every FFN gradient agrees to 3.5e-4 (measured: 2 flipped units among 23 x 1024, 1.7e-4 -- what is left is
    # the flipped token's dx travelling on into the layers below, which no mask of hidden units removes).
    flips = 0
    worst_masked = 0.0
    for n_, g_ref in ref_grads.items():
        if not n_.endswith("feed_forward.w_1.bias"):
            continue
        pre = n_[:-len("w_1.bias")]
        d = (grads[n_].cpu() - g_ref).abs()
        flipped = d > 1e-4 * g_ref.abs().max()                      # units off by more than the gate of 1e-4
        flips += int(flipped.sum())
        keep = ~flipped
        for suffix, sel in (("w_1.bias", lambda t: t[keep]), ("w_1.weight", lambda t: t[keep]), ("w_2.weight", lambda t: t[:, keep])):
            a_, b_ = sel(grads[pre + suffix].cpu()), sel(ref_grads[pre + suffix])
            e = float((a_ - b_).abs().max() / (ref_grads[pre + suffix].abs().max() + 1e-30))
            worst_masked = max(worst_masked, e)
    REPORT["cfg5_k4_L12[fp32].ffn_grads_with_flipped_gates_masked"] = {"rel_err": worst_masked, "tol": 3.5e-4, "flipped_units": flips}
    assert flips <= 12, flips                                        # a handful among 23 x 1024 hidden units x 180 tokens
    assert worst_masked < 3.5e-4, worst_masked


def _one_train_step(dtype, B, T, L, multi=0, K=3, seed=1234, batch=None):
    """One eager train step of the product at an arbitrary size (filler weights, synthetic batch, dropout 0, frozen encoder in
    eval mode): loss, the gradient of every trained parameter, the batch."""
    from medical_tri_modal_pilot_amd import synthetic
    from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
    from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
    from medical_tri_modal_pilot_amd.optim import FusedAdamW
    torch.manual_seed(5)
    over = dict(hip_graph=0, batch_size=B, TIE_len=T, dropout=0.0)
    if multi:
        over["n_images"] = K
    args, model = _product_model(L, multi, dtype, **over)
    model.train()
    model.img_encoder.eval()
    opt = FusedAdamW(model.hot_parameters(), lr=args.lr_init, weight_decay=args.weight_decay)
    sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=args.t_0 * 10, cycle_mult=args.t_mult,
                                          max_lr=args.lr_init * math.sqrt(args.batch_size), min_lr=1e-6,
                                          warmup_steps=args.t_up * 10, gamma=args.gamma)
    bt = batch if batch is not None else synthetic.make_batch(seed, B, T, ragged=True, missing_mode="mixed", multiimages=multi, n_images=K)
    static = torch.stack([bt["gen"], bt["age"]], 1)
    grads = {}
    hooks = [p.register_post_accumulate_grad_hook(lambda p_, n_=n: grads.__setitem__(n_, p_.grad.detach().clone()) if p_.grad is not None else None)
             for n, p in model.named_parameters()]
    # (gradients before the optimizer touches anything: FusedAdamW.step reads them, it does not change them)
    _, loss = get_trainer(args=args, iteration=1, x=bt["x"], static=static, y=bt["y"], output_lengths=None, model=model,
                          logger=_Logger(), device=torch.device(DEV), scheduler=sched, optimizer=opt,
                          criterion=torch.nn.BCEWithLogitsLoss(reduction="mean"), x_txt=bt["txt"], x_img=bt["img"],
                          imgtxt_time=(bt["img_time"], bt["txt_time"]), scaler=None, missing=bt["missing"],
                          input_lengths=bt["input_lengths"].clone(), txt_lengths=bt["txt_lengths"].clone(), flow_type="train",
                          reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
    for h in hooks:
        h.remove()
    hot = set(n for n, _ in model.hot_parameters())
    for n, p in model.named_parameters():     # gradients written straight into the flat buffer bypass the hooks
        if n not in grads and n in hot and p.grad is not None:
            grads[n] = p.grad.detach().clone()
    torch.cuda.synchronize()
    out = {n: g.float().cpu() for n, g in grads.items()}
    del model, opt, grads
    torch.cuda.empty_cache()
    return float(loss), out, bt


def _tensor_errors(got, ref):
    """per-tensor |got - ref|_2 / |ref|_2 over the tensors whose reference gradient is not noise-sized; sorted, worst first"""
    med = float(np.median([float(v.norm()) for v in ref.values()]))
    return sorted(((float((got[n].float() - g.float()).norm() / g.float().norm()), n) for n, g in ref.items()
                   if float(g.norm()) >= 1e-4 * med), reverse=True)


def test_config2_full_size_fp32_step_vs_oracle(ops):
    """parity AT THE SIZE THE BENCH RUNS -- BASELINE configs[1]: B 64, TIE-len 1000 (N_v 1005), 6 layers,
    one 224 x 224 image, 128 text tokens, ragged lengths and mixed missing modalities -- one train step of the fp32 HIP build
    against the oracle's step on the host cores (~1 min): loss at 1e-4, every trained parameter's gradient (relative L2 error)
    at 1e-4 for the typical tensor; the worst tensor is stated and gated (ReLU gates whose pre-activation differs in the
    last bit flip at this depth and width, as at configs[4]: see test_cfg5_shape_four_images_twelve_layers_vs_oracle)."""
    B, T, L = 64, 1000, 6
    loss, grads, bt = _one_train_step("fp32", B, T, L)
    tr = O.OracleTrainer(_model_sd(L), O.Cfg(n_layers=L), lr_init=1e-5, batch_size=B, iters_per_epoch=10)
    ref = tr.step(bt, 1)
    assert sorted(tr.grads) == sorted(grads)
    errs = _tensor_errors(grads, tr.grads)
    worst, typical = errs[0][0], errs[len(errs) // 2][0]
    REPORT["config2_full_size[fp32].loss"] = {"rel_err": abs(loss - ref), "tol": 1e-4}
    REPORT["config2_full_size[fp32].median_grad"] = {"rel_err": typical, "tol": 1e-4, "tensors": len(errs)}
    REPORT["config2_full_size[fp32].worst_grad"] = {"rel_err": worst, "tol": 1.5e-2, "tensor": errs[0][1]}
    over = [(e, n) for e, n in errs if e > 1e-3]
    # The tensors over 1e-3 are the ones DOWNSTREAM (in the backward) of a ReLU gate that differs between the two machines: a hidden
    # unit whose pre-activation sits within rounding of zero for one token is on for one side and off for the other -- ONE entry of
    # that layer's w_1.bias gradient moves by a whole token's dh (orders of magnitude above rounding), and the token's dx carries the
    # difference into every tensor in front of it (tools/dbg/fullsize_err_probe.py: B 16 / T 300 / L 6 has one such unit in the image
    # stream's layer 2 and 65 tensors over 1e-3 behind it; many other shapes have none).  Counted here from the gradients themselves,
    # as in the configs[4] test.  Measured (round 5, this batch): loss |diff| 1.8e-7, median tensor 1.6e-5, 12 tensors over 1e-3 --
    # the text stream's layer 0 and the text inputs in front of it --, worst 6.0e-3.
    flips = 0
    for n_, g_ref in tr.grads.items():
        if n_.endswith("feed_forward.w_1.bias"):
            flips += int(((grads[n_] - g_ref).abs() > 1e-4 * g_ref.abs().max()).sum())
    REPORT["config2_full_size[fp32].tensors_over_1e-3"] = {"rel_err": float(len(over)), "tol": 24.0, "flipped_relu_gates": flips,
                                                            "names": [n for _, n in over]}
    assert abs(loss - ref) < 1e-4, (loss, ref)
    assert typical < 1e-4 and worst < 1.5e-2 and len(over) <= 24, errs[:10]
    assert flips <= 12 and (flips >= 1 or not over), (flips, over[:4])      # every excursion has a flipped gate to point at


# bf16 build against the fp32 build (both HIP), same weights and batch, at the benchmarked sizes (VERDICT r4 item 2 ii): loss |diff| and,
# per trained tensor, the relative L2 error and the cosine of the gradient.  What sets these numbers (tools/dbg/bf16_err_probe.py,
# round 5): not a kernel -- padded / packed rows, grouped / single launches give the SAME bf16 gradients to 1e-7 -- but ReLU gates.  The
# FFN pre-activation is computed from bf16-rounded inputs (2^-9 relative), so the ~0.25 % of hidden units that sit within that
# rounding of zero are on in one build and off in the other; each such unit moves its token's dH by 100 %, i.e. ~5 % relative L2 per
# FFN, in quadrature over the layers: median tensor 0.09-0.12 at 64 tokens, 0.16-0.18 at 1000 (cosine 0.988-0.996), the worst
# tensors (pre-norm gains / biases: sums with heavy cancellation) 0.5-0.6 (cosine 0.84).  Any bf16 implementation of this model,
# torch autocast on the reference included, has this property.  Gates: 2 x the measured (1 - cosine), 2 x the loss |diff|.
BF16_VS_FP32_MEASURED = {"config2": dict(loss=1.15e-3, l2_median=0.161, cos_median=0.9877, cos_p5=0.8876, cos_worst=0.8431),
                         "cfg5": dict(loss=2.23e-3, l2_median=0.136, cos_median=0.9912, cos_p5=0.9298, cos_worst=0.8387)}


@pytest.mark.parametrize("name,B,T,L,multi,K", [("config2", 64, 1000, 6, 0, 3), ("cfg5", 128, 2000, 12, 1, 4)])
def test_bf16_build_vs_fp32_build_at_benchmark_sizes(ops, name, B, T, L, multi, K):
    """The benchmarked (bf16) build against the parity (fp32) build on the benchmark's own shapes -- configs[1] and configs[4] at
    full size, ragged lengths, mixed missing modalities: what 'bf16, measured tolerance' means where the numbers are quoted
    (SURVEY section 7), not at the B = 4 toy shape of BF16_STEP_GATES.  Both are HIP; seconds."""
    lf, gf, bt = _one_train_step("fp32", B, T, L, multi, K)
    lb, gb, _ = _one_train_step("bf16", B, T, L, multi, K, batch=bt)
    assert sorted(gf) == sorted(gb)
    errs = _tensor_errors(gb, gf)
    keys = [n for _, n in errs]
    e = np.array([x for x, _ in errs])
    cos = np.array([float((gb[n] * gf[n]).sum() / (gb[n].norm() * gf[n].norm())) for n in keys])
    got = dict(loss=abs(lb - lf), l2_median=float(np.median(e)), l2_p95=float(np.percentile(e, 95)), l2_worst=float(e.max()),
               cos_median=float(np.median(cos)), cos_p5=float(np.percentile(cos, 5)), cos_worst=float(cos.min()))
    m = BF16_VS_FP32_MEASURED[name]
    gates = dict(loss=2 * m["loss"], l2_median=2 * m["l2_median"], cos_median=1 - 2 * (1 - m["cos_median"]),
                 cos_p5=1 - 2 * (1 - m["cos_p5"]), cos_worst=1 - 2 * (1 - m["cos_worst"]))
    for k, v in got.items():
        REPORT[f"bf16_vs_fp32[{name},B={B},T={T},L={L}].{k}"] = {"rel_err": v, "tol": gates.get(k, float("nan")), "tensors": len(errs),
                                                                  **({"tensor": errs[0][1]} if k == "l2_worst" else {})}
    assert got["loss"] < gates["loss"] and got["l2_median"] < gates["l2_median"], got
    assert got["cos_median"] > gates["cos_median"] and got["cos_p5"] > gates["cos_p5"] and got["cos_worst"] > gates["cos_worst"], got


def test_cfg5_full_size_properties(ops):
    """BASELINE configs[4] at FULL size on one GPU (B 128, TIE-len 2000 -> N_v 2005, 12 layers, 4 images -> N_i 201, bf16):
    size-independent properties of the whole training step -- finite loss that moves under AdamW, events past a sample's
    length never read (poisoned pad rows: bit-identical losses and parameters), hipGraph replay == eager."""
    from medical_tri_modal_pilot_amd import synthetic
    from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
    from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
    from medical_tri_modal_pilot_amd.optim import FusedAdamW
    L, K, B, T = 12, 4, 128, 2000

    def run(hip_graph, poison):
        torch.manual_seed(11)
        args, model = _product_model(L, 1, "bf16", hip_graph=hip_graph, n_images=K, batch_size=B, TIE_len=T, dropout=0.0)
        model.train()
        model.img_encoder.eval()                  # (no StochasticDepth draws: eager and replayed steps must see the same numbers)
        opt = FusedAdamW(model.hot_parameters(), lr=1e-4, weight_decay=args.weight_decay)
        sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=100, cycle_mult=1, max_lr=1e-3, min_lr=1e-6,
                                              warmup_steps=10, gamma=1.0)
        bt = synthetic.make_batch(99, B, T, ragged=True, missing_mode="none", multiimages=1, n_images=K)
        assert int(bt["input_lengths"].max()) == T
        x = bt["x"].clone()
        if poison:                                # rows past each sample's length: never read by the forward; a FINITE value (the
            for b_ in range(B):                   # inputs pass through fp16): 0 x poison in the backward stays 0 as in the reference
                x[b_, int(bt["input_lengths"][b_]):] = 3e4
        static = torch.stack([bt["gen"], bt["age"]], 1)
        losses = []
        for it in range(3):
            _, loss = get_trainer(args=args, iteration=it + 1, x=x, static=static, y=bt["y"], output_lengths=None, model=model,
                                  logger=_Logger(), device=torch.device(DEV), scheduler=sched, optimizer=opt,
                                  criterion=torch.nn.BCEWithLogitsLoss(), x_txt=bt["txt"], x_img=bt["img"],
                                  imgtxt_time=(bt["img_time"], bt["txt_time"]), scaler=None, missing=bt["missing"],
                                  input_lengths=bt["input_lengths"], txt_lengths=bt["txt_lengths"], flow_type="train",
                                  reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
            losses.append(loss)
        torch.cuda.synchronize()
        return losses, opt.flat.data.detach().clone()

    le, pe = run(0, False)
    assert all(math.isfinite(v) for v in le) and le[0] != le[2], le
    lp, pp = run(0, True)
    assert le == lp and torch.equal(pe, pp), (le, lp)             # poisoned pad events change nothing, bit for bit
    lg, pg = run(1, False)
    assert le == lg and torch.equal(pe, pg), (le, lg)             # graph replay == eager
    REPORT["cfg5_full_size[bf16].properties"] = {"rel_err": 0.0, "tol": 0.0, "losses": le}


def test_cfg1_sample_data_windows_product_vs_oracle(ops):
    """BASELINE configs[0] shape on the reference's own sample data: real TIE windows (the reference's `__getitem__`
    goldens, tests/golden/tie_windows.npz), B 4, TIE-len 256, 2 layers, image and text missing (missing_num 3), fed as
    a ragged PackedTieBatch -- HIP train step (fp32 build) against the CPU oracle's step on the padded batch."""
    from medical_tri_modal_pilot_amd.builder.data import collate_packed
    from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
    from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
    from medical_tri_modal_pilot_amd.optim import FusedAdamW
    g = G("tie_windows")
    off = np.concatenate([[0], np.cumsum(g["seq_rows"])])
    want = [c for c in range(len(g["case"])) if g["case"][c][0] == 1 and g["case"][c][1] == 1000 and 20 <= g["len"][c] <= 256]
    pick = [want[0], want[len(want) // 3], want[2 * len(want) // 3], want[-1]]
    B, T = 4, 256
    samples = [(g["seq_cat"][off[c]:off[c] + int(g["len"][c])], g["static"][c], g["ttime"][c]) for c in pick]
    pb = collate_packed(samples)
    x_pad = pb.to_padded(T)
    bt = dict(x=x_pad, age=pb.static[:, 1].clone(), gen=pb.static[:, 0].clone(), input_lengths=pb.input_lengths,
              txt=torch.zeros(B, 128, 768), txt_lengths=torch.zeros(B, dtype=torch.long), img=torch.zeros(B, 1, 224, 224),
              img_time=torch.full((B,), -1.0), txt_time=pb.txt_time.clone(), y=torch.tensor([0, 1, 1, 0]),
              missing=torch.tensor([[0., 1., 1.]] * B))
    args, model = _product_model(2, 0, "fp32", hip_graph=0)
    args.TIE_len = T
    sd = _model_sd(2)
    model.train()
    model.img_encoder.eval()
    opt = FusedAdamW(model.hot_parameters(), lr=args.lr_init, weight_decay=args.weight_decay)
    sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=args.t_0 * 10, cycle_mult=args.t_mult,
                                          max_lr=args.lr_init * math.sqrt(args.batch_size), min_lr=1e-6,
                                          warmup_steps=args.t_up * 10, gamma=args.gamma)
    kw = dict(args=args, x=pb, static=pb.static, y=bt["y"], output_lengths=None, model=model, logger=_Logger(),
              device=torch.device(DEV), scheduler=sched, optimizer=opt, criterion=torch.nn.BCEWithLogitsLoss(),
              x_txt=bt["txt"], x_img=bt["img"], imgtxt_time=(bt["img_time"], bt["txt_time"]), scaler=None,
              missing=bt["missing"], reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
    _, loss1 = get_trainer(iteration=1, input_lengths=pb.input_lengths, txt_lengths=bt["txt_lengths"], flow_type="train", **kw)
    _, loss2 = get_trainer(iteration=2, input_lengths=pb.input_lengths, txt_lengths=bt["txt_lengths"], flow_type="train", **kw)
    tr = O.OracleTrainer(sd, O.Cfg(n_layers=2), lr_init=args.lr_init, batch_size=args.batch_size, iters_per_epoch=10)
    ref1 = tr.step(bt, 1)
    ref2 = tr.step(bt, 2)
    REPORT["cfg1_sample_data.loss1"] = {"rel_err": abs(loss1 - ref1), "tol": 1e-4}
    REPORT["cfg1_sample_data.loss2"] = {"rel_err": abs(loss2 - ref2), "tol": 1e-4}
    assert abs(loss1 - ref1) < 1e-4 and abs(loss2 - ref2) < 1e-4, (loss1, ref1, loss2, ref2)


@pytest.mark.timeout(420, method="thread")      # (a stalled rendezvous / communicator must fail this test, not hang the suite)
def test_ddp_reducer_single_rank_rccl_matches_plain_run(ops):
    """The data-parallel path on the one GPU of this box: RCCL process group of one rank, ddp.GradReducer attached to
    FusedAdamW (bucketed all-reduce of the flat gradient on a side stream, ready callbacks from the kernels that write
    gradients directly, the modality side streams joined before each bucket).  With one rank the all-reduce is the
    identity, so losses and parameters must equal the plain run bit for bit -- eager (overlapped buckets) and graph."""
    if _in_child_process("test_ddp_reducer_single_rank_rccl_matches_plain_run", timeout=360):
        # the comparison ran (and passed: the child's exit code) in the child session, whose own report is not merged: say so
        # instead of writing a measured-looking 0.0 here
        REPORT["ddp_single_rank_vs_plain[fp32]"] = {"ran_in_child": True, "passed": True, "tol": 0.0}
        return
    import torch.distributed as dist
    from medical_tri_modal_pilot_amd.ddp import GradReducer
    import medical_tri_modal_pilot_amd.optim as optim_mod
    full = [[96, 96, 50, 7]] * 4
    ref = {g: _loop(g, 0.0, "fp32", 4, full)[:2] for g in (0, 1)}
    import datetime
    import socket
    with socket.socket() as sk:                  # a port nobody holds (a fixed one can be taken on a shared host)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")       # one node: RCCL's bootstrap needs no routable interface
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(DEV),
                            timeout=datetime.timedelta(seconds=180))
    orig_init = optim_mod.FusedAdamW.__init__
    made = []

    def init_with_reducer(self, *a, **k):
        orig_init(self, *a, **k)
        self.reducer = GradReducer(self.flat)
        self.reducer.debug = True
        self.grad_scale = 1.0
        made.append(self.reducer)

    optim_mod.FusedAdamW.__init__ = init_with_reducer
    try:
        for g in (0, 1):
            losses, flat, _ = _loop(g, 0.0, "fp32", 4, full)
            assert losses == ref[g][0], (g, losses, ref[g][0])
            assert torch.equal(flat, ref[g][1]), g
        assert made and all(len(r.buckets) >= 1 for r in made)
        # Two independently constructed "ranks" (model + optimizer + reducer + captured graphs each) must issue the same merged
        # collectives in the same order -- a mismatch is the classic multi-rank hang, and with one rank nothing else would notice.
        n_before = len(made)
        _, _, gs2 = _loop(1, 0.0, "fp32", 4, full)
        a, b = made[n_before - 1], made[-1]
        assert a is not b and a.last_issued and a.last_issued == b.last_issued, (a.last_issued, b.last_issued)
        ent = next(iter(gs2.entries.values()))
        plan = b.plan(ent["ready"])
        assert b.last_issued == [tuple(r) for st in plan for r in st], (b.last_issued, plan)
        assert len(ent["graphs"]) == 2 and sum(len(st) for st in plan) <= 3, plan       # two stages: neighbouring buckets merged
    finally:
        optim_mod.FusedAdamW.__init__ = orig_init
        for r in made:
            r.remove()
        dist.destroy_process_group()
    REPORT["ddp_single_rank_vs_plain[fp32]"] = {"rel_err": 0.0, "tol": 0.0}


@pytest.mark.parametrize("overlap", [True, False])
def test_graph_full_size_training_tracks_eager(ops, overlap):
    """Bench-sized steps (B 64, T 1000, 6 layers, frozen Swin in train mode like 2_train.py:128) with the optimizer
    launched right behind every replay.  Without the host wait after graph.replay() this configuration went NaN
    within ~15 steps on ROCm 7.2 (work enqueued after hipGraphLaunch overtook the graph's tail); with it the
    replayed run follows the eager one (not bit-equal: StochasticDepth draws differ under capture)."""
    def loop(hip_graph, n_steps=40):
        from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
        from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
        from medical_tri_modal_pilot_amd.optim import FusedAdamW
        torch.manual_seed(11)
        args, model = _product_model(6, 0, "bf16", hip_graph=hip_graph, dropout=0.0, batch_size=64)
        for mod in model.modules():              # library initialisation like bench.py (the filler weights are tamer)
            if hasattr(mod, "reset_parameters"):
                mod.reset_parameters()
        model.fusion_transformer.overlap_streams = overlap
        model.train()
        opt = FusedAdamW(model.hot_parameters(), lr=1e-5, weight_decay=args.weight_decay)
        sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=5000, cycle_mult=1, max_lr=8e-5, min_lr=1e-6,
                                              warmup_steps=500, gamma=1.0)
        crit = torch.nn.BCEWithLogitsLoss(reduction="mean")
        bt = filler.make_batch(1234, 64, 1000, ragged=False, missing_mode="none")
        d = {k: v.to(DEV) for k, v in bt.items() if k != "missing"}
        static = torch.stack([d["gen"], d["age"]], 1)
        losses = []
        for it in range(n_steps):
            _, loss = get_trainer(args=args, iteration=it + 1, x=d["x"], static=static, y=d["y"], output_lengths=None,
                                  model=model, logger=_Logger(), device=torch.device(DEV), scheduler=sched,
                                  optimizer=opt, criterion=crit, x_txt=d["txt"], x_img=d["img"],
                                  imgtxt_time=(d["img_time"], d["txt_time"]), scaler=None, missing=bt["missing"],
                                  input_lengths=bt["input_lengths"], txt_lengths=d["txt_lengths"], flow_type="train",
                                  reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
            losses.append(loss)
        return losses

    le, lg = loop(0), loop(1)
    assert all(math.isfinite(x) for x in lg), lg
    REPORT[f"graph_vs_eager[bf16,B64,T1000,overlap={overlap}].final_loss_abs_diff"] = {"rel_err": abs(le[-1] - lg[-1]), "tol": 0.03}
    assert abs(le[-1] - lg[-1]) < 0.03 and lg[-1] < lg[0] - 0.02, (le[::8], lg[::8])


def test_graph_length_buckets_match_trimmed_eager(ops):
    """Graph mode rounds the ragged trim up to a 128-row bucket; the extra pad rows sit behind kv_len, so the
    losses agree with the exactly-trimmed eager steps to rounding (different split of the dW reductions)."""
    lens = [[40, 33, 12, 5], [200, 150, 20, 9], [130, 100, 1, 64], [100, 3, 77, 128], [256, 256, 256, 256]]
    le, pe, _ = _loop(0, 0.0, "fp32", 5, lens, T=300)
    lg, pg, gs = _loop(1, 0.0, "fp32", 5, lens, T=300)
    assert gs.captures == 2 and gs.replays == 3 and not gs.disabled     # buckets 128 and 256, first use of each eager
    err = max(abs(a - b) for a, b in zip(le, lg))
    REPORT["graph_bucketed_vs_trimmed[fp32].loss"] = {"rel_err": err, "tol": 1e-5}
    assert err < 1e-5
    # AdamW normalises every element's update to ~lr, so rounding-level gradient differences on noise-sized
    # gradients (mathematically-zero ones included) move parameters by up to 2*lr per step: 5 steps at lr <= 5e-4
    check("graph_bucketed_vs_trimmed[fp32].params", pg, pe, 5e-3)


def test_graph_cache_is_bounded_and_length_buckets_share_one_pool(ops):
    """VERDICT r4 item 5 / ADVICE r4: captured graphs cannot be released on this ROCm (graph.py), so (i) a trainer captures at most
    --hip-graph-max input shapes and runs further shapes EAGERLY (never evicts and re-captures), (ii) all its captures share one
    memory pool -- N length buckets cost about one step's activations, not N -- and (iii) the growth is reported.  Twelve
    layers (configs[4]'s depth), TIE-len 2000: the loader's batch maximum wanders over all twelve length buckets, largest first."""
    from medical_tri_modal_pilot_amd.builder.trainer.trainer import graph_len_bucket
    T, B, L, CAP = 2000, 4, 12, 5
    maxima = [2000, 1700, 1500, 1200, 1000, 880, 760, 630, 500, 380, 250, 100]
    buckets = [graph_len_bucket(m, T) for m in maxima]
    assert buckets == [2000, 1792, 1536, 1280, 1024, 896, 768, 640, 512, 384, 256, 128] and len(set(buckets)) == 12
    order = [m for m in maxima for _ in range(2)] + maxima          # every shape twice (eager warm-up, capture), then once more
    lens = [[m, max(3, m // 2), max(3, m // 3), 3] for m in order]
    torch.cuda.empty_cache()                                         # (earlier tests' cached blocks are not this trainer's)
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_reserved()
    lg, _, gs = _loop(1, 0.0, "bf16", len(lens), lens, L=L, B=B, T=T, hip_graph_max=CAP)
    peak = torch.cuda.max_memory_reserved() - base
    st = gs.stats()
    assert gs.captures == CAP and st["signatures_captured"] == CAP and not gs.disabled, st
    assert gs.replays == 2 * CAP                                       # the capturing visit (a capture is replayed at once) and the third one
    assert st["eager_over_budget"] == 2 * (12 - CAP), st               # second and third visit of the seven others
    assert all(math.isfinite(v) for v in lg)
    # Against plain eager steps over the ten steps that contain the five captures.  The two runs are two AdamW trajectories of a
    # B = 4 model with a BatchNorm head at lr up to 1e-3: once a length bucket pads the batch differently from the eager trim, a
    # weight-gradient sum is split differently, a gradient element of rounding-noise size changes sign, AdamW moves that weight
    # the other way, and the losses drift apart from there on (measured: the first deviation, ~1e-3, appears at step 4 or 7
    # depending on unrelated reduction orders; 7e-2 two steps later; tools/dbg/cache_drift_dbg.py) -- which says nothing about
    # the cache.  What the cache must not do is change a step: the steps in front of the first differently-padded shape are
    # bit-identical, the later ones stay in the neighbourhood.
    le, _, _ = _loop(0, 0.0, "bf16", 10, lens[:10], L=L, B=B, T=T)
    head_err = max(abs(a - b) for a, b in zip(le[:2], lg[:2]))
    err = max(abs(a - b) for a, b in zip(le, lg[:10]))
    REPORT["graph_cache_budget[bf16,L12,T2000].loss_vs_eager_first_2_steps"] = {"rel_err": head_err, "tol": 0.0}
    REPORT["graph_cache_budget[bf16,L12,T2000].loss_vs_eager_first_10_steps"] = {"rel_err": err, "tol": 0.25}
    assert head_err == 0.0 and err < 0.25, (le, lg[:10])
    grow = [c["reserved_after"] - c["reserved_before"] for c in gs.capture_log]
    first, total = grow[0], sum(grow)
    REPORT["graph_cache_budget[bf16,L12,T2000].reserved_growth_bytes"] = {"rel_err": float(total), "tol": float(2 * max(first, 1))}
    REPORT["graph_cache_budget[bf16,L12,T2000].peak_reserved_bytes_of_this_trainer"] = {"rel_err": float(peak), "tol": 8.0 * 2 ** 30}
    # one pool: the four later (smaller) shapes reuse the blocks of the first capture -- under twice its growth in all
    assert total <= 2 * max(first, 0) + (64 << 20), grow
    assert peak < 8 * 2 ** 30, peak


def test_graph_replay_draws_fresh_dropout_masks(ops):
    """Scalar seeds are frozen in a captured graph; the device step word must still change the masks per replay."""
    from medical_tri_modal_pilot_amd.graph import GraphedTrainStep
    gs = GraphedTrainStep(torch.device(DEV), warmup=0)
    x = torch.randn(256, 256, device=DEV, dtype=torch.bfloat16)
    w = torch.randn(256, 256, device=DEV, dtype=torch.bfloat16)
    outs = []

    def fn(t):
        y = ops.gemm_nt(t["x"], w, drop_p=0.5, seed=99)
        gmask = ops.dropout_bwd(torch.ones_like(y), 99, 0.5)
        outs.append((y, gmask))
        return y.float().sum()

    zero = []
    for _ in range(3):
        gs.run({"x": x}, fn)
        y, gmask = outs[0]                      # static outputs of the captured graph
        zero.append((y == 0).clone())
        assert torch.equal(y == 0, gmask == 0)  # forward mask == regenerated backward mask within a replay
    assert gs.captures == 1 and gs.replays == 3
    assert not torch.equal(zero[0], zero[1]) and not torch.equal(zero[1], zero[2])
    ops.set_seed_word(None)


# ------------------------------------------------------------------ next rows: eval metrics, checkpoints
def test_evaluator_metrics_on_gpu(ops):
    """Evaluator (metrics.py:26-108 of the reference, minus torchmetrics/.cuda()) on device tensors vs scikit-learn."""
    from types import SimpleNamespace
    from sklearn.metrics import average_precision_score, roc_auc_score
    from medical_tri_modal_pilot_amd.builder.utils.metrics import Evaluator
    ev = Evaluator(SimpleNamespace(output_dim=1, batch_size=64, model_types="detection", loss_types="bce",
                                   auxiliary_loss_type="None"))
    g = torch.Generator().manual_seed(3)
    ys, ps = [], []
    for _ in range(40):
        y = (torch.rand(64, generator=g) < 0.25).float()
        p = (torch.sigmoid(torch.randn(64, generator=g) + 1.2 * y) * 50).round() / 50      # ties
        ev.add_batch(y.to(DEV), p.to(DEV))
        ys.append(y)
        ps.append(p)
    auc, apr, f1 = ev.performance_metric()
    yy, pp = torch.cat(ys).numpy(), torch.cat(ps).numpy()
    assert auc == round(roc_auc_score(yy, pp), 4) and apr == round(average_precision_score(yy, pp), 4)
    # F1 (the exact value, not a range): the reference's threshold loop binarises its predictions in place in the
    # first pass (metrics.py:76-83), so what it reports is F1 at threshold 0.01 -- brute force here; the sweep that loop was
    # meant to be (best_f1_over_thresholds) against its own brute force as well
    from medical_tri_modal_pilot_amd.builder.utils.metrics import best_f1_over_thresholds

    def f1_at(th):
        pred = pp >= th
        tp, fp, fn = float((pred & (yy == 1)).sum()), float((pred & (yy == 0)).sum()), float((~pred & (yy == 1)).sum())
        return 2 * tp / (2 * tp + fp + fn) if tp > 0 else 0.0
    assert f1 == round(f1_at(0.01), 4), (f1, f1_at(0.01))
    sweep = float(best_f1_over_thresholds(torch.from_numpy(pp).to(DEV), torch.from_numpy(yy).to(DEV)))
    assert abs(sweep - max(f1_at(k / 100.0) for k in range(1, 100))) < 1e-6


def test_resume_from_reference_optimizer_state(ops):
    """A torch.optim.AdamW(model.parameters()) checkpoint (Logger.save layout) resumes on the fused optimizer: the
    next update equals torch's next update."""
    from medical_tri_modal_pilot_amd.builder.utils import checkpoint as C
    from medical_tri_modal_pilot_amd.optim import FusedAdamW

    def tiny():
        torch.manual_seed(0)
        return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8)).to(DEV)

    x = torch.randn(12, 16, device=DEV)
    ref = tiny()
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=1e-2)
    for _ in range(3):
        ropt.zero_grad()
        ref(x).square().sum().backward()
        ropt.step()
    ckpt = C.make_checkpoint(ref, ropt, 3, 1, 0.5)
    model = tiny()
    fopt = FusedAdamW(list(model.named_parameters()), lr=1.0)
    C.load_checkpoint(ckpt, model, fopt)
    for m, o in ((ref, ropt), (model, fopt)):
        o.zero_grad()
        m(x).square().sum().backward()
        o.step()
    for a, b in zip(model.parameters(), ref.parameters()):
        check("resume.param", a, b, 1e-6)


# ----------------------------------------------------------------------------- packed vital-sign stream (--pack-rows)
def test_row_starts(ops):
    """mtmp_row_starts: exclusive prefix sums of min(max(kv_len, 0), n_max), total last (B below / above one scan block)."""
    for B, n_max in ((1, 7), (64, 1005), (300, 37)):
        g = torch.Generator().manual_seed(B)
        kv = torch.randint(-2, n_max + 9, (B,), generator=g, dtype=torch.int32)
        out = ops.row_starts(kv.to(DEV), n_max).cpu()
        c = kv.clamp(0, n_max).to(torch.int64)
        ref = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(c, 0)])
        assert torch.equal(out[:B + 1].to(torch.int64), ref), (B, out[:5], ref[:5])
        # the attention grids' sample order: a permutation; rank r (by length, ties by index) sits in XCD chunk r % 8, slot r // 8
        order = out[B + 1:].tolist()
        assert sorted(order) == list(range(B))
        ranked = sorted(range(B), key=lambda b: (-int(c[b]), b))
        chunks, pos = [], 0
        for x in range(8):
            n_x = (B - x + 7) // 8 if x < B else 0
            chunks.append(order[pos:pos + n_x])
            pos += n_x
        for x in range(8):
            assert chunks[x] == ranked[x::8], (B, x)
    REPORT["row_starts"] = {"rel_err": 0.0, "tol": 0.0}


def _pack_rows(t, kv, pack):
    """padded [B, N, C] -> the packed layout inside a same-sized buffer (rows behind the live ones poisoned)"""
    B, N, C = t.shape
    out = torch.full((B * N, C), float("nan"), dtype=t.dtype, device=t.device)
    for b in range(B):
        out[int(pack[b]):int(pack[b]) + int(kv[b])] = t[b, :int(kv[b])]
    return out.view(B, N, C)


def _unpack_rows(t, kv, pack):
    B, N, C = t.shape
    out = torch.zeros(B, N, C, dtype=t.dtype, device=t.device)
    flat = t.reshape(B * N, C)
    for b in range(B):
        out[b, :int(kv[b])] = flat[int(pack[b]):int(pack[b]) + int(kv[b])]
    return out


@pytest.mark.parametrize("bounded", [False, True])
@pytest.mark.parametrize("N,lens", [(200, [200, 5, 64, 129, 1, 77]), (1005, [1005, 6, 700, 333]), (133, [5, 133, 64, 65, 1, 128, 129, 2, 97]),
                                    (600, [257, 256, 255, 513, 512, 600])])
def test_packed_attention_fp32_vs_oracle(ops, N, lens, bounded):
    """The row_start addressing of the attention kernels in the fp32 PARITY build, against the oracle directly (the
    packed layout used to meet the 1e-4 gate only through a chain -- fp32 padded == golden, bf16 packed == bf16 padded): forward
    (O, O + residual, and the LSE through the backward), dQ and dK / dV at the 1e-4 gate on samples stored back to back, lengths on
    both sides of the 64-key tile and 256-query item boundaries, every row behind the live ones poisoned with NaN.  The oracle runs
    per sample on its own rows: a packed stream has no pad rows, so a sample's queries are its kv_len rows."""
    dt = torch.float32
    g = torch.Generator().manual_seed(1000 + N)
    B = len(lens)
    qkv = torch.randn(B, N, 768, generator=g)
    res = torch.randn(B, N, 256, generator=g)
    w = torch.randn(B, N, 256, generator=g)
    kv = torch.tensor(lens, dtype=torch.int32, device=DEV)
    pack = ops.row_starts(kv, N)
    pk = pack.cpu()
    qd, rd, wd = (_pack_rows(t.to(DEV), lens, pk) for t in (qkv, res, w))
    kn = None
    if bounded:
        kn = ops.key_norms(torch.nan_to_num(qd))                    # (the table is taken over the buffer's rows; poison would poison it)
    o, o_res, lse = ops.attn_fwd_grouped([qd], [kv], [rd], [kn], [pack])
    o_u, or_u = _unpack_rows(o[0], lens, pk).cpu(), _unpack_rows(o_res[0], lens, pk).cpu()
    dqkv = ops.attn_bwd_grouped([qd], o, [torch.nan_to_num(wd)], lse, [kv], [pack])[0]
    dq_u = _unpack_rows(dqkv, lens, pk).cpu()
    worst = {"o": 0.0, "o_res": 0.0, "dqkv": 0.0}
    for b, n in enumerate(lens):
        q_ref = qkv[b:b + 1, :n].clone().requires_grad_()
        o_ref = O.attention_core(q_ref, None)
        (o_ref * w[b:b + 1, :n]).sum().backward()
        rel = lambda a, r: float((a - r).abs().max() / r.abs().max().clamp_min(1e-30))
        worst["o"] = max(worst["o"], rel(o_u[b:b + 1, :n], o_ref.detach()))
        worst["o_res"] = max(worst["o_res"], rel(or_u[b:b + 1, :n], o_ref.detach() + res[b:b + 1, :n]))
        worst["dqkv"] = max(worst["dqkv"], rel(dq_u[b:b + 1, :n], q_ref.grad))
    tag = f"attn_packed[fp32,N={N},B={B}{',bounded' if bounded else ''}]"
    for k, v in worst.items():
        REPORT[f"{tag}.{k}"] = {"rel_err": v, "tol": 1e-4}
        assert v < 1e-4, (tag, k, v)


@pytest.mark.parametrize("N,lens", [(1005, [1005, 6, 700, 333, 257, 256, 512, 513, 64, 65, 769, 1]),
                                    (2005, [2005, 1290, 255, 1025])])
def test_packed_attention_bf16_long_streams_vs_oracle(ops, N, lens):
    """The bf16 build's long-stream backward (64 rows per wave, accumulators in the AGPR file: attention.hip `attn_bwd_dkdv64_kernel`,
    `attn_bwd_dq64_kernel`) on a PACKED stream against the oracle: lengths on both sides of the 64-row tile, the 256-row workgroup
    and the 512-row dispatch boundaries (waves and whole workgroups without a live row), dead rows poisoned with NaN."""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(2000 + N)
    B = len(lens)
    qkv = torch.randn(B, N, 768, generator=g).to(dt).float()
    res = torch.randn(B, N, 256, generator=g).to(dt).float()
    w = torch.randn(B, N, 256, generator=g).to(dt).float()
    kv = torch.tensor(lens, dtype=torch.int32, device=DEV)
    pack = ops.row_starts(kv, N)
    pk = pack.cpu()
    qd, rd, wd = (_pack_rows(t.to(DEV, dt), lens, pk) for t in (qkv, res, w))
    kn = ops.key_norms(torch.nan_to_num(qd))
    o, o_res, lse = ops.attn_fwd_grouped([qd], [kv], [rd], [kn], [pack])
    dqkv = ops.attn_bwd_grouped([qd], o, [torch.nan_to_num(wd)], lse, [kv], [pack])[0]
    o_u = _unpack_rows(o[0], lens, pk).float().cpu()
    dq_u = _unpack_rows(dqkv, lens, pk).float().cpu()
    # errors are taken against the largest reference entry of the whole batch: a sample of one row has dq = dk = 0 exactly, and
    # what the bf16 kernels leave there (the rounding of O inside delta) is noise on the scale of the other samples' gradients
    err = {"o": 0.0, "dq": 0.0, "dk": 0.0, "dv": 0.0}
    top = {"o": 0.0, "dq": 0.0, "dk": 0.0, "dv": 0.0}
    for b, n in enumerate(lens):
        q_ref = qkv[b:b + 1, :n].clone().requires_grad_()
        o_ref = O.attention_core(q_ref, None)
        (o_ref * w[b:b + 1, :n]).sum().backward()
        err["o"] = max(err["o"], float((o_u[b:b + 1, :n] - o_ref.detach()).abs().max()))
        top["o"] = max(top["o"], float(o_ref.abs().max()))
        for i, nm in enumerate(("dq", "dk", "dv")):
            ref = q_ref.grad[..., 256 * i:256 * (i + 1)]
            err[nm] = max(err[nm], float((dq_u[b:b + 1, :n, 256 * i:256 * (i + 1)] - ref).abs().max()))
            top[nm] = max(top[nm], float(ref.abs().max()))
    worst = {k: err[k] / top[k] for k in err}
    tag = f"attn_packed[bf16,N={N},B={B}]"
    for k, v in worst.items():
        REPORT[f"{tag}.{k}"] = {"rel_err": v, "tol": 4e-2}
        assert v < 4e-2, (tag, k, v)



def test_packed_layer_fp32_vs_golden_and_oracle(ops):
    """One encoder layer on a PACKED stream in the fp32 parity build ("a packed layer meets the layer golden at 1e-4
    directly"): the layer, inputs and key lengths of test_encoder_layer_vs_golden, the samples' valid rows back to back, every
    row behind them NaN (ops.layer_forward(pack=...): row kernels on the live rows, attention through row_start).  The output's
    valid rows against the golden of the REAL reference layer; input and parameter gradients against the oracle's layer under
    autograd with the loss restricted to the valid rows (a packed stream has no pad query rows, the golden's loss has)."""
    from medical_tri_modal_pilot_amd.builder.models.src.transformer.encoder import TransformerEncoderLayer
    Gd = G("blocks")
    lay = TransformerEncoderLayer(256, 4, 1024, 0.0)
    lay.load_state_dict({k: filler.fill_tensor("g3." + k, v) for k, v in lay.state_dict().items()})
    g = torch.Generator().manual_seed(11)
    for N in (54, 133, 261):
        torch.randn(4, N, 256, generator=g); torch.randn(4, N, 256, generator=g)
    torch.randn(2, 40, 256, generator=g)
    torch.randn(3, 17, 256, generator=g); torch.randn(3, 17, 256, generator=g)
    x = torch.randn(3, 70, 256, generator=g)
    w = torch.randn(3, 70, 256, generator=g)
    lens = [int(v) for v in Gd["lay_len"]]
    B, N = 3, 70
    valid = (torch.arange(N)[None, :] < torch.tensor(lens)[:, None])
    # oracle, loss over the valid rows
    sd = {"L." + k: v.detach().clone().requires_grad_() for k, v in lay.state_dict().items()}
    x_ref = x.clone().requires_grad_()
    y_ref = O.encoder_layer(sd, "L", x_ref, O.key_pad_mask(N, torch.tensor(lens)), 4)
    (y_ref * w * valid[..., None]).sum().backward()
    # product, packed
    lay = lay.to(DEV)
    kv = torch.tensor(lens, dtype=torch.int32, device=DEV)
    pack = ops.row_starts(kv, N)
    pk = pack.cpu()
    P = lay.param_list()
    fused = type(lay).fused_weights_of([lay], torch.float32)[0]
    y_pk, saved = ops.layer_forward(_pack_rows(x.to(DEV), lens, pk), kv, P, fused, 0.0, (0, 0), pack=pack)
    dz_pk, grads = ops.layer_backward(saved, torch.nan_to_num(_pack_rows((w * valid[..., None]).to(DEV), lens, pk)).contiguous())
    y = _unpack_rows(y_pk, lens, pk).cpu()
    dz = _unpack_rows(dz_pk, lens, pk).cpu()
    gold = torch.from_numpy(Gd["lay_y"])                       # rows ::5 of the real layer's output
    m5 = valid[:, ::5]
    check("packed_layer[fp32].y_vs_golden", y[:, ::5][m5], gold[m5], 1e-4)
    check("packed_layer[fp32].y_vs_oracle", y[valid], y_ref.detach()[valid], 1e-4)
    check("packed_layer[fp32].dz", dz[valid], x_ref.grad[valid], 1e-4)
    names = ["attention_prenorm.gamma", "attention_prenorm.beta", "self_attention.query_proj.linear.weight",
             "self_attention.query_proj.linear.bias", "self_attention.key_proj.linear.weight", "self_attention.key_proj.linear.bias",
             "self_attention.value_proj.linear.weight", "self_attention.value_proj.linear.bias", "feed_forward_prenorm.gamma",
             "feed_forward_prenorm.beta", "feed_forward.w_1.weight", "feed_forward.w_1.bias", "feed_forward.w_2.weight",
             "feed_forward.w_2.bias"]
    for nme, got in zip(names, grads):
        ref = sd["L." + nme].grad
        assert torch.isfinite(got).all(), nme
        if nme == "self_attention.key_proj.linear.bias":
            # softmax does not see a constant added to every key: this gradient is exactly zero in real arithmetic and pure
            # rounding noise in both computations -- bounded on the scale of the query bias' gradient
            scale = float(sd["L.self_attention.query_proj.linear.bias"].grad.abs().max())
            e = float((got.cpu() - ref).abs().max()) / scale
            REPORT[f"packed_layer[fp32].d[{nme}]"] = {"rel_err": e, "tol": 1e-4}
            assert e < 1e-4, e
            continue
        check(f"packed_layer[fp32].d[{nme}]", got.reshape(ref.shape).cpu(), ref, 1e-4)


@pytest.mark.parametrize("N,lens", [(200, [200, 5, 64, 129, 1, 77]), (1005, [1005, 6, 700, 333]), (300, [300] * 3), (70, [1]),
                                    (133, [5, 133, 64, 65, 1, 128, 129, 2, 97]), (64, [1] * 11)])
def test_packed_layer_equals_padded_layer(ops, N, lens):
    """One encoder layer of the grouped bf16 kernels on a PACKED stream (valid rows back to back, every row behind them NaN) against
    the same layer on the padded [B, N] layout: outputs and input gradients on the valid rows, parameter gradients.  The row-panel
    kernels are row-local (bit-identical rows); the attention forward may pick the other softmax body for a wave and the weight
    gradients cut their token range elsewhere, so the comparison is to bf16 rounding."""
    from medical_tri_modal_pilot_amd.builder.models.src.transformer.encoder import TransformerEncoderLayer
    torch.manual_seed(3)
    B = len(lens)
    kv = torch.tensor(lens, dtype=torch.int32, device=DEV)
    layer = TransformerEncoderLayer(d_model=256, num_heads=4, d_ff=1024, dropout_p=0.0).to(DEV)
    with torch.no_grad():
        for prm in layer.parameters():
            if prm.dim() > 1:
                prm.mul_(3.0)
    P = layer.param_list()
    fused = type(layer).fused_weights_of([layer], torch.bfloat16)[0]
    z = torch.randn(B, N, 256, device=DEV).to(torch.bfloat16)
    valid = (torch.arange(N, device=DEV)[None, :] < kv[:, None])
    d_out = torch.randn(B, N, 256, device=DEV).to(torch.bfloat16) * valid[..., None]      # a pad row has no gradient
    pack = ops.row_starts(kv, N)
    outs, grads = {}, {}
    for mode in ("padded", "packed"):
        pk = [pack] if mode == "packed" else None
        zin = _pack_rows(z, lens, pack.cpu()) if mode == "packed" else z
        dout = _pack_rows(d_out, lens, pack.cpu()) if mode == "packed" else d_out
        (y,), saved = ops.layer_forward_grouped([zin], [kv], [P], [fused], 0.0, [(0, 0)], pk)
        (dz,), (g,) = ops.layer_backward_grouped(saved, [dout.contiguous()], [None], None)
        if mode == "packed":
            y, dz = _unpack_rows(y, lens, pack.cpu()), _unpack_rows(dz, lens, pack.cpu())
        outs[mode] = (y * valid[..., None], dz * valid[..., None])
        grads[mode] = g
    tag = f"packed_layer[N={N},B={B}]"
    check(tag + ".y", outs["packed"][0].float(), outs["padded"][0].float(), 1e-2)
    check(tag + ".dz", outs["packed"][1].float(), outs["padded"][1].float(), 2e-2)
    for k, (a, b) in enumerate(zip(grads["packed"], grads["padded"])):
        assert torch.isfinite(a).all(), (tag, k)
        if k in (4, 5):                  # key-projection weight / bias gradients: mathematically ~0 (softmax shift invariance), noise
            continue
        check(tag + f".grad{k}", a.float(), b.float(), 2e-2)


@pytest.mark.parametrize("Ns,lens", [((54, 133), [None, [133, 5, 64, 1, 90]]), ((20,), [None]), ((54, 133), [None, None])])
def test_layer_with_ffn_on_read_rows_only(ops, Ns, lens):
    """layer_forward_grouped(ffn_rows=4) -- the image / text streams in the last layer they run in, whose outputs feed the
    bottleneck exchange (rows 0..3) and nothing else -- against the dense layer: output rows 0..3, and with an output gradient
    that lives in those rows only (what the exchange backward hands over) the same input gradient and parameter gradients."""
    from medical_tri_modal_pilot_amd.builder.models.src.transformer.encoder import TransformerEncoderLayer
    torch.manual_seed(5)
    B, n = 5, len(Ns)
    layers = [TransformerEncoderLayer(d_model=256, num_heads=4, d_ff=1024, dropout_p=0.0).to(DEV) for _ in Ns]
    with torch.no_grad():
        for layer in layers:
            for prm in layer.parameters():
                if prm.dim() > 1:
                    prm.mul_(3.0)
    Ps = [layer.param_list() for layer in layers]
    fused = type(layers[0]).fused_weights_of(layers, torch.bfloat16)
    zs = [torch.randn(B, N, 256, device=DEV).to(torch.bfloat16) for N in Ns]
    kvs = [None if ln is None else torch.tensor(ln, dtype=torch.int32, device=DEV) for ln in lens]
    d_outs = []
    for N in Ns:
        d = torch.zeros(B, N, 256, device=DEV, dtype=torch.bfloat16)
        d[:, :4] = torch.randn(B, 4, 256, device=DEV).to(torch.bfloat16)
        d_outs.append(d)
    res = {}
    for R in (None, 4):
        ys, saved = ops.layer_forward_grouped(list(zs), kvs, Ps, fused, 0.0, [(0, 0)] * n, None, ffn_rows=R)
        dzs, gs = ops.layer_backward_grouped(saved, [d.clone() for d in d_outs], [None] * n, None)
        res[R] = ([y[:, :4].float() for y in ys], [d.float() for d in dzs], gs)
    for i in range(n):
        tag = f"ffn_rows[{Ns},{i}]"
        check(tag + ".y", res[4][0][i], res[None][0][i], 1e-2)
        check(tag + ".dz", res[4][1][i], res[None][1][i], 2e-2)
        for k, (a, b) in enumerate(zip(res[4][2][i], res[None][2][i])):
            assert torch.isfinite(a).all(), (tag, k)
            if k in (4, 5):
                continue
            check(tag + f".grad{k}", a.float(), b.float(), 2e-2)


def test_joint_embedding_node_equals_separate_nodes(ops, monkeypatch):
    """ops.TieTimeEmbed (event embedding + image / text time embeddings as one autograd node whose backward is ONE launch over the
    events and the image / text times and one mtmp_reduce_scatter into the flat gradient) against ops.TieEmbed + ops.TimeEmbed
    through autograd's own accumulation: the shared ie_time / ie_feat gradients are summed in another order (one slab instead of
    two sums added afterwards), so losses and parameters agree to rounding, not bit for bit (bf16 replayed, fp32 eager; three
    AdamW steps at lr <= 1e-4: an element whose gradient is rounding noise may move by 2 lr either way)."""
    import importlib
    cls = importlib.import_module("medical_tri_modal_pilot_amd.builder.models.8_missing_models.tri_mbt_vsltcls").TRI_MBT_VSLTCLS
    lens = [[96, 50, 7, 1], [96, 96, 96, 96], [3, 96, 20, 64]]
    res = {}
    for joint in (True, False):
        monkeypatch.setattr(cls, "joint_embeddings", joint)
        for graph, dtype in ((1, "bf16"), (0, "fp32")):
            res[joint, graph, dtype] = _loop(graph, 0.0, dtype, 3, lens)[:2]
    # Parameters whose gradient is mathematically zero hold rounding noise on both sides (the key projections' biases: softmax is
    # shift invariant; biases in front of the BatchNorm head -- DESIGN section 2): its sign, and with it AdamW's move of up to
    # lr per step, follows any change of summation order.  Those may differ by 2 sum(lr) = 1.2e-3 after three steps; everything
    # else agrees to rounding.
    noise = ("key_proj.linear.bias", "fc_list.0.bias", "layer_norms_after_concat.bias", "ie_demo.1.bias", "ie_demo.0.bias")
    worst_l = worst_p = worst_noise = 0.0
    for graph, dtype in ((1, "bf16"), (0, "fp32")):
        a, b = res[True, graph, dtype], res[False, graph, dtype]
        worst_l = max(worst_l, max(abs(x - y) for x, y in zip(a[0], b[0])))
        d = (a[1] - b[1]).abs()
        lay = _loop.last_layout + [("end", d.numel())]
        for (n, o), (_, o2) in zip(lay[:-1], lay[1:]):
            m = float(d[o:o2].max())
            if n.endswith(noise):
                worst_noise = max(worst_noise, m)
            else:
                worst_p = max(worst_p, m)
    REPORT["joint_embedding_node.loss"] = {"rel_err": worst_l, "tol": 1e-5}
    REPORT["joint_embedding_node.params"] = {"rel_err": worst_p, "tol": 2e-5}
    REPORT["joint_embedding_node.params_with_zero_gradient"] = {"rel_err": worst_noise, "tol": 1.2e-3}
    assert worst_l < 1e-5 and worst_p < 2e-5 and worst_noise < 1.2e-3, (worst_l, worst_p, worst_noise)


def test_packed_training_steps_equal_padded_steps(ops):
    """--pack-rows 1 against --pack-rows 0 through get_trainer (bf16, dropout 0, ragged batches with mixed missing modalities,
    eager and hipGraph replay with the lengths changing under one captured graph): same losses to bf16 rounding, parameters after
    four AdamW steps within 2 lr."""
    lens = [[96, 50, 7, 1], [96, 96, 96, 96], [3, 96, 20, 64], [96, 1, 1, 2]]
    res = {}
    for pack in (0, 1):
        for graph in (0, 1):
            res[pack, graph] = _loop(graph, 0.0, "bf16", 4, lens, pack_rows=pack)
            assert _loop.last_packed == bool(pack)      # the mode under test really ran
    gs = res[1, 1][2]
    assert gs is not None and gs.captures == 1 and gs.replays == 3 and not gs.disabled
    for graph in (0, 1):
        lp, l0 = res[1, graph][0], res[0, graph][0]
        worst = max(abs(a - b) for a, b in zip(lp, l0))
        REPORT[f"packed_vs_padded[graph={graph}].loss"] = {"rel_err": worst, "tol": 2e-3}
        assert worst < 2e-3, (lp, l0)
        dp = float((res[1, graph][1] - res[0, graph][1]).abs().max())
        REPORT[f"packed_vs_padded[graph={graph}].params"] = {"rel_err": dp, "tol": 8e-4}
        assert dp < 8e-4, dp                      # lr <= 1e-4: AdamW moves a parameter by at most ~lr per step
    assert res[1, 0][0] == res[1, 1][0], (res[1, 0][0], res[1, 1][0])       # replay == eager, packed
    assert torch.equal(res[1, 0][1], res[1, 1][1])


# ----------------------------------------------------------------------------- frozen encoder on the present images only
def test_image_slots(ops):
    """mtmp_image_slots: present images first (batch order), then the others; image -> slot (B = the zero slot); live rows per
    (part, stage)."""
    for B, hw0 in ((1, 3136), (64, 3136), (300, 64)):
        g = torch.Generator().manual_seed(B)
        pres = torch.rand(B, generator=g) < 0.6
        out = ops.image_slots(torch.where(pres, 1, 3).to(DEV), 2, hw0).cpu().tolist()
        live = [b for b in range(B) if pres[b]]
        dead = [b for b in range(B) if not pres[b]]
        assert out[:B] == live + dead
        assert out[B:2 * B] == [live.index(b) if pres[b] else B for b in range(B)]
        assert out[2 * B] == len(live)
        half = B // 2
        for p, n in enumerate((len(live), min(len(live), half), max(len(live) - half, 0))):
            assert out[2 * B + 1 + 5 * p:2 * B + 6 + 5 * p] == [n * (hw0 >> (2 * s)) for s in range(4)] + [n], (B, p)
    REPORT["image_slots"] = {"rel_err": 0.0, "tol": 0.0}


@pytest.mark.parametrize("split", [False, True])
def test_swin_encodes_present_images_only(ops, split):
    """SwinTransformer.forward(slots=...): the features of the samples that have an image are bit-identical to the full-batch
    forward (every kernel of the encoder is local to an image), the others come back as zeros; with and without the two-stream tail."""
    _, model = _product_model(2, 0, "bf16")
    enc = model.img_encoder.eval()
    B = 16
    g = torch.Generator().manual_seed(5)
    img = torch.rand(B, 1, 224, 224, generator=g).to(DEV)
    pres = torch.tensor([1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 1, 0, 1, 1, 0, 1], dtype=torch.bool, device=DEV)
    tails = (torch.cuda.Stream(), torch.cuda.Stream()) if split else None

    def run(slots):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            f = enc(img, tail_streams=tails, slots=slots)
            if tails is not None:
                side.wait_stream(tails[0])
        torch.cuda.synchronize()
        return f
    full = run(None)
    part = run(ops.image_slots(torch.where(pres, 0, 2), 2, 56 * 56))
    assert part.shape == full.shape
    assert torch.equal(part[pres], full[pres])
    assert float(part[~pres].float().abs().max()) == 0.0
    none = run(ops.image_slots(torch.full((B,), 3, dtype=torch.int64, device=DEV), 2, 56 * 56))
    assert float(none.float().abs().max()) == 0.0
    REPORT[f"swin_present_only[split={int(split)}]"] = {"rel_err": 0.0, "tol": 0.0}


def test_training_steps_do_not_depend_on_features_of_missing_images(ops):
    """--skip-missing-images 1 against 0 through get_trainer (bf16, mixed missing modalities, eager and hipGraph replay): the
    image stream of a sample without an image feeds nothing (exchange weight 0, zero gradient), so losses and parameters are
    bit-identical whether its features are Swin(zero image) or zeros."""
    lens = [[96, 50, 7, 1], [96, 96, 96, 96], [3, 96, 20, 64], [96, 1, 1, 2]]
    res = {(sk, gr): _loop(gr, 0.0, "bf16", 4, lens, skip_missing_images=sk) for sk in (0, 1) for gr in (0, 1)}
    for gr in (0, 1):
        assert res[1, gr][0] == res[0, gr][0], (res[1, gr][0], res[0, gr][0])
        assert torch.equal(res[1, gr][1], res[0, gr][1])
    assert res[1, 1][2].captures == 1 and res[1, 1][2].replays == 3
    REPORT["skip_missing_images.steps"] = {"rel_err": 0.0, "tol": 0.0}


def test_packed_stream_under_staged_graphs_and_packed_events(ops):
    """The packed vital-sign stream (bf16) in the other step forms: (i) the step cut into two / three chained graphs as under
    DDP -- the packed buffers and the row map cross the cut; bit-identical to the one-graph step; (ii) eval-mode forward
    (flow_type "test", no autograd graph): same logits packed and padded."""
    lens = [[96, 50, 7, 1], [96, 96, 96, 96], [3, 96, 20, 64], [96, 1, 1, 2]]
    l1, p1, _ = _loop(1, 0.0, "bf16", 4, lens, L=4, graph_stages=1)
    assert _loop.last_packed
    for stages in (2, 3):
        ls, ps, gs = _loop(1, 0.0, "bf16", 4, lens, L=4, graph_stages=stages)
        assert _loop.last_packed and [len(e["graphs"]) for e in gs.entries.values()] == [stages]
        assert ls == l1, (stages, ls, l1)
        assert torch.equal(ps, p1)
    REPORT["packed_staged_graphs_vs_single[bf16]"] = {"rel_err": 0.0, "tol": 0.0}
    # (ii) eval forward, packed against padded
    outs = {}
    for pack in (0, 1):
        torch.manual_seed(7)
        args, model = _product_model(2, 0, "bf16", pack_rows=pack)
        model.eval()
        bt = filler.make_batch(901, 4, 96, ragged=True, missing_mode="mixed")
        bt["input_lengths"] = torch.tensor([96, 50, 7, 1])
        dv = lambda t: t.to(DEV)
        with torch.no_grad():
            out, _, _ = model(dv(bt["x"]), None, None, None, None, dv(bt["age"]), dv(bt["gen"]), dv(bt["input_lengths"]),
                              dv(bt["txt"]), dv(bt["txt_lengths"]), dv(bt["img"]), dv(bt["missing_num"]), None,
                              dv(bt["img_time"].half().float()), dv(bt["txt_time"].half().float()), "test", None, None)
        assert (model.fusion_transformer.last_pack is not None) == bool(pack)
        outs[pack] = out.float()
    check("packed_eval_logits[bf16]", outs[1], outs[0], 1e-3)


def test_packed_steps_with_empty_windows_and_four_images(ops):
    """Edge cases of the packed vital-sign stream through get_trainer (bf16, eager + graph): samples with NO event at all (their
    stream is bottleneck prefix + CLS, five rows), a batch of all-empty windows, and --multiimages 1 with K = 4 (the image
    stream carries its own key lengths): losses as with --pack-rows 0."""
    lens = [[0, 50, 0, 1], [0, 0, 0, 0], [96, 0, 20, 64]]
    for over in (dict(), dict(multiimages=1, n_images=4)):
        res = {}
        for pack in (0, 1):
            for graph in (0, 1):
                if over:
                    args_over = dict(over)
                    ls, ps, _ = _loop_multi(graph, lens, pack, **args_over)
                else:
                    ls, ps, _ = _loop(graph, 0.0, "bf16", 3, lens, pack_rows=pack)
                    assert _loop.last_packed == bool(pack)
                res[pack, graph] = (ls, ps)
        for graph in (0, 1):
            worst = max(abs(a - b) for a, b in zip(res[1, graph][0], res[0, graph][0]))
            REPORT[f"packed_edge_cases[{'K4' if over else 'empty'},graph={graph}].loss"] = {"rel_err": worst, "tol": 2e-3}
            assert all(math.isfinite(v) for v in res[1, graph][0]) and worst < 2e-3, (res[1, graph][0], res[0, graph][0])
            assert float((res[1, graph][1] - res[0, graph][1]).abs().max()) < 8e-4
        assert res[1, 0][0] == res[1, 1][0]


def test_multi_image_steps_do_not_depend_on_images_behind_the_key_length(ops):
    """--multiimages 1, K = 4 with absent images: the images at positions >= the stream's image count are masked as keys and
    not encoded (--skip-missing-images 1); losses and parameters bit-identical to encoding them."""
    lens = [[96, 50, 7, 1], [96, 96, 96, 96], [3, 96, 20, 64]]
    res = {(sk, gr): _loop_multi(gr, lens, 1, n_images=4, skip_missing_images=sk)[:2] for sk in (0, 1) for gr in (0, 1)}
    for gr in (0, 1):
        assert res[1, gr][0] == res[0, gr][0], (res[1, gr][0], res[0, gr][0])
        assert torch.equal(res[1, gr][1], res[0, gr][1])
    REPORT["skip_missing_images.multi_image_steps"] = {"rel_err": 0.0, "tol": 0.0}


def _loop_multi(hip_graph, lens_per_step, pack, **over):
    """_loop for --multiimages 1 (the synthetic batch then carries K images per sample)"""
    from medical_tri_modal_pilot_amd.builder.trainer import get_trainer
    from medical_tri_modal_pilot_amd.builder.utils.cosine_annealing_with_warmup_v2 import CosineAnnealingWarmupRestarts
    from medical_tri_modal_pilot_amd.optim import FusedAdamW
    torch.manual_seed(7)
    K = over.get("n_images", 3)
    args, model = _product_model(2, 1, "bf16", hip_graph=hip_graph, dropout=0.0, pack_rows=pack, n_images=K,
                                 skip_missing_images=over.get("skip_missing_images", 1))
    model.train()
    model.img_encoder.eval()
    opt = FusedAdamW(model.hot_parameters(), lr=1e-4, weight_decay=args.weight_decay)
    sched = CosineAnnealingWarmupRestarts(opt, first_cycle_steps=100, cycle_mult=1, max_lr=1e-3, min_lr=1e-6, warmup_steps=10, gamma=1.0)
    crit = torch.nn.BCEWithLogitsLoss(reduction="mean")
    losses = []
    for it, lens in enumerate(lens_per_step):
        bt = filler.make_batch(900 + it, 4, 96, ragged=True, missing_mode="mixed" if it % 2 else "none", multiimages=1, n_images=K)
        bt["input_lengths"] = torch.tensor(lens)
        static = torch.stack([bt["gen"], bt["age"]], 1)
        _, loss = get_trainer(args=args, iteration=it + 1, x=bt["x"], static=static, y=bt["y"], output_lengths=None, model=model,
                              logger=_Logger(), device=torch.device(DEV), scheduler=sched, optimizer=opt, criterion=crit,
                              x_txt=bt["txt"], x_img=bt["img"], imgtxt_time=(bt["img_time"], bt["txt_time"]), scaler=None,
                              missing=bt["missing"], input_lengths=bt["input_lengths"], txt_lengths=bt["txt_lengths"],
                              flow_type="train", reports_tokens=None, reports_lengths=None, criterion_aux=(None, None))
        losses.append(loss)
    assert (model.fusion_transformer.last_pack is not None) == bool(pack)
    return losses, opt.flat.data.detach().clone(), getattr(model, "_mtmp_graph_step", None)


def test_packed_full_size_steps_match_padded(ops):
    """The packed stream at the benchmark's size (B 64, T 1000, 6 layers, bf16, hipGraph replay, ragged lengths that change from
    step to step under one captured graph, mixed missing modalities): the grouped LDS-DMA weight-gradient plan, eight XCD chunks,
    503-block grids with half the blocks idle.  Losses as with --pack-rows 0 --skip-missing-images 0."""
    g = torch.Generator().manual_seed(4)
    lens = [torch.randint(3, 1001, (64,), generator=g).tolist() for _ in range(3)]
    lens[1][0] = 1000
    res = {}
    for mode in (0, 1):
        ls, ps, gs = _loop(1, 0.0, "bf16", 3, lens, L=6, B=64, T=1000, batch_size=64, pack_rows=mode, skip_missing_images=mode)
        assert _loop.last_packed == bool(mode) and gs.captures == 1 and gs.replays == 2
        res[mode] = (ls, ps)
    worst = max(abs(a - b) for a, b in zip(res[1][0], res[0][0]))
    REPORT["packed_full_size[bf16,B64,T1000].loss"] = {"rel_err": worst, "tol": 2e-3}
    assert all(math.isfinite(v) for v in res[1][0]) and worst < 2e-3, (res[1][0], res[0][0])
    dp = float((res[1][1] - res[0][1]).abs().max())
    REPORT["packed_full_size[bf16,B64,T1000].params"] = {"rel_err": dp, "tol": 6e-4}
    assert dp < 6e-4, dp


@pytest.mark.parametrize("over", [dict(mbt_only_vslt=0), dict(residual_bottlenecks=1), dict(mbt_fusion_startIdx=1)],
                         ids=["vsltonly0", "resbottle", "fusion_start1"])
def test_packed_stream_with_other_encoder_flags(ops, over):
    """--pack-rows 1 against 0 (bf16, eager + graph, ragged lengths) with the encoder's other switches: all three streams in the
    last layer (--mbt-only-vslt 0), residual bottlenecks, and uni-modal layers in front of the fusion layers
    (--mbt-fusion-startIdx 1: the stream inputs are not fused there, so the stream stays padded -- the flag must be harmless)."""
    lens = [[96, 50, 7, 1], [3, 96, 20, 64], [96, 1, 1, 2]]
    res = {}
    for pack in (0, 1):
        for graph in (0, 1):
            res[pack, graph] = _loop(graph, 0.0, "bf16", 3, lens, pack_rows=pack, **over)[:2]
            assert _loop.last_packed == (bool(pack) and "mbt_fusion_startIdx" not in over)
    tag = next(iter(over))
    for graph in (0, 1):
        worst = max(abs(a - b) for a, b in zip(res[1, graph][0], res[0, graph][0]))
        REPORT[f"packed_flags[{tag},graph={graph}].loss"] = {"rel_err": worst, "tol": 2e-3}
        assert all(math.isfinite(v) for v in res[1, graph][0]) and worst < 2e-3, (res[1, graph][0], res[0, graph][0])
    assert res[1, 0][0] == res[1, 1][0]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("packed", [False, True])
def test_attn_cls_row_vs_dense_attention(ops, dt, packed):
    """mtmp_attn_cls_fwd / _bwd (the CLS query of the last layer alone) against the oracle's dense attention: output and
    output + residual of row cls_tok, and -- for a gradient that is zero outside that row -- the dense dq / dk / dv of
    O.attention_core's autograd; padded and packed layouts, ragged key lengths incl. a sample whose only rows are the prefix."""
    torch.manual_seed(5)
    B, N, cls_tok = 6, 333, 4
    lens = [333, 5, 64, 129, 300, 65]
    kv = torch.tensor(lens, dtype=torch.int32, device=DEV)
    qkv32 = torch.randn(B, N, 768, device=DEV) * 0.7
    z32 = torch.randn(B, N, 256, device=DEV)
    d32 = torch.randn(B, 256, device=DEV)
    qkv, z, d_o = qkv32.to(dt), z32.to(dt), d32.to(dt)
    # oracle: dense attention (keys masked behind kv), row cls_tok
    qr = qkv.float().clone().requires_grad_(True)
    q, k, v = (t.reshape(B, N, 4, 64).permute(0, 2, 1, 3) for t in qr.split(256, dim=2))
    mask = (torch.arange(N, device=DEV)[None, :] >= kv[:, None])[:, None, None, :]
    sc = (q @ k.transpose(-1, -2)) / 8.0
    sc = sc.masked_fill(mask, -65504.0)
    o_ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, 256)[:, cls_tok]
    (o_ref * d_o.float()).sum().backward()
    g_ref = qr.grad.clone()
    valid = (torch.arange(N, device=DEV)[None, :] < kv[:, None])[..., None]
    pack = ops.row_starts(kv, N) if packed else None
    if packed:
        qkv_in, z_in = _pack_rows(qkv, lens, pack.cpu()), _pack_rows(z, lens, pack.cpu())
    else:
        qkv_in, z_in = qkv, z
    o, r1, lse = ops.attn_cls_fwd(qkv_in.contiguous(), z_in.contiguous(), kv, pack, cls_tok)
    tol = 1e-4 if dt == torch.float32 else 1e-2
    tag = f"attn_cls[{str(dt).replace('torch.', '')},packed={int(packed)}]"
    check(tag + ".o", o.float(), o_ref, tol)
    check(tag + ".r1", r1.float(), o_ref.to(dt).float() + z.float()[:, cls_tok], tol)
    dqkv = ops.attn_cls_bwd(qkv_in.contiguous(), o, d_o, lse, kv, pack, cls_tok)
    if packed:
        dqkv = _unpack_rows(dqkv, lens, pack.cpu())
    assert torch.isfinite(dqkv.float()[valid.expand(-1, -1, 768)]).all()
    assert float((dqkv.float() * (~valid)).abs().max()) == 0.0            # rows behind kv_len: zeros
    check(tag + ".dqkv", dqkv.float() * valid, g_ref * valid, 2e-4 if dt == torch.float32 else 2e-2)
    nz = dqkv[:, :, :256].float().abs().sum(-1) > 0
    assert not nz[:, :cls_tok].any() and not nz[:, cls_tok + 1:].any()    # dq lives in the CLS row only


@pytest.mark.gpu
@pytest.mark.parametrize("dt", DT)
def test_attn_cls_row_all_keys_masked(ops, dt):
    """A sample whose keys are ALL masked (kv_len = 0, padded layout): the reference's masked_fill(-65504) + softmax gives the
    uniform average over all N keys and no gradient to q / k (attention.py:36-41) -- the CLS-row kernels must agree with the
    dense kernel's `uniform` path and with the oracle (they used to attend to key 0 alone)."""
    torch.manual_seed(9)
    B, N, cls_tok = 3, 70, 4
    kv = torch.tensor([0, 40, 70], dtype=torch.int32, device=DEV)
    qkv = (torch.randn(B, N, 768, device=DEV) * 0.7).to(dt)
    z = torch.randn(B, N, 256, device=DEV).to(dt)
    d_o = torch.randn(B, 256, device=DEV).to(dt)
    qr = qkv.float().clone().requires_grad_(True)
    q, k, v = (t.reshape(B, N, 4, 64).permute(0, 2, 1, 3) for t in qr.split(256, dim=2))
    mask = (torch.arange(N, device=DEV)[None, :] >= kv[:, None])[:, None, None, :]
    sc = ((q @ k.transpose(-1, -2)) / 8.0).masked_fill(mask, -65504.0)
    o_ref = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B, N, 256)[:, cls_tok]
    (o_ref * d_o.float()).sum().backward()
    o, r1, lse = ops.attn_cls_fwd(qkv, z, kv, None, cls_tok)
    tol = 1e-4 if dt == torch.float32 else 1e-2
    tag = f"attn_cls_all_masked[{str(dt).replace('torch.', '')}]"
    check(tag + ".o", o.float(), o_ref, tol)
    assert torch.allclose(o.float()[0], qkv.float()[0, :, 512:].mean(0), atol=2e-2 if dt != torch.float32 else 1e-5)   # the plain mean of V
    dqkv = ops.attn_cls_bwd(qkv, o, d_o, lse, kv, None, cls_tok)
    check(tag + ".dqkv", dqkv.float(), qr.grad, 2e-4 if dt == torch.float32 else 2e-2)
    assert float(dqkv[0, :, :512].float().abs().max()) == 0.0               # no gradient to q, k of the all-masked sample
    # the dense kernels on the same inputs take the same path
    od, _, _ = ops.attn_fwd_grouped([qkv], [kv], [z], [None], None)
    assert torch.allclose(od[0][:, cls_tok].float(), o.float(), atol=tol * 4)


@pytest.mark.parametrize("workload", ["full", "ragged"])
def test_two_rank_rehearsal_on_one_gpu(workload):
    """The N > 1 code path end to end without a second GPU ("never run with a second rank"): two ranks of bench.py on
    GPU 0, collectives over gloo (--rehearse-on-one-gpu; everything but RCCL itself is the production path -- broadcast of rank 0's
    state, two captured graphs per step, merged bucket collectives issued behind each stage, the cross-rank plan check, AdamW on the
    summed gradients with the 1 / world factor).  The ranks see DIFFERENT batches; after warm-up + timed steps they must hold
    bit-identical parameters (bench.py checks it with MIN / MAX all-reduces of three checksums and fails otherwise)."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--workload", workload,
           "--steps", "3", "--warmup", "3", "--no-cpu-baseline", "--probe-launches", "0", "--instep-steps", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["graphs_per_step"] == 2 and d["hip_graph"]
    assert d["ranks_hold_identical_parameters"] is True and d["rehearsal_on_one_gpu_over_gloo"] is True
    REPORT[f"two_rank_rehearsal[{workload}].ranks_hold_identical_parameters"] = {"rel_err": 0.0, "tol": 0.0}
