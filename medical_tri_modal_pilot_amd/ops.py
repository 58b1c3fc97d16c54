"""Torch-facing wrappers of the C-ABI kernels (include/mtmp.h) and the hand-written
forward/backward of one encoder layer.

PyTorch is plumbing here: it owns device memory and streams, and runs the *plain*
backward GEMMs (dW = dY^T X, dX = dY W) through its BLAS.  Every fused op of the hot
path (LN+projection, attention fwd/bwd, FFN, LN backward, TIE embedding, stem,
AdamW) is a libmtmp_hip.so kernel; there is no CPU or eager fallback -- inputs
that are not on a GPU raise.
"""
import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import BF16, F32, call

LN_EPS = 1e-6          # builder/models/src/transformer/module.py:132
N_HEAD, D_MODEL, D_HEAD = 4, 256, 64


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"libmtmp_hip kernels take float32 or bfloat16, got {t.dtype}")


def _p(t: Optional[torch.Tensor], byte_offset: int = 0):
    return None if t is None else ctypes.c_void_p(t.data_ptr() + byte_offset)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("medical_tri_modal_pilot_amd ops run on an MI355X only (tensor is on "
                               f"{t.device}); there is no CPU fallback")


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- raw kernels
def ln_gemm(x2d, gamma, beta, w, bias, n_out, relu=False, drop_p=0.0, seed=0, want_xn=True):
    """y = act(LN(x) w^T + bias); returns (y[M,n_out], xn[M,256] | None, stats[M,2])."""
    _gpu(x2d, w)
    M = x2d.shape[0]
    y = torch.empty(M, n_out, dtype=x2d.dtype, device=x2d.device)
    xn = torch.empty(M, D_MODEL, dtype=x2d.dtype, device=x2d.device) if want_xn else None
    stats = torch.empty(M, 2, dtype=torch.float32, device=x2d.device)
    call("mtmp_ln_gemm", _dt(x2d), _p(x2d), _p(gamma), _p(beta), _p(w), _p(bias), _p(y), _p(xn), _p(stats),
         M, n_out, x2d.stride(0), n_out, LN_EPS, int(relu), float(drop_p), int(seed) & 0xFFFFFFFF, _stream())
    return y, xn, stats


def gemm_nt(a2d, w, bias=None, res2d=None, relu=False, drop_p=0.0, seed=0):
    """y = drop(act(a w^T + bias)) (+ res)."""
    _gpu(a2d, w)
    M, K = a2d.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=a2d.dtype, device=a2d.device)
    call("mtmp_gemm_nt", _dt(a2d), _p(a2d), _p(w), _p(bias), _p(res2d), _p(y), M, N, K, a2d.stride(0), N,
         0 if res2d is None else res2d.stride(0), int(relu), float(drop_p), int(seed) & 0xFFFFFFFF, _stream())
    return y


def gemm_tn(dy2d, x2d, want_bias=True):
    """(dW[N,K], db[N] | None) in fp32: dW = dy^T x, db = column sums of dy (split over the M tokens)."""
    _gpu(dy2d, x2d)
    M, N = dy2d.shape
    K = x2d.shape[1]
    dw = torch.empty(N, K, dtype=torch.float32, device=dy2d.device)
    db = torch.empty(N, dtype=torch.float32, device=dy2d.device) if want_bias else None
    ws = torch.empty(_lib.lib().mtmp_gemm_tn_ws_floats(M, N, K), dtype=torch.float32, device=dy2d.device)
    call("mtmp_gemm_tn", _dt(dy2d), _p(dy2d), _p(x2d), _p(dw), _p(db), _p(ws), M, N, K, dy2d.stride(0),
         x2d.stride(0), _stream())
    return dw, db


def attn_fwd(qkv, kv_len, res=None):
    """qkv [B,N,768] (q|k|v), kv_len int32[B] or None -> (o[B,N,256], o+res | None, lse[B,4,N])."""
    _gpu(qkv)
    B, N, _ = qkv.shape
    es = qkv.element_size()
    o = torch.empty(B, N, D_MODEL, dtype=qkv.dtype, device=qkv.device)
    o_res = torch.empty_like(o) if res is not None else None
    lse = torch.empty(B, N_HEAD, N, dtype=torch.float32, device=qkv.device)
    call("mtmp_attn_fwd", _dt(qkv), _p(qkv), _p(qkv, D_MODEL * es), _p(qkv, 2 * D_MODEL * es), _p(o), _p(res),
         _p(o_res), _p(lse), _p(kv_len), B, N, N_HEAD, qkv.stride(1), D_MODEL, D_HEAD ** -0.5, _stream())
    return o, o_res, lse


def attn_bwd(qkv, o, d_o, lse, kv_len):
    """-> dqkv [B,N,768]."""
    _gpu(qkv, o, d_o)
    B, N, _ = qkv.shape
    es = qkv.element_size()
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B * N_HEAD * N, dtype=torch.float32, device=qkv.device)
    call("mtmp_attn_bwd", _dt(qkv), _p(qkv), _p(qkv, D_MODEL * es), _p(qkv, 2 * D_MODEL * es), _p(o), _p(d_o),
         _p(lse), _p(kv_len), _p(dqkv), _p(dqkv, D_MODEL * es), _p(dqkv, 2 * D_MODEL * es), _p(delta),
         B, N, N_HEAD, qkv.stride(1), o.stride(1), d_o.stride(1), dqkv.stride(1), D_HEAD ** -0.5, _stream())
    return dqkv


def ln_bwd(z2d, stats, gamma, dy2d, d_res2d=None):
    """-> (dz[M,256], dgamma[256], dbeta[256]) of the custom LayerNorm (+ residual gradient)."""
    _gpu(z2d, dy2d)
    M = z2d.shape[0]
    dz = torch.empty(M, D_MODEL, dtype=z2d.dtype, device=z2d.device)
    gb = torch.empty(2 * D_MODEL, dtype=torch.float32, device=z2d.device)
    ws = torch.empty(_lib.lib().mtmp_ln_bwd_ws_floats(M), dtype=torch.float32, device=z2d.device)
    call("mtmp_ln_bwd", _dt(z2d), _p(z2d), z2d.stride(0), _p(stats), _p(gamma), _p(dy2d), _p(d_res2d),
         0 if d_res2d is None else d_res2d.stride(0), _p(dz), _p(gb), _p(ws), M, LN_EPS, _stream())
    return dz, gb[:D_MODEL], gb[D_MODEL:]


def dropout_bwd(g, seed, p):
    out = torch.empty_like(g)
    call("mtmp_dropout_bwd", _dt(g), _p(g), _p(out), g.numel(), int(seed) & 0xFFFFFFFF, float(p), _stream())
    return out


def swin_stem(img, w, b, ln_w, ln_b, dtype):
    """img [n,1,H,W] fp32 -> [n,H/4,W/4,96] (Conv 4x4/4 + LayerNorm), no gradient (frozen encoder)."""
    _gpu(img, w)
    n, _, H, W = img.shape
    img = _c(img.float())
    out = torch.empty(n, H // 4, W // 4, 96, dtype=dtype, device=img.device)
    call("mtmp_swin_stem_fwd", _dt(out), _p(img), _p(_c(w)), _p(b), _p(ln_w), _p(ln_b), _p(out), n, H, W, _stream())
    return out


def adamw_step(param, grad, exp_avg, exp_avg_sq, shadow, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _gpu(param, grad)
    call("mtmp_adamw_step", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(shadow), param.numel(), float(lr),
         float(beta1), float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale), _stream())


def _mm_f32(a, b):
    """Plain library GEMM with an fp32 result (weight gradients are kept in fp32)."""
    if a.dtype == torch.float32:
        return a @ b
    try:
        return torch.mm(a, b, out_dtype=torch.float32)
    except (TypeError, RuntimeError):
        return (a @ b).float()


# ----------------------------------------------------------------------------- TIE embedding
class TieEmbed(torch.autograd.Function):
    """tri_mbt_vsltcls.py:183-190.  events [B,T,3] fp32 (already rounded through fp16 by the
    trainer); parameters are the reference's ie_vslt / ie_time / ie_feat tensors."""

    @staticmethod
    def forward(ctx, events, wv, bv, gv, hv, wt, bt, gt, ht, ftab, dtype):
        _gpu(events, ftab)
        B, T, _ = events.shape
        ev = _c(events.float()).view(B * T, 3)
        prm = torch.stack([wv.reshape(-1), bv, gv, hv, wt.reshape(-1), bt, gt, ht]).float().contiguous()
        ft = _c(ftab.float())
        out = torch.empty(B, T, D_MODEL, dtype=dtype, device=events.device)
        call("mtmp_tie_embed_fwd", _dt(out), _p(ev), _p(prm), _p(ft), _p(out), B * T, _stream())
        ctx.save_for_backward(ev, prm)
        ctx.wshape = (wv.shape, wt.shape)
        return out

    @staticmethod
    def backward(ctx, d_out):
        ev, prm = ctx.saved_tensors
        n = ev.shape[0]
        d_out = _c(d_out)
        grads = torch.empty(28, D_MODEL, dtype=torch.float32, device=ev.device)
        ws = torch.empty(_lib.lib().mtmp_tie_bwd_ws_floats(n), dtype=torch.float32, device=ev.device)
        call("mtmp_tie_embed_bwd", _dt(d_out), _p(ev), _p(prm), _p(d_out), _p(grads), _p(ws), n, _stream())
        g = grads
        return (None, g[0].view(ctx.wshape[0]), g[1], g[2], g[3], g[4].view(ctx.wshape[1]), g[5], g[6], g[7],
                g[8:28], None)


# ----------------------------------------------------------------------------- encoder layer
class EncoderLayerFn(torch.autograd.Function):
    """One pre-LN encoder block (builder/models/src/transformer/encoder.py:23-34) on a
    [B, N, 256] stream with per-sample valid-key counts:

        r1  = z + MHA(LN1(z); kv_len)          mtmp_ln_gemm(QKV) + mtmp_attn_fwd(+residual)
        out = r1 + FFN(LN2(r1))                mtmp_ln_gemm(ReLU, drop1) + mtmp_gemm_nt(drop2, +residual)

    Backward is written out by hand (no autograd graph inside): HIP kernels for attention,
    LayerNorm and dropout; plain BLAS GEMMs for dW / dX.  Activations needed by the backward
    are kept (HBM is 288 GB; one vslt layer at B=64, T=1000 keeps ~0.4 GB in bf16).
    """

    @staticmethod
    def forward(ctx, z, kv_len, g1, b1, wq, bq, wk, bk, wv, bv, g2, b2, w1, c1, w2, c2, fused, drop_p, seeds):
        _gpu(z)
        z = _c(z)
        B, N, D = z.shape
        M = B * N
        dt = z.dtype
        wqkv, bqkv, w1c, w2c = fused           # compute-dtype weights prepared by the module
        z2 = z.view(M, D)
        qkv, xn1, st1 = ln_gemm(z2, g1, b1, wqkv, bqkv, 3 * D)
        qkv = qkv.view(B, N, 3 * D)
        o, r1, lse = attn_fwd(qkv, kv_len, res=z)
        r1_2 = r1.view(M, D)
        h, xn2, st2 = ln_gemm(r1_2, g2, b2, w1c, c1, 4 * D, relu=True, drop_p=drop_p, seed=seeds[0])
        out = gemm_nt(h, w2c, c2, res2d=r1_2, drop_p=drop_p, seed=seeds[1])
        ctx.save_for_backward(z, kv_len, g1, g2, wqkv, w1c, w2c, xn1, st1, qkv, o, lse, r1, xn2, st2, h)
        ctx.drop_p, ctx.seeds = drop_p, seeds
        ctx.wshapes = (w1.shape, w2.shape)
        return out.view(B, N, D)

    @staticmethod
    def backward(ctx, d_out):
        z, kv_len, g1, g2, wqkv, w1c, w2c, xn1, st1, qkv, o, lse, r1, xn2, st2, h = ctx.saved_tensors
        B, N, D = z.shape
        M = B * N
        p = ctx.drop_p
        d_out = _c(d_out).view(M, D)
        if d_out.dtype != z.dtype:
            d_out = d_out.to(z.dtype)
        # ---- FFN: out = drop2(h w2^T + c2) + r1,  h = drop1(relu(LN2(r1) w1^T + c1))
        dy2 = dropout_bwd(d_out, ctx.seeds[1], p) if p > 0 else d_out
        dw2, dc2 = gemm_tn(dy2, h)                        # [256,1024], [256]
        dh = dy2 @ w2c                                    # [M,1024]
        dh = torch.where(h > 0, dh, torch.zeros((), dtype=dh.dtype, device=dh.device))
        if p > 0:
            dh = dh * (1.0 / (1.0 - p))                   # h > 0 already encodes relu AND drop1's mask
        dw1, dc1 = gemm_tn(dh, xn2)                       # [1024,256], [1024]
        dxn2 = dh @ w1c                                   # [M,256]
        dr1, dg2, db2 = ln_bwd(r1.view(M, D), st2, g2, dxn2, d_res2d=d_out)
        # ---- attention: r1 = z + o  ->  d_o = dr1
        dqkv = attn_bwd(qkv, o, dr1.view(B, N, D), lse, kv_len).view(M, 3 * D)
        dwqkv, dbqkv = gemm_tn(dqkv, xn1)                 # [768,256], [768]
        dxn1 = dqkv @ wqkv                                # [M,256]
        dz, dg1, db1 = ln_bwd(z.view(M, D), st1, g1, dxn1, d_res2d=dr1)
        return (dz.view(B, N, D), None, dg1, db1,
                dwqkv[:D], dbqkv[:D], dwqkv[D:2 * D], dbqkv[D:2 * D], dwqkv[2 * D:], dbqkv[2 * D:],
                dg2, db2, dw1.view(ctx.wshapes[0]), dc1, dw2.view(ctx.wshapes[1]), dc2, None, None, None)
