// Backward of the Swin-T image encoder's building blocks for gfx950 (round 5): the sibling models that TRAIN the encoder
// (bi_vsltimg_mbt_v1.py:203-206, tri_mbt_v2.py:208-211, ... call self.img_encoder(img) without torch.no_grad()).
//
//  mtmp_layernorm_rows_bwd    : autograd of nn.LayerNorm(C, eps 1e-5) over the rows of an NHWC map
//                               (builder/models/src/swin_transformer.py:428-449 norm1 / norm2, :34-85 the merge norm, :611 norm)
//  mtmp_gelu_fwd / _bwd       : nn.GELU of the MLP (:437-439) as its own pass in the trainable path (the forward-only path
//                               keeps it in the projection's epilogue), and its derivative
//  mtmp_swin_window_attn_bwd  : autograd of mtmp_swin_window_attn (:115-225): dq, dk, dv written into a [n,H,W,3C] map at
//                               each token's own pixel, and the gradient of the additive table (relative-position bias)
// The GEMM-shaped parts of the backward (dX = dY W, dW = dY^T X) run on mtmp_gemm_nt / mtmp_gemm_tn.
#include "common.hip.h"

namespace {

constexpr int WS = 7, L = 49, LP = 64, DH = 32, LDV = LP + 8;
constexpr float LOG2E = 1.4426950408889634f;

// ------------------------------------------------------------------------------------------ LayerNorm backward
// Same row grouping as the forward (swin.hip ln_rows_kernel): a group of G lanes owns a row, a lane NCH chunks of 8 channels.
//   xh = (x - mean) rstd,  g = dy w,  dx = rstd (g - mean(g) - xh mean(g xh)),  dw += dy xh,  db += dy
// The parameter gradients are summed per lane over the rows the lane visits, then over the workgroup's row groups through LDS
// in a FIXED order (deterministic), and leave as one slab row per workgroup: slab[block][0][C] = dw, slab[block][1][C] = db.
template <typename T, int G, int NCH>
__global__ __launch_bounds__(256) void ln_rows_bwd_kernel(const T* x, const float* w, const T* dy, T* dx, float* slab, long long rows,
                                                          int C, float eps) {
    constexpr int RPW = 64 / G;
    extern __shared__ __attribute__((aligned(16))) float red[];      // [4 * RPW][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = lane & (G - 1), grp = lane / G;
    const int chunks = C >> 3;
    float aw[NCH][8], ab[NCH][8];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) { aw[k][i] = 0.f; ab[k][i] = 0.f; }
    for (long long rb = (long long)blockIdx.x * 4 + wave; rb * RPW < rows; rb += (long long)gridDim.x * 4) {
        const long long row = rb * RPW + grp;
        const bool valid = row < rows;
        const long long rc = valid ? row : rows - 1;
        float v[NCH][8], g[NCH][8];
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = gl + k * G;
            const bool on = ch < chunks;
            const int e = 8 * (on ? ch : 0);
            const Frag<T> f = frag_load<T>(x + rc * C + e);
            const Frag<T> d = frag_load<T>(dy + rc * C + e);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[k][i] = on ? to_f32(f.v[i]) : 0.f;
                g[k][i] = (on && valid) ? to_f32(d.v[i]) : 0.f;
                s1 += v[k][i];
            }
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
        const float mean = s1 / (float)C;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (gl + k * G < chunks) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float d = v[k][i] - mean; s2 += d * d; }
            }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        const float rstd = rsqrtf(s2 / (float)C + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = gl + k * G;
            if (ch < chunks) {
                const int e = 8 * ch;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + e), w1 = *reinterpret_cast<const f32x4*>(w + e + 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float xh = (v[k][i] - mean) * rstd, dyv = g[k][i];
                    aw[k][i] += dyv * xh;
                    ab[k][i] += dyv;
                    const float gw = dyv * (i < 4 ? w0[i] : w1[i - 4]);
                    v[k][i] = xh;
                    g[k][i] = gw;
                    sg += gw;
                    sgx += gw * xh;
                }
            }
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) { sg += __shfl_xor(sg, o, 64); sgx += __shfl_xor(sgx, o, 64); }
        const float mg = sg / (float)C, mgx = sgx / (float)C;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = gl + k * G;
            if (valid && ch < chunks) {
                Frag<T> o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o.v[i] = from_f32<T>(rstd * (g[k][i] - mg - v[k][i] * mgx));
                frag_store<T>(dx + row * C + 8 * ch, o);
            }
        }
    }
    // workgroup reduction of the parameter gradients, fixed order
    float* mine = red + (size_t)(wave * RPW + grp) * 2 * C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = gl + k * G;
        if (ch < chunks) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { mine[8 * ch + i] = aw[k][i]; mine[C + 8 * ch + i] = ab[k][i]; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        float s = 0.f;
        for (int j = 0; j < 4 * RPW; ++j) s += red[(size_t)j * 2 * C + c];
        slab[(size_t)blockIdx.x * 2 * C + c] = s;
    }
}

int ln_bwd_blocks(long long rows, int C) {
    const int chunks = C / 8;
    int G = 16;
    while (G < chunks && G < 64) G <<= 1;
    const long long nrb = (rows + (64 / G) - 1) / (64 / G);
    const long long nb = (nrb + 3) / 4;
    return (int)(nb < 1024 ? nb : 1024);
}

template <typename T>
int launch_ln_rows_bwd(const void* x, const float* w, const void* dy, void* dx, float* slab, long long rows, int C, float eps,
                       hipStream_t st) {
    const int chunks = C / 8;
    int G = 16;
    while (G < chunks && G < 64) G <<= 1;
    const int nch = (chunks + G - 1) / G;
    const int nb = ln_bwd_blocks(rows, C);
    const size_t sm = (size_t)4 * (64 / G) * 2 * C * sizeof(float);
#define MTMP_LNB_CASE(g, n)                                                                                             \
    if (G == g && nch == n) {                                                                                           \
        if (sm > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(ln_rows_bwd_kernel<T, g, n>),          \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) { \
            mtmp_set_error("mtmp_layernorm_rows_bwd: hipFuncSetAttribute(%zu) failed", sm);                             \
            return MTMP_ERR_LAUNCH;                                                                                     \
        }                                                                                                               \
        hipLaunchKernelGGL((ln_rows_bwd_kernel<T, g, n>), dim3(nb), dim3(256), sm, st, (const T*)x, w, (const T*)dy, (T*)dx, slab, \
                           rows, C, eps);                                                                               \
        return MTMP_OK;                                                                                                 \
    }
    MTMP_LNB_CASE(16, 1) MTMP_LNB_CASE(32, 1) MTMP_LNB_CASE(64, 1) MTMP_LNB_CASE(64, 2) MTMP_LNB_CASE(64, 3)
#undef MTMP_LNB_CASE
    mtmp_set_error("mtmp_layernorm_rows_bwd: unsupported C=%d", C);
    return MTMP_ERR_ARG;
}

// ------------------------------------------------------------------------------------------ GELU
// Forward: common.hip.h gelu<T> (exact erf form in the fp32 parity build, x sigmoid(2 z) in the bf16 build -- the same function the
// projection epilogue of the forward-only path applies).  Derivative of the SAME form per build.
template <typename T> MTMP_DEV float gelu_grad(float x);
template <> MTMP_DEV float gelu_grad<float>(float x) {
    const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f));
    return cdf + x * 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.5f * x * x * LOG2E);
}
template <> MTMP_DEV float gelu_grad<bf16>(float x) {
    const float x2 = x * x;
    const float s = x * fmaf(x2, 0.0713548163f, 1.5957691216f);                // 2 z
    const float sig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-s * LOG2E));
    return sig + x * sig * (1.0f - sig) * fmaf(x2, 3.0f * 0.0713548163f, 1.5957691216f);
}

template <typename T, bool BWD> __global__ __launch_bounds__(256) void gelu_kernel(const T* x, const T* dy, T* out, long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const Frag<T> f = frag_load<T>(x + 8 * i);
        Frag<T> o;
        if (BWD) {
            const Frag<T> d = frag_load<T>(dy + 8 * i);
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = from_f32<T>(to_f32(d.v[j]) * gelu_grad<T>(to_f32(f.v[j])));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = from_f32<T>(gelu<T>(to_f32(f.v[j])));
        }
        frag_store<T>(out + 8 * i, o);
    }
}

// ------------------------------------------------------------------------------------------ window attention backward
// One wave per (image, window, head), like the forward: 49 tokens padded to 64, head_dim 32.  Two passes over the 64 x 64 scores:
//   pass 1, keys on the accumulator rows / queries on the lanes (the forward's orientation):
//       S^T = K Q^T -> softmax per lane -> P^T;  dP^T = V dO^T;  delta = sum_k P dP;  dS^T = P^T (dP^T - delta)
//       dQ^T += K^T dS^T (contraction over keys = accumulator rows);  d table[q][k] += dS  (float atomics: every window of a
//       type and every image adds into the same [type][head] plane);  log-sum-exp and delta of every query go to LDS
//   pass 2, queries on the rows / keys on the lanes:  S = Q K^T, P = exp(S - lse[q]), dP = dO V^T, dS = P (dP - delta[q]) with the
//       row constants read back from LDS;  dV^T += dO^T P,  dK^T += Q^T dS  (contraction over queries = accumulator rows)
// The transposed operands (K^T, Q^T, dO^T: head dims on the rows) are staged per wave in LDS like V^T in the forward.  Pad
// tokens (49..63) are zero fragments; their keys carry -30000 in the table, their queries a zero dO: they add nothing.
template <typename T>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1))
void swin_wattn_bwd_kernel(const T* qkv, const T* table, const T* dout, T* dqkv, float* dtab, int n_img, int H, int W, int C,
                           int heads, int shift, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    T* sKt = reinterpret_cast<T*>(smem_raw) + (size_t)wave * 3 * DH * LDV;      // this wave's [32 d][LDV tokens] images
    T* sQt = sKt + DH * LDV;
    T* sOt = sQt + DH * LDV;
    float* sRow = reinterpret_cast<float*>(reinterpret_cast<T*>(smem_raw) + (size_t)4 * 3 * DH * LDV) + wave * 2 * LP;   // lse | delta
    const int nWh = H / WS, nWw = W / WS;
    const long long total = (long long)n_img * nWh * nWw * heads;
    const long long task = (long long)blockIdx.x * 4 + wave;
    const bool live = task < total;
    const long long tsk = live ? task : 0;
    const int head = (int)(tsk % heads);
    const long long wl = tsk / heads;
    const int win = (int)(wl % (nWh * nWw)), img = (int)(wl / (nWh * nWw));
    const int wi = win / nWw, wj = win - wi * nWw;
    const int type = shift > 0 ? ((wi == nWh - 1 ? 2 : 0) + (wj == nWw - 1 ? 1 : 0)) : 0;
    const int C3 = 3 * C;
    auto pix = [&](int t) -> long long {       // token t of this window -> its pixel in the UN-shifted map (swin.hip)
        const int ty = t / WS, tx = t - ty * WS;
        int yy = wi * WS + ty + shift, xx = wj * WS + tx + shift;
        if (yy >= H) yy -= H;
        if (xx >= W) xx -= W;
        return ((long long)img * H + yy) * W + xx;
    };
    // operand fragments of one token set: natural order (B operands) or through swz23 (A operands)
    auto load_set = [&](const T* base, int ld, int col0, bool swz, Frag<T> (&f)[2][2]) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int t = 32 * blk + (swz ? swz23(r) : r);
            const T* ptr = base + pix(t < L ? t : 0) * ld + col0 + head * DH + 8 * half;
#pragma unroll
            for (int c = 0; c < 2; ++c) f[blk][c] = frag_keep(frag_load<T>(ptr + 16 * c), live && t < L);
        }
    };
    // [32 d][tokens] image of one tensor in LDS (lane = token pair x 8-dim group, two passes over the 32 dims)
    auto stage_t = [&](const T* base, int ld, int col0, T* dst) {
        const int kp = (lane & 31) * 2, dg = (lane >> 5) * 8;
        const T* pa = base + pix(kp < L ? kp : 0) * ld + col0 + head * DH + dg;
        const T* pb = base + pix(kp + 1 < L ? kp + 1 : 0) * ld + col0 + head * DH + dg;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const Frag<T> fa = frag_keep(frag_load<T>(pa + 16 * ps), live && kp < L);
            const Frag<T> fb = frag_keep(frag_load<T>(pb + 16 * ps), live && kp + 1 < L);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                T* d = dst + (dg + 16 * ps + e) * LDV + kp;
                d[0] = fa.v[e];
                d[1] = fb.v[e];
            }
        }
    };
    stage_t(qkv, C3, C, sKt);
    stage_t(qkv, C3, 0, sQt);
    stage_t(dout, C, 0, sOt);
    const T* tab = table + ((size_t)type * heads + head) * LP * LP;
    float* dtb = dtab + ((size_t)type * heads + head) * LP * LP;
    __syncthreads();
    // ------------------------------------------------ pass 1: keys on rows, queries on lanes
    {
        Frag<T> qB[2][2], kA[2][2], vA[2][2], oB[2][2];
        load_set(qkv, C3, 0, false, qB);
        load_set(qkv, C3, C, true, kA);
        load_set(qkv, C3, 2 * C, true, vA);
        load_set(dout, C, 0, false, oB);
        f32x16 st[2][2] = {{{0}, {0}}, {{0}, {0}}}, dp[2][2] = {{{0}, {0}}, {{0}, {0}}};      // [key block][query block]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    mma<T>(st[kb][qb], kA[kb][c], qB[qb][c]);
                    mma<T>(dp[kb][qb], vA[kb][c], oB[qb][c]);
                }
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const T* trow = tab + (32 * qb + r) * LP + 8 * half;
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const Frag<T> t0 = frag_load<T>(trow + 32 * kb), t1 = frag_load<T>(trow + 32 * kb + 16);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    st[kb][qb][t] = fmaf(st[kb][qb][t], scale, to_f32(t0.v[t]));
                    st[kb][qb][t + 8] = fmaf(st[kb][qb][t + 8], scale, to_f32(t1.v[t]));
                    mx = fmaxf(mx, fmaxf(st[kb][qb][t], st[kb][qb][t + 8]));
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float l = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float pv = fast_exp2((st[kb][qb][t] - mx) * LOG2E);
                    st[kb][qb][t] = pv;
                    l += pv;
                }
            l += __shfl_xor(l, 32, 64);
            const float linv = 1.0f / l;
            float dl = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    st[kb][qb][t] *= linv;
                    dl = fmaf(st[kb][qb][t], dp[kb][qb][t], dl);
                }
            dl += __shfl_xor(dl, 32, 64);
            if (half == 0) {
                sRow[32 * qb + r] = mx + __builtin_amdgcn_logf(l) * 0.6931471805599453f;      // natural-log LSE of the logits
                sRow[LP + 32 * qb + r] = dl;
            }
            const int q = 32 * qb + r;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float ds = st[kb][qb][t] * (dp[kb][qb][t] - dl);
                    st[kb][qb][t] = ds;
                    const int key = 32 * kb + acc_row_swz(t, half);
                    if (live && q < L && key < L) atomicAdd(dtb + q * LP + key, ds);
                }
        }
        // dQ^T = K^T dS^T (rows = head dims, cols = queries), scaled; written to the query's own pixel
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 dq = {0};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    mma<T>(dq, frag_load<T>(sKt + r * LDV + 32 * kb + 16 * s + 8 * half), frag_from_acc<T>(st[kb][qb], s));
            const int tq = 32 * qb + r;
            if (live && tq < L) {
                T* po = dqkv + pix(tq) * C3 + head * DH + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    store4<T>(po + 8 * g, dq[4 * g] * scale, dq[4 * g + 1] * scale, dq[4 * g + 2] * scale, dq[4 * g + 3] * scale);
            }
        }
    }
    __syncthreads();                       // lse / delta of this wave's queries are in LDS
    // ------------------------------------------------ pass 2: queries on rows, keys on lanes
    {
        Frag<T> qA[2][2], kB[2][2], oA[2][2], vB[2][2];
        load_set(qkv, C3, 0, true, qA);
        load_set(qkv, C3, C, false, kB);
        load_set(dout, C, 0, true, oA);
        load_set(qkv, C3, 2 * C, false, vB);
        f32x16 dk[2] = {{0}, {0}}, dv[2] = {{0}, {0}};                    // [key block]: rows = head dims, cols = keys
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            // row constants of this query block: register t <-> query 32 qb + 16 (t >> 3) + 8 half + (t & 7)
            float lse[16], dl[16];
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(sRow + 32 * qb + 16 * h8 + 8 * half + 4 * v);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(sRow + LP + 32 * qb + 16 * h8 + 8 * half + 4 * v);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { lse[8 * h8 + 4 * v + i] = a[i]; dl[8 * h8 + 4 * v + i] = b[i]; }
                }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 s2 = {0}, d2 = {0};
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    mma<T>(s2, qA[qb][c], kB[kb][c]);
                    mma<T>(d2, oA[qb][c], vB[kb][c]);
                }
                const int key = 32 * kb + r;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int q = 32 * qb + acc_row_swz(t, half);
                    const float logit = fmaf(s2[t], scale, to_f32(tab[q * LP + key]));
                    const float pv = fast_exp2((logit - lse[t]) * LOG2E);
                    s2[t] = pv;
                    d2[t] = pv * (d2[t] - dl[t]);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    mma<T>(dv[kb], frag_load<T>(sOt + r * LDV + 32 * qb + 16 * s + 8 * half), frag_from_acc<T>(s2, s));
                    mma<T>(dk[kb], frag_load<T>(sQt + r * LDV + 32 * qb + 16 * s + 8 * half), frag_from_acc<T>(d2, s));
                }
            }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int tk = 32 * kb + r;
            if (live && tk < L) {
                T* pk = dqkv + pix(tk) * C3 + C + head * DH + 4 * half;
                T* pv = dqkv + pix(tk) * C3 + 2 * C + head * DH + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    store4<T>(pk + 8 * g, dk[kb][4 * g] * scale, dk[kb][4 * g + 1] * scale, dk[kb][4 * g + 2] * scale,
                              dk[kb][4 * g + 3] * scale);
                    store4<T>(pv + 8 * g, dv[kb][4 * g], dv[kb][4 * g + 1], dv[kb][4 * g + 2], dv[kb][4 * g + 3]);
                }
            }
        }
    }
}

}  // namespace

// slab rows (= workgroups) mtmp_layernorm_rows_bwd writes: slab must hold 2 * C floats per row
extern "C" int mtmp_layernorm_rows_bwd_slab_rows(long long rows, int C) { return ln_bwd_blocks(rows, C); }

// dx [rows, C] (dtype) and the per-workgroup partial sums of the weight / bias gradients, slab float[slab_rows][2][C] (the caller
// sums the rows): autograd of nn.LayerNorm(C, eps) as mtmp_layernorm_rows computes it (biased variance, eps inside the root;
// builder/models/src/swin_transformer.py:428-449, :34-85, :611).  x, dy [rows, C] contiguous.
extern "C" int mtmp_layernorm_rows_bwd(int dtype, const void* x, const float* w, const void* dy, void* dx, float* slab,
                                       long long rows, int C, float eps, void* stream) {
    MTMP_CHECK_ARG(x && w && dy && dx && slab && rows > 0, "mtmp_layernorm_rows_bwd: bad pointer / rows");
    MTMP_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 1536, "mtmp_layernorm_rows_bwd: bad C=%d", C);
    hipStream_t st = (hipStream_t)stream;
    int e;
    if (dtype == 0) e = launch_ln_rows_bwd<float>(x, w, dy, dx, slab, rows, C, eps, st);
    else if (dtype == 1) e = launch_ln_rows_bwd<bf16>(x, w, dy, dx, slab, rows, C, eps, st);
    else { mtmp_set_error("mtmp_layernorm_rows_bwd: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    if (e) return e;
    MTMP_CHECK_LAUNCH("mtmp_layernorm_rows_bwd");
    return MTMP_OK;
}

// y = GELU(x) over n elements (n % 8 == 0): nn.GELU of the Swin MLP (swin_transformer.py:437-439) as its own pass.
extern "C" int mtmp_gelu_fwd(int dtype, const void* x, void* y, long long n, void* stream) {
    MTMP_CHECK_ARG(x && y && n > 0 && n % 8 == 0, "mtmp_gelu_fwd: bad argument (n=%lld)", n);
    hipStream_t st = (hipStream_t)stream;
    const long long n8 = n / 8;
    const int nb = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
    if (dtype == 0) hipLaunchKernelGGL((gelu_kernel<float, false>), dim3(nb), dim3(256), 0, st, (const float*)x, (const float*)nullptr, (float*)y, n8);
    else if (dtype == 1) hipLaunchKernelGGL((gelu_kernel<bf16, false>), dim3(nb), dim3(256), 0, st, (const bf16*)x, (const bf16*)nullptr, (bf16*)y, n8);
    else { mtmp_set_error("mtmp_gelu_fwd: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_gelu_fwd");
    return MTMP_OK;
}

// dx = dy * GELU'(x) (the derivative of the form mtmp_gelu_fwd / the projection epilogue applies in this build).
extern "C" int mtmp_gelu_bwd(int dtype, const void* x, const void* dy, void* dx, long long n, void* stream) {
    MTMP_CHECK_ARG(x && dy && dx && n > 0 && n % 8 == 0, "mtmp_gelu_bwd: bad argument (n=%lld)", n);
    hipStream_t st = (hipStream_t)stream;
    const long long n8 = n / 8;
    const int nb = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
    if (dtype == 0) hipLaunchKernelGGL((gelu_kernel<float, true>), dim3(nb), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, n8);
    else if (dtype == 1) hipLaunchKernelGGL((gelu_kernel<bf16, true>), dim3(nb), dim3(256), 0, st, (const bf16*)x, (const bf16*)dy, (bf16*)dx, n8);
    else { mtmp_set_error("mtmp_gelu_bwd: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_gelu_bwd");
    return MTMP_OK;
}

// Autograd of mtmp_swin_window_attn: qkv [n,H,W,3C], table [4][heads][64][64] (dtype), dout [n,H,W,C] ->
// dqkv [n,H,W,3C] (every element written once: H, W multiples of 7) and dtab float[4][heads][64][64], which the caller ZEROES
// first (float atomics: the windows of a type add into one plane).
extern "C" int mtmp_swin_window_attn_bwd(int dtype, const void* qkv, const void* table, const void* dout, void* dqkv, float* dtab,
                                         int n_img, int H, int W, int C, int heads, int shift, float scale, void* stream) {
    MTMP_CHECK_ARG(qkv && table && dout && dqkv && dtab, "mtmp_swin_window_attn_bwd: null pointer");
    MTMP_CHECK_ARG(n_img > 0 && H > 0 && W > 0 && H % WS == 0 && W % WS == 0 && heads > 0 && C == heads * DH && shift >= 0 && shift < WS,
                   "mtmp_swin_window_attn_bwd: bad shape n=%d H=%d W=%d C=%d heads=%d shift=%d", n_img, H, W, C, heads, shift);
    hipStream_t st = (hipStream_t)stream;
    const long long tasks = (long long)n_img * (H / WS) * (W / WS) * heads;
    const int nb = (int)((tasks + 3) / 4);
    if (dtype == 0) {
        const size_t sm = (size_t)4 * 3 * DH * LDV * sizeof(float) + 4 * 2 * LP * sizeof(float);
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(swin_wattn_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sm) != hipSuccess) {
            mtmp_set_error("mtmp_swin_window_attn_bwd: hipFuncSetAttribute(%zu) failed", sm);
            return MTMP_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(swin_wattn_bwd_kernel<float>, dim3(nb), dim3(256), sm, st, (const float*)qkv, (const float*)table,
                           (const float*)dout, (float*)dqkv, dtab, n_img, H, W, C, heads, shift, scale);
    } else if (dtype == 1) {
        const size_t sm = (size_t)4 * 3 * DH * LDV * sizeof(bf16) + 4 * 2 * LP * sizeof(float);       // 56 KB
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(swin_wattn_bwd_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sm) != hipSuccess) {
            mtmp_set_error("mtmp_swin_window_attn_bwd: hipFuncSetAttribute(%zu) failed", sm);
            return MTMP_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(swin_wattn_bwd_kernel<bf16>, dim3(nb), dim3(256), sm, st, (const bf16*)qkv, (const bf16*)table,
                           (const bf16*)dout, (bf16*)dqkv, dtab, n_img, H, W, C, heads, shift, scale);
    } else { mtmp_set_error("mtmp_swin_window_attn_bwd: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_swin_window_attn_bwd");
    return MTMP_OK;
}
