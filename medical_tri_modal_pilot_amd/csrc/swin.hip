// Frozen Swin-T image encoder beyond the stem, for gfx950: forward only.
//
//  mtmp_layernorm_rows  : nn.LayerNorm(C, eps 1e-5) over the rows of an NHWC map (norm1 / norm2 /
//      final norm of builder/models/src/swin_transformer.py:428-449,611-612), optionally fused with
//      the 2x2 patch-merging gather of :34-44 (the [.., 4C] concat is never materialised).
//  mtmp_swin_window_attn: shifted-window multi-head attention of :115-225 -- cyclic shift,
//      window partition, q*scale, QK^T, + relative-position bias (+ -100 shift mask), softmax,
//      PV, window reverse and un-shift -- as ONE kernel that addresses tokens of the un-shifted
//      [B,H,W,3C] qkv map directly (no roll / permute / reshape copies).  One wave per
//      (image, window, head): 49 tokens padded to 64, head_dim 32 -> 8 + 8 MFMAs 32x32x16;
//      the query is a lane (softmax in registers), P feeds P.V from the accumulators.
//      The additive table [4 window types][heads][64][64] (bias + mask, -30000 on pad keys) is
//      constant per block and precomputed once by the host.
#include "common.cuh"

namespace {

constexpr int WS = 7, L = 49, LP = 64, DH = 32, LDV = LP + 8;
constexpr float LOG2E = 1.4426950408889634f;

// ------------------------------------------------------------------------------------------
// A wave is split into 64/G groups of G lanes; a group normalises one row, a lane owns NCH chunks of
// 8 consecutive channels (16-byte loads/stores).  C = 96 -> 4 rows per wave, 192 -> 2, >= 384 -> 1.
template <typename T, int G, int NCH>
__global__ __launch_bounds__(256) void ln_rows_kernel(const T* x, const float* w, const float* b, T* y, long long rows,
                                                      int C, float eps, int merge, int H, int W) {
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = lane & (G - 1), grp = lane / G;
    const int chunks = C >> 3;
    const int Cs = merge ? (C >> 2) : C;           // channels of one source pixel
    for (long long rb = (long long)blockIdx.x * 4 + wave; rb * RPW < rows; rb += (long long)gridDim.x * 4) {
        const long long row = rb * RPW + grp;
        const bool valid = row < rows;
        const long long rc = valid ? row : rows - 1;
        const T* src[4];
        if (merge) {
            const int Ho = H >> 1, Wo = W >> 1;
            const long long img = rc / (Ho * Wo);
            const int rem = (int)(rc - img * Ho * Wo), i = rem / Wo, j = rem - i * Wo;
            const T* base = x + ((img * H + 2 * i) * W + 2 * j) * (long long)Cs;
            src[0] = base;                           // x[0::2, 0::2]
            src[1] = base + (long long)W * Cs;       // x[1::2, 0::2]
            src[2] = base + Cs;                      // x[0::2, 1::2]
            src[3] = base + (long long)W * Cs + Cs;  // x[1::2, 1::2]
        } else {
            src[0] = src[1] = src[2] = src[3] = x + rc * C;
        }
        float v[NCH][8];
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = gl + k * G;
            const bool on = ch < chunks;
            const int e = 8 * (on ? ch : 0);
            const Frag<T> f = frag_load<T>(merge ? (src[e / Cs] + (e % Cs)) : (src[0] + e));
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[k][i] = on ? to_f32(f.v[i]) : 0.f; s1 += v[k][i]; }
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
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = gl + k * G;
            if (valid && ch < chunks) {
                const int e = 8 * ch;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + e), w1 = *reinterpret_cast<const f32x4*>(w + e + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + e), b1 = *reinterpret_cast<const f32x4*>(b + e + 4);
                Frag<T> o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o.v[i] = from_f32<T>(fmaf((v[k][i] - mean) * rstd, w0[i], b0[i]));
                    o.v[i + 4] = from_f32<T>(fmaf((v[k][i + 4] - mean) * rstd, w1[i], b1[i]));
                }
                frag_store<T>(y + row * C + e, o);
            }
        }
    }
}

template <typename T>
int launch_ln_rows(const void* x, const float* w, const float* b, void* y, long long rows, int C, float eps, int merge, int H,
                   int W, hipStream_t st) {
    const int chunks = C / 8;
    int G = 16;
    while (G < chunks && G < 64) G <<= 1;
    const int nch = (chunks + G - 1) / G;
    const long long nrb = (rows + (64 / G) - 1) / (64 / G);
    const int nb = (int)((nrb + 3) / 4 < 4096 ? (nrb + 3) / 4 : 4096);
#define MTMP_LN_CASE(g, n)                                                                                          \
    if (G == g && nch == n) {                                                                                       \
        hipLaunchKernelGGL((ln_rows_kernel<T, g, n>), dim3(nb), dim3(256), 0, st, (const T*)x, w, b, (T*)y, rows, C, eps, \
                           merge, H, W);                                                                            \
        return MTMP_OK;                                                                                             \
    }
    MTMP_LN_CASE(16, 1) MTMP_LN_CASE(32, 1) MTMP_LN_CASE(64, 1) MTMP_LN_CASE(64, 2) MTMP_LN_CASE(64, 3)
#undef MTMP_LN_CASE
    mtmp_set_error("mtmp_layernorm_rows: unsupported C=%d", C);
    return MTMP_ERR_ARG;
}

// ------------------------------------------------------------------------------------------
template <typename T> MTMP_DEV void store_pair2(T* p, T a, T b);
template <> MTMP_DEV void store_pair2<bf16>(bf16* p, bf16 a, bf16 b) { *reinterpret_cast<bf16x2*>(p) = bf16x2{a, b}; }
template <> MTMP_DEV void store_pair2<float>(float* p, float a, float b) { *reinterpret_cast<f32x2*>(p) = f32x2{a, b}; }

template <typename T>
__global__ __launch_bounds__(256) void swin_wattn_kernel(const T* qkv, const T* table, T* out, int n_img, int H, int W,
                                                         int C, int heads, int shift, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    T* sVt = reinterpret_cast<T*>(smem_raw) + wave * DH * LDV;          // this wave's [32 d][LDV keys]
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
    // token t of this window -> element offset of its pixel in the UN-shifted map (roll by -shift, :160-161)
    auto pix = [&](int t) -> long long {
        const int ty = t / WS, tx = t - ty * WS;
        int yy = wi * WS + ty + shift, xx = wj * WS + tx + shift;
        if (yy >= H) yy -= H;
        if (xx >= W) xx -= W;
        return ((long long)img * H + yy) * W + xx;
    };
    // ---- Q (B operand: natural token order) and K (A operand: rows through swz23), straight from global
    Frag<T> qf[2][2], kf[2][2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int tq = 32 * blk + r, tk = 32 * blk + swz23(r);
        const T* pq = qkv + pix(tq < L ? tq : 0) * C3 + head * DH + 8 * half;
        const T* pk = qkv + pix(tk < L ? tk : 0) * C3 + C + head * DH + 8 * half;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            qf[blk][c] = frag_keep(frag_load<T>(pq + 16 * c), live && tq < L);
            kf[blk][c] = frag_keep(frag_load<T>(pk + 16 * c), live && tk < L);
        }
    }
    // ---- V -> LDS transposed: lane = (key pair, 8-dim group), two passes over the 32 dims
    {
        const int kp = (lane & 31) * 2, dg = (lane >> 5) * 8;
        const T* pa = qkv + pix(kp < L ? kp : 0) * C3 + 2 * C + head * DH + dg;
        const T* pb = qkv + pix(kp + 1 < L ? kp + 1 : 0) * C3 + 2 * C + head * DH + dg;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const Frag<T> fa = frag_keep(frag_load<T>(pa + 16 * ps), live && kp < L);
            const Frag<T> fb = frag_keep(frag_load<T>(pb + 16 * ps), live && kp + 1 < L);
#pragma unroll
            for (int e = 0; e < 8; ++e) store_pair2<T>(sVt + (dg + 16 * ps + e) * LDV + kp, fa.v[e], fb.v[e]);
        }
    }
    __syncthreads();
    // ---- S^T = K Q^T (rows = keys, cols = queries on lanes), + scale, + bias/mask table
    f32x16 st[2][2] = {{{0}, {0}}, {{0}, {0}}};                  // [key block][query block]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int c = 0; c < 2; ++c) mma<T>(st[kb][qb], kf[kb][c], qf[qb][c]);
    const T* tab = table + ((size_t)type * heads + head) * LP * LP;
    float linv[2];
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
        linv[qb] = 1.0f / l;
    }
    // ---- O^T = V^T P^T (rows = head dims, cols = queries)
    f32x16 o[2] = {{0}, {0}};
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                mma<T>(o[qb], frag_load<T>(sVt + r * LDV + 32 * kb + 16 * s + 8 * half), frag_from_acc<T>(st[kb][qb], s));
    // ---- write back to the token's own pixel (window reverse + roll back are the same address map)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int tq = 32 * qb + r;
        if (live && tq < L) {
            T* po = out + pix(tq) * C + head * DH + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                store4<T>(po + 8 * g, o[qb][4 * g] * linv[qb], o[qb][4 * g + 1] * linv[qb], o[qb][4 * g + 2] * linv[qb],
                          o[qb][4 * g + 3] * linv[qb]);
        }
    }
}

}  // namespace

// y[rows,C] = LayerNorm(x rows; w, b, eps) in `dtype`; w,b fp32.  merge != 0: x is an NHWC map
// [n,H,W,C/4] and row (img,i,j) is the patch-merging concat of its 2x2 neighbourhood
// (swin_transformer.py:34-44, order x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2]); rows = n*(H/2)*(W/2).
extern "C" int mtmp_layernorm_rows(int dtype, const void* x, const float* w, const float* b, void* y, long long rows,
                                   int C, float eps, int merge, int H, int W, void* stream) {
    MTMP_CHECK_ARG(x && w && b && y && rows > 0, "mtmp_layernorm_rows: bad pointer / rows");
    MTMP_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 1536 && (!merge || (C % 32 == 0 && H % 2 == 0 && W % 2 == 0)),
                   "mtmp_layernorm_rows: bad shape C=%d merge=%d H=%d W=%d", C, merge, H, W);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dtype == 0) rc = launch_ln_rows<float>(x, w, b, y, rows, C, eps, merge, H, W, st);
    else if (dtype == 1) rc = launch_ln_rows<bf16>(x, w, b, y, rows, C, eps, merge, H, W, st);
    else { mtmp_set_error("mtmp_layernorm_rows: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    if (rc) return rc;
    MTMP_CHECK_LAUNCH("mtmp_layernorm_rows");
    return MTMP_OK;
}

// out[n,H,W,C] = shifted-window attention of qkv[n,H,W,3C] (q|k|v, head h = 32 columns); window 7,
// head_dim 32 (C = 32*heads), H % 7 == W % 7 == 0; table [4][heads][64][64] in `dtype` (see header).
extern "C" int mtmp_swin_window_attn(int dtype, const void* qkv, const void* table, void* out, int n_img, int H, int W,
                                     int C, int heads, int shift, float scale, void* stream) {
    MTMP_CHECK_ARG(qkv && table && out, "mtmp_swin_window_attn: null pointer");
    MTMP_CHECK_ARG(n_img > 0 && H > 0 && W > 0 && H % WS == 0 && W % WS == 0 && heads > 0 && C == heads * DH && shift >= 0 &&
                       shift < WS, "mtmp_swin_window_attn: bad shape n=%d H=%d W=%d C=%d heads=%d shift=%d", n_img, H, W, C, heads, shift);
    const long long total = (long long)n_img * (H / WS) * (W / WS) * heads;
    const int nb = (int)((total + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(swin_wattn_kernel<float>, dim3(nb), dim3(256), 4 * DH * LDV * sizeof(float), st, (const float*)qkv,
                           (const float*)table, (float*)out, n_img, H, W, C, heads, shift, scale);
    else if (dtype == 1)
        hipLaunchKernelGGL(swin_wattn_kernel<bf16>, dim3(nb), dim3(256), 4 * DH * LDV * sizeof(bf16), st, (const bf16*)qkv,
                           (const bf16*)table, (bf16*)out, n_img, H, W, C, heads, shift, scale);
    else { mtmp_set_error("mtmp_swin_window_attn: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_swin_window_attn");
    return MTMP_OK;
}
