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
#include "common.hip.h"

namespace {

constexpr int WS = 7, L = 49, LP = 64, DH = 32, LDV = LP + 8;
constexpr float LOG2E = 1.4426950408889634f;

// ------------------------------------------------------------------------------------------
// A wave is split into 64/G groups of G lanes; a group normalises one row, a lane owns NCH chunks of
// 8 consecutive channels (16-byte loads/stores).  C = 96 -> 4 rows per wave, 192 -> 2, >= 384 -> 1.
template <typename T, int G, int NCH>
__global__ __launch_bounds__(256) void ln_rows_kernel(const T* x, const float* w, const float* b, T* y, long long rows_max,
                                                      int C, float eps, int merge, int H, int W, const int* rows_live) {
    constexpr int RPW = 64 / G;
    const long long rows = rows_live ? min(rows_max, (long long)*rows_live) : rows_max;
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
                   int W, const int* rows_live, hipStream_t st) {
    const int chunks = C / 8;
    int G = 16;
    while (G < chunks && G < 64) G <<= 1;
    const int nch = (chunks + G - 1) / G;
    const long long nrb = (rows + (64 / G) - 1) / (64 / G);
    const int nb = (int)((nrb + 3) / 4 < 4096 ? (nrb + 3) / 4 : 4096);
#define MTMP_LN_CASE(g, n)                                                                                          \
    if (G == g && nch == n) {                                                                                       \
        hipLaunchKernelGGL((ln_rows_kernel<T, g, n>), dim3(nb), dim3(256), 0, st, (const T*)x, w, b, (T*)y, rows, C, eps, \
                           merge, H, W, rows_live);                                                                 \
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
__global__ __launch_bounds__(256) void swin_wattn_kernel(const T* qkv, const T* table, T* out, int n_img_max, int H, int W,
                                                         int C, int heads, int shift, float scale, const int* rows_live) {
    const int n_img = rows_live ? min(n_img_max, *rows_live / (H * W)) : n_img_max;
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


// ------------------------------------------------------------------------------------------
// MLP half of a Swin block (swin_transformer.py:428-449: x + stochastic_depth(mlp(norm2(x)))) in ONE kernel for the
// narrow stages (C = 96 / 192, bf16), where the 4C-wide hidden activation is what the three-launch chain
// (mtmp_layernorm_rows, mtmp_gemm_nt + GELU, mtmp_gemm_nt + residual) moves through HBM: 154 MB written and read
// again per stage-1 block of 64 images against 38.5 MB of tokens.  Here it never leaves the registers:
//   * a workgroup owns 128 tokens (32 per wave); each wave normalises its rows in registers (nn.LayerNorm: biased
//     variance, eps inside the root) and keeps them as the KC = C/16 B-operand fragments of the first product;
//   * the hidden units are walked in PANELS of HP: rows [j HP, (j+1) HP) of W1 and the matching columns of W2 are the
//     only things staged in LDS (two buffers, one barrier per panel, next panel prefetched into registers);
//   * H^T = W1_j xn^T comes out of the MFMA with hidden units on the accumulator ROWS; W1's rows are read through
//     swz23() so that, after bias + GELU, accumulator registers 8s..8s+7 ARE the k-step-s operand fragment of the
//     second product (common.hip.h, "acc -> Frag") -- no LDS round trip between the two GEMMs;
//   * out^T += W2[:, panel] H accumulates over the panels; the epilogue adds b2, rounds, applies the per-image
//     StochasticDepth factor and the residual exactly like mtmp_gemm_nt's epilogue, through a wave-private LDS tile so
//     that HBM sees 64-byte row pieces.
#ifdef MTMP_LAB_CLOCK                                         // lab builds only (tools/dbg/swin_mlp_clock.py): s_memtime stamps of every 8th workgroup
__device__ long long mtmp_dbg_swin_stamps[256 * 4 * 32];
#define MLP_STAMP(k) do { if ((blockIdx.x & 7) == 0 && blockIdx.x < 2048 && lane == 0 && (k) < 32) mtmp_dbg_swin_stamps[((blockIdx.x >> 3) * 4 + wave) * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MLP_STAMP(k) do {} while (0)
#endif
template <int C, int HP> struct MlpGeom {
    static constexpr int KC = C / 16, OG = C / 32, HG = HP / 32, NPAN = 4 * C / HP;
    static constexpr int LD1 = C + 8, LD2 = HP + 8;               // LDS row strides (elements): 16 B pad, conflict-free b128 reads
    static constexpr int P1 = HP * LD1, P2 = C * LD2;             // elements per W1 / W2 panel
    static constexpr int CPR1 = C / 8, CPR2 = HP / 8;             // 16-byte chunks per panel row
    static constexpr int L1 = HP * CPR1 / 256, L2 = C * CPR2 / 256;   // loads per thread (3 + 3 for both shapes)
    static constexpr int FS = 40;                                 // staging row: 32 features + 16 B pad
    static constexpr size_t b1_off = (size_t)(2 * (P1 + P2) + 4 * 32 * FS) * sizeof(bf16);   // fc1's bias, 4 C floats
    static constexpr size_t lds_bytes = b1_off + (size_t)4 * C * sizeof(float);
    static_assert(HP * CPR1 % 256 == 0 && C * CPR2 % 256 == 0, "panel chunks must divide over 256 threads");
    static_assert(b1_off % 16 == 0, "bias block must be 16-byte aligned");
};
template <int C, int HP> struct MlpRegs { u32x4_t a[MlpGeom<C, HP>::L1], b[MlpGeom<C, HP>::L2]; };

template <int C, int HP>
MTMP_DEV void mlp_fetch(MlpRegs<C, HP>& g, const bf16* w1, const bf16* w2, int j, int tid) {
    using G = MlpGeom<C, HP>;
#pragma unroll
    for (int i = 0; i < G::L1; ++i) {            // W1 panel: HP consecutive rows of [4C][C] = one contiguous block
        const int id = i * 256 + tid;
        g.a[i] = *reinterpret_cast<const u32x4_t*>(w1 + (size_t)j * HP * C + 8 * id);
    }
#pragma unroll
    for (int i = 0; i < G::L2; ++i) {            // W2 panel: columns [j HP, (j+1) HP) of every row of [C][4C]
        const int id = i * 256 + tid, row = id / G::CPR2, ch = id % G::CPR2;
        g.b[i] = *reinterpret_cast<const u32x4_t*>(w2 + (size_t)row * 4 * C + j * HP + 8 * ch);
    }
}
template <int C, int HP>
MTMP_DEV void mlp_commit(bf16* s1, bf16* s2, const MlpRegs<C, HP>& g, int tid) {
    using G = MlpGeom<C, HP>;
#pragma unroll
    for (int i = 0; i < G::L1; ++i) {
        const int id = i * 256 + tid, row = id / G::CPR1, ch = id % G::CPR1;
        *reinterpret_cast<u32x4_t*>(s1 + row * G::LD1 + 8 * ch) = g.a[i];
    }
#pragma unroll
    for (int i = 0; i < G::L2; ++i) {
        const int id = i * 256 + tid, row = id / G::CPR2, ch = id % G::CPR2;
        *reinterpret_cast<u32x4_t*>(s2 + row * G::LD2 + 8 * ch) = g.b[i];
    }
}

template <int C, int HP>
__global__ __launch_bounds__(256, 2) void swin_mlp_kernel(const bf16* x, const float* ln_w, const float* ln_b, const bf16* w1,
                                                          const float* b1, const bf16* w2, const float* b2,
                                                          const float* row_scale, int rows_per_scale, bf16* y, long long M_max,
                                                          float eps, const int* rows_live) {
    using G = MlpGeom<C, HP>;
    const long long M = rows_live ? min(M_max, (long long)*rows_live) : M_max;
    if ((long long)blockIdx.x * 128 >= M) return;                 // (workgroup-uniform, in front of every barrier)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16* sP = reinterpret_cast<bf16*>(smem_raw);                                  // [2][P1 + P2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    bf16* sS = sP + 2 * (G::P1 + G::P2) + wave * 32 * G::FS;                       // wave-private [32][FS]
    float* sB1 = reinterpret_cast<float*>(smem_raw + G::b1_off);                   // fc1's bias (read per hidden group in the loop)
    const long long m_wave = (long long)blockIdx.x * 128 + wave * 32;
    const long long row = min(m_wave + r, M - 1);
    MlpRegs<C, HP> preg;
    MLP_STAMP(0);
    mlp_fetch<C, HP>(preg, w1, w2, 0, tid);
    for (int i = tid; i < 4 * C; i += 256) sB1[i] = b1[i];                         // (visible behind the barrier in front of the loop)
    // ---- LayerNorm prologue in registers: lane (r, half) holds channels 16c + 8 half + j of token r
    Frag<bf16> af[G::KC];
    const bf16* xrow = x + row * C + 8 * half;
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < G::KC; ++c) {
        af[c] = frag_load<bf16>(xrow + 16 * c);
#pragma unroll
        for (int j = 0; j < 8; ++j) s1 += to_f32(af[c].v[j]);
    }
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * (1.0f / C);
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < G::KC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = to_f32(af[c].v[j]) - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = rsqrtf(s2 * (1.0f / C) + eps);
#pragma unroll
    for (int c = 0; c < G::KC; ++c) {
        const int k = 16 * c + 8 * half;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(ln_w + k), g1 = *reinterpret_cast<const f32x4*>(ln_w + k + 4);
        const f32x4 o0 = *reinterpret_cast<const f32x4*>(ln_b + k), o1 = *reinterpret_cast<const f32x4*>(ln_b + k + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[c].v[i] = from_f32<bf16>(fmaf((to_f32(af[c].v[i]) - mean) * rstd, g0[i], o0[i]));
            af[c].v[i + 4] = from_f32<bf16>(fmaf((to_f32(af[c].v[i + 4]) - mean) * rstd, g1[i], o1[i]));
        }
    }
    MLP_STAMP(1);
    mlp_commit<C, HP>(sP, sP + G::P1, preg, tid);
    mlp_fetch<C, HP>(preg, w1, w2, 1, tid);
    f32x16 acc2[G::OG];
#pragma unroll
    for (int o = 0; o < G::OG; ++o) acc2[o] = f32x16{0};
    __syncthreads();
    MLP_STAMP(2);
    for (int j = 0; j < G::NPAN; ++j) {
        const bf16* c1 = sP + (j & 1) * (G::P1 + G::P2);
        const bf16* c2 = c1 + G::P1;
        bf16* n1 = sP + ((j & 1) ^ 1) * (G::P1 + G::P2);
#pragma unroll
        for (int g = 0; g < G::HG; ++g) {
            // Every LDS operand of the group is requested BEFORE its first MFMA: W1's KC fragments, the bias block (staged in LDS by
            // the prologue -- as global loads at this point they put an s_waitcnt vmcnt(0), i.e. the whole L2 latency AND the wait
            // for the next panel's prefetch, in front of the first MFMA of every group) and W2's 2 OG fragments, which do not depend
            // on the hidden tile and land under the first product and the GELU.  Left to itself hipcc keeps one fragment register
            // and emits ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma 24 times per panel: ~130 cycles of LDS latency per MFMA.
            // accumulator register t of this lane = hidden unit j HP + 32 g + 16 (t >> 3) + 8 half + (t & 7)
            const bf16* wrow = c1 + (32 * g + swz23(r)) * G::LD1 + 8 * half;
            Frag<bf16> wf[G::KC], vf[G::OG][2];
#pragma unroll
            for (int c = 0; c < G::KC; ++c) wf[c] = frag_load<bf16>(wrow + 16 * c);
            const float* bp = sB1 + j * HP + 32 * g + 8 * half;
            const f32x4 ba = *reinterpret_cast<const f32x4*>(bp), bb = *reinterpret_cast<const f32x4*>(bp + 4);
            const f32x4 bc = *reinterpret_cast<const f32x4*>(bp + 16), bd = *reinterpret_cast<const f32x4*>(bp + 20);
#pragma unroll
            for (int o = 0; o < G::OG; ++o) {
                const bf16* vrow = c2 + (32 * o + r) * G::LD2 + 32 * g + 8 * half;
                vf[o][0] = frag_load<bf16>(vrow);
                vf[o][1] = frag_load<bf16>(vrow + 16);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, G::KC + 4 + 2 * G::OG, 0);      // (all DS reads of the group first)
            f32x16 h = {ba[0], ba[1], ba[2], ba[3], bb[0], bb[1], bb[2], bb[3], bc[0], bc[1], bc[2], bc[3], bd[0], bd[1], bd[2], bd[3]};
#pragma unroll
            for (int c = 0; c < G::KC; ++c) mma<bf16>(h, wf[c], af[c]);
#pragma unroll
            for (int t = 0; t < 16; ++t) h[t] = gelu<bf16>(h[t]);
            const Frag<bf16> h0 = frag_from_acc<bf16>(h, 0), h1 = frag_from_acc<bf16>(h, 1);
#pragma unroll
            for (int o = 0; o < G::OG; ++o) {
                mma<bf16>(acc2[o], vf[o][0], h0);
                mma<bf16>(acc2[o], vf[o][1], h1);
            }
        }
        if (j < 6) MLP_STAMP(3 + 3 * j);
        mlp_commit<C, HP>(n1, n1 + G::P1, preg, tid);               // panel j+1 (or a harmless repeat of the last one)
        mlp_fetch<C, HP>(preg, w1, w2, min(j + 2, G::NPAN - 1), tid);
        if (j < 6) MLP_STAMP(4 + 3 * j);
        __syncthreads();
        if (j < 6) MLP_STAMP(5 + 3 * j);
    }
    MLP_STAMP(21);
    // ---- epilogue: acc2[o] register t = output feature 32 o + acc_row(t, half) of token r
    const int tok = lane >> 2, ch = lane & 3;
    // every residual piece (and the two rows' StochasticDepth factors) is requested up front: inside the loop below -- whose waits
    // are memory barriers for the compiler -- each of the 2 OG global loads was a round trip of its own in front of its store
    Frag<bf16> resv[G::OG][2];
    float rsc[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const long long grow = min(m_wave + tok + 16 * ps, M - 1);
        rsc[ps] = row_scale ? row_scale[grow / rows_per_scale] : 1.0f;
#pragma unroll
        for (int o = 0; o < G::OG; ++o) resv[o][ps] = frag_load<bf16>(x + grow * C + 32 * o + 8 * ch);
    }
    f32x4 b2v[G::OG][4];
#pragma unroll
    for (int o = 0; o < G::OG; ++o)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) b2v[o][i4] = *reinterpret_cast<const f32x4*>(b2 + 32 * o + 8 * i4 + 4 * half);
#ifdef MTMP_LAB_CLOCK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MLP_STAMP(22);
#endif
#pragma unroll
    for (int o = 0; o < G::OG; ++o) {
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const f32x4 bv = b2v[o][i4];
            store4<bf16>(sS + r * G::FS + 8 * i4 + 4 * half, acc2[o][4 * i4] + bv[0], acc2[o][4 * i4 + 1] + bv[1],
                         acc2[o][4 * i4 + 2] + bv[2], acc2[o][4 * i4 + 3] + bv[3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int t = tok + 16 * ps;
            const long long grow = min(m_wave + t, M - 1);
            const Frag<bf16> v = frag_load<bf16>(sS + t * G::FS + 8 * ch);
            const Frag<bf16> res = resv[o][ps];
            const float rsv = rsc[ps];
            Frag<bf16> out;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float f = to_f32(v.v[i]);
                if (row_scale) f *= rsv;
                out.v[i] = from_f32<bf16>(round_as<bf16>(f) + to_f32(res.v[i]));
            }
            frag_store<bf16>(y + grow * C + 32 * o + 8 * ch, out);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#ifdef MTMP_LAB_CLOCK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MLP_STAMP(23);
#endif
}
#ifdef MTMP_LAB_CLOCK
}  // namespace
extern "C" int mtmp_dbg_read_swin_stamps(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mtmp_dbg_swin_stamps), (size_t)n * sizeof(long long));
}
namespace {
#endif

template <int C, int HP>
int launch_swin_mlp(const void* x, const float* ln_w, const float* ln_b, const void* w1, const float* b1, const void* w2,
                    const float* b2, const float* row_scale, int rows_per_scale, void* y, long long M, float eps,
                    const int* rows_live, hipStream_t st) {
    using G = MlpGeom<C, HP>;
    const void* fn = (const void*)swin_mlp_kernel<C, HP>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes) != hipSuccess) {
        mtmp_set_error("mtmp_swin_mlp: cannot raise dynamic LDS to %zu", G::lds_bytes);
        return MTMP_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((swin_mlp_kernel<C, HP>), dim3((unsigned)((M + 127) / 128)), dim3(256), G::lds_bytes, st, (const bf16*)x,
                       ln_w, ln_b, (const bf16*)w1, b1, (const bf16*)w2, b2, row_scale, rows_per_scale, (bf16*)y, M, eps, rows_live);
    return MTMP_OK;
}

// ------------------------------------------------------------------------------------------
// y[M,N] = LayerNorm(x[M,C]) W[N,C]^T + bias for the narrow stages (C = 96 / 192, bf16): norm1 -> qkv of a Swin block
// (swin_transformer.py:428-449 + :115-225) in one launch.  At K = C the projection is bound by writing its 3C-wide
// output, and the separate LayerNorm pass reads and writes the token map once more for nothing.  Every wave is on its
// own: it normalises 32 tokens in registers (the KC = C/16 B-operand fragments), then walks the output features in
// groups of 32, reading the weight fragments straight from global memory -- W is 55 / 221 KB and stays in L2, there is
// nothing to share through LDS and no barrier -- and stores each 32 x 32 tile through a wave-private LDS tile as
// 64-byte row pieces.  The next group's weight fragments are requested before the current group's stores.
template <int C>
__global__ __launch_bounds__(256, 3) void swin_ln_linear_kernel(const bf16* x, const float* ln_w, const float* ln_b, const bf16* w,
                                                                const float* bias, bf16* y, long long M_max, int N, float eps,
                                                                const int* rows_live) {
    constexpr int KC = C / 16, FS = 40;
    const long long M = rows_live ? min(M_max, (long long)*rows_live) : M_max;
    __shared__ __attribute__((aligned(16))) bf16 stage[4 * 32 * FS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    bf16* sS = stage + wave * 32 * FS;
    const long long m_wave = (long long)blockIdx.x * 128 + wave * 32;
    if (m_wave >= M) return;                                     // whole wave out of range (no barriers in this kernel)
    const long long row = min(m_wave + r, M - 1);
    const int ngroups = N / 32;
    Frag<bf16> wf[KC];
    const bf16* wlane = w + (size_t)r * C + 8 * half;
#pragma unroll
    for (int c = 0; c < KC; ++c) wf[c] = frag_load<bf16>(wlane + 16 * c);
    Frag<bf16> af[KC];
    const bf16* xrow = x + row * C + 8 * half;
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        af[c] = frag_load<bf16>(xrow + 16 * c);
#pragma unroll
        for (int j = 0; j < 8; ++j) s1 += to_f32(af[c].v[j]);
    }
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * (1.0f / C);
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = to_f32(af[c].v[j]) - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = rsqrtf(s2 * (1.0f / C) + eps);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const int k = 16 * c + 8 * half;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(ln_w + k), g1 = *reinterpret_cast<const f32x4*>(ln_w + k + 4);
        const f32x4 o0 = *reinterpret_cast<const f32x4*>(ln_b + k), o1 = *reinterpret_cast<const f32x4*>(ln_b + k + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[c].v[i] = from_f32<bf16>(fmaf((to_f32(af[c].v[i]) - mean) * rstd, g0[i], o0[i]));
            af[c].v[i + 4] = from_f32<bf16>(fmaf((to_f32(af[c].v[i + 4]) - mean) * rstd, g1[i], o1[i]));
        }
    }
    const int tok = lane >> 2, ch = lane & 3;
    for (int g = 0; g < ngroups; ++g) {
        // accumulator register t = output feature 32 g + acc_row(t, half) of token r; start from the bias
        f32x16 acc;
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bias) bv = *reinterpret_cast<const f32x4*>(bias + 32 * g + 8 * i4 + 4 * half);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[4 * i4 + i] = bv[i];
        }
#pragma unroll
        for (int c = 0; c < KC; ++c) mma<bf16>(acc, wf[c], af[c]);
        const bf16* wnext = wlane + (size_t)32 * min(g + 1, ngroups - 1) * C;
#pragma unroll
        for (int c = 0; c < KC; ++c) wf[c] = frag_load<bf16>(wnext + 16 * c);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
            store4<bf16>(sS + r * FS + 8 * i4 + 4 * half, acc[4 * i4], acc[4 * i4 + 1], acc[4 * i4 + 2], acc[4 * i4 + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int t = tok + 16 * ps;
            const long long grow = min(m_wave + t, M - 1);
            const u32x4_t d = *reinterpret_cast<const u32x4_t*>(sS + t * FS + 8 * ch);
            *reinterpret_cast<u32x4_t*>(y + grow * N + 32 * g + 8 * ch) = d;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
// ------------------------------------------------------------------------------------------
// Attention half of a Swin block (swin_transformer.py:428-449: x + stochastic_depth(attn(norm1(x))), :115-225) in ONE kernel
// for the narrow stages (bf16, C = 96 / 192, maps that are multiples of the window): norm1, the qkv projection, the window
// attention, the output projection, the per-image StochasticDepth factor and the residual.  The four-launch chain
// (mtmp_swin_ln_linear, mtmp_swin_window_attn, mtmp_gemm_nt) moves the 3C-wide qkv map and the attention output through
// HBM (424 MB per stage-1 block of 64 images against 77 MB of tokens read and written); here a token is read once and
// written once.  One workgroup per window, one wave per head (head_dim 32):
//   * the window's 49 tokens (padded to 64: two 32-token blocks) are normalised once (nn.LayerNorm, two or four lanes per token)
//     into LDS, bf16 -- the B operand of every head's projections;
//   * Q^T, K^T = W_{q,k}[head] xn^T come out of the MFMA with the head dims on the accumulator ROWS and tokens on the lanes,
//     V = xn W_v[head]^T with tokens on the rows and head dims on the lanes: accumulator registers 8s..8s+7 of each ARE the
//     k-step-s operand fragments of S^T = K Q^T and O^T = V^T P^T (common.hip.h, "acc -> Frag": both operands of a product
//     come from accumulators with the same row order, so the contraction index pairs up) -- q, k, v never leave registers;
//   * the additive table (relative-position bias + shift mask, PAD_LOGIT on the 15 pad keys) is stored by the host with its
//     key columns in accumulator-register order, so a lane reads its 16 keys of a block as two 16-byte loads;
//   * the heads' outputs meet in the same LDS tile ([64 tokens][C], bf16 as the unfused chain rounds them), then wave w
//     computes output channels [32 w, 32 w + 32) of the projection and writes x + scale * (proj + bias) to the token's pixel.
template <int C>
__global__ __launch_bounds__(2 * C, 3) void swin_attn_block_kernel(const bf16* x, const float* ln_w, const float* ln_b, float eps,
                                                                   const bf16* wqkv, const float* bqkv, const bf16* table,
                                                                   const bf16* wproj, const float* bproj, const float* row_scale,
                                                                   bf16* out, int n_img_max, int H, int W, int shift, float scale,
                                                                   const int* rows_live) {
    constexpr int HEADS = C / DH, KC = C / 16, LDO = C + 8, UNR = KC >= 12 ? 4 : KC;     // (UNR: unroll of the k loops)
    __shared__ __attribute__((aligned(16))) bf16 sX[LP * LDO];    // normalised tokens, later the heads' outputs
    const int n_img = rows_live ? min(n_img_max, *rows_live / (H * W)) : n_img_max;
    const int nWh = H / WS, nWw = W / WS;
    const int img = blockIdx.x / (nWh * nWw), win = blockIdx.x - img * (nWh * nWw);
    if (img >= n_img) return;                                     // (whole workgroup: no barrier is skipped by part of it)
    const int tid = threadIdx.x, lane = tid & 63, head = tid >> 6, r = lane & 31, half = lane >> 5;
    const int wi = win / nWw, wj = win - wi * nWw;
    const int type = shift > 0 ? ((wi == nWh - 1 ? 2 : 0) + (wj == nWw - 1 ? 1 : 0)) : 0;
    auto pix = [&](int t) -> long long {                          // token of this window -> its pixel on the UN-shifted map
        const int ty = t / WS, tx = t - ty * WS;
        int yy = wi * WS + ty + shift, xx = wj * WS + tx + shift;
        if (yy >= H) yy -= H;
        if (xx >= W) xx -= W;
        return ((long long)img * H + yy) * W + xx;
    };
    // ---- norm1 of the window's tokens (waves 0 and 1: 32 tokens each, a lane pair per token) -> LDS, bf16
    // (wide rows: a wave normalises 16 tokens with four lanes per token, so that a lane keeps KC / 2 fragments, not KC)
    constexpr int LPT = KC >= 12 ? 4 : 2, TPW = 64 / LPT, KL = KC * 2 / LPT;      // lanes per token, tokens per wave, fragments per lane
    if (head < LP / TPW) {
        const int tq = TPW * head + (LPT == 2 ? r : (lane & 15)), part = LPT == 2 ? half : (lane >> 4);
        Frag<bf16> xf[KL];
        const bf16* xrow = x + pix(tq < L ? tq : 0) * C + 8 * part;
        float s1 = 0.f;
#pragma unroll
        for (int c = 0; c < KL; ++c) {
            xf[c] = frag_load<bf16>(xrow + 8 * LPT * c);
#pragma unroll
            for (int j = 0; j < 8; ++j) s1 += to_f32(xf[c].v[j]);
        }
        s1 += __shfl_xor(s1, 32, 64);
        if (LPT == 4) s1 += __shfl_xor(s1, 16, 64);
        const float mean = s1 * (1.0f / C);
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < KL; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = to_f32(xf[c].v[j]) - mean; s2 += d * d; }
        s2 += __shfl_xor(s2, 32, 64);
        if (LPT == 4) s2 += __shfl_xor(s2, 16, 64);
        const float rstd = rsqrtf(s2 * (1.0f / C) + eps);
#pragma unroll
        for (int c = 0; c < KL; ++c) {
            const int k = 8 * LPT * c + 8 * part;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(ln_w + k), g1 = *reinterpret_cast<const f32x4*>(ln_w + k + 4);
            const f32x4 o0 = *reinterpret_cast<const f32x4*>(ln_b + k), o1 = *reinterpret_cast<const f32x4*>(ln_b + k + 4);
            Frag<bf16> y;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                y.v[i] = from_f32<bf16>(fmaf((to_f32(xf[c].v[i]) - mean) * rstd, g0[i], o0[i]));
                y.v[i + 4] = from_f32<bf16>(fmaf((to_f32(xf[c].v[i + 4]) - mean) * rstd, g1[i], o1[i]));
            }
            frag_store<bf16>(sX + tq * LDO + k, frag_keep(y, tq < L));          // (pad tokens: zero rows)
        }
    }
    __syncthreads();
    // ---- this head's projections, from the bias: K^T [head dim][token] and V [token][head dim] first, then Q^T -- each
    //      becomes operand fragments as soon as it is complete (fewer live accumulators)
    const bf16* wl = wqkv + (size_t)(head * DH + r) * C + 8 * half;
    const bf16* xl = sX + r * LDO + 8 * half;
    Frag<bf16> kf[2][2], vf[2][2], qf[2][2];
    {
        f32x16 ka[2], va[2];
        const float bv = bqkv[2 * C + head * DH + r];
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const f32x4 bk = *reinterpret_cast<const f32x4*>(bqkv + C + head * DH + 8 * i4 + 4 * half);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ka[0][4 * i4 + i] = ka[1][4 * i4 + i] = bk[i];
                va[0][4 * i4 + i] = va[1][4 * i4 + i] = bv;
            }
        }
#pragma unroll UNR
        for (int c = 0; c < KC; ++c) {
            const Frag<bf16> wk = frag_load<bf16>(wl + (size_t)C * C + 16 * c), wv = frag_load<bf16>(wl + (size_t)2 * C * C + 16 * c);
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const Frag<bf16> xb = frag_load<bf16>(xl + 32 * tb * LDO + 16 * c);
                mma<bf16>(ka[tb], wk, xb);
                mma<bf16>(va[tb], xb, wv);
            }
        }
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) { kf[tb][s_] = frag_from_acc<bf16>(ka[tb], s_); vf[tb][s_] = frag_from_acc<bf16>(va[tb], s_); }
    }
    {
        f32x16 qa[2];
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bqkv + head * DH + 8 * i4 + 4 * half);
#pragma unroll
            for (int i = 0; i < 4; ++i) qa[0][4 * i4 + i] = qa[1][4 * i4 + i] = bq[i];
        }
#pragma unroll UNR
        for (int c = 0; c < KC; ++c) {
            const Frag<bf16> wq = frag_load<bf16>(wl + 16 * c);
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) mma<bf16>(qa[tb], wq, frag_load<bf16>(xl + 32 * tb * LDO + 16 * c));
        }
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) qf[tb][s_] = frag_from_acc<bf16>(qa[tb], s_);
    }
    __syncthreads();                                              // every wave is done with the normalised tokens
    // Requested HERE, a phase ahead of their use: the residual pieces of the epilogue (in a branch per token block they were eight
    // load -> s_waitcnt vmcnt(0) -> store round trips at the end of every workgroup: pad tokens read token 0 instead) and, where
    // the registers allow (C = 96), the projection's weight fragments (read one fragment ahead, every second MFMA of the
    // projection waited for an L2 round trip).
    f32x4 resv[2][4];
    long long pxv[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int tq = 32 * tb + r;
        pxv[tb] = pix(tq < L ? tq : 0);
        const bf16* xr = x + pxv[tb] * C + head * DH + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) resv[tb][g] = load4<bf16>(xr + 8 * g);
    }
    constexpr bool PREW = KC <= 6;
    const bf16* wpl = wproj + (size_t)(head * DH + r) * C + 8 * half;
    Frag<bf16> wpf[PREW ? KC : 1];
    if constexpr (PREW) {
#pragma unroll
        for (int c = 0; c < KC; ++c) wpf[c] = frag_load<bf16>(wpl + 16 * c);
    }
    // ---- per query block: S^T = K Q^T (rows = keys, cols = queries on lanes), * scale + bias / mask table, softmax over
    //      the rows, O^T = V^T P^T (rows = head dims), normalised, to LDS as [token][channel]
    const bf16* tab = table + ((size_t)type * HEADS + head) * LP * LP;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        f32x16 st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            st[kb] = mma0<bf16>(kf[kb][0], qf[qb][0]);
            mma<bf16>(st[kb], kf[kb][1], qf[qb][1]);
        }
        const bf16* trow = tab + (32 * qb + r) * LP + 8 * half;
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const Frag<bf16> t0 = frag_load<bf16>(trow + 32 * kb), t1 = frag_load<bf16>(trow + 32 * kb + 16);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                st[kb][t] = fmaf(st[kb][t], scale, to_f32(t0.v[t]));
                st[kb][t + 8] = fmaf(st[kb][t + 8], scale, to_f32(t1.v[t]));
                mx = fmaxf(mx, fmaxf(st[kb][t], st[kb][t + 8]));
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float pv = fast_exp2((st[kb][t] - mx) * LOG2E);
                st[kb][t] = pv;
                l += pv;
            }
        l += __shfl_xor(l, 32, 64);
        const float linv = 1.0f / l;
        f32x16 o = {0};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) mma<bf16>(o, vf[kb][s_], frag_from_acc<bf16>(st[kb], s_));
        bf16* po = sX + (32 * qb + r) * LDO + head * DH + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) store4<bf16>(po + 8 * g, o[4 * g] * linv, o[4 * g + 1] * linv, o[4 * g + 2] * linv, o[4 * g + 3] * linv);
    }
    __syncthreads();
    // ---- output projection: wave w owns channels [32 w, 32 w + 32); acc^T [channel][token]
    f32x16 pa[2];
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
        const f32x4 bp = *reinterpret_cast<const f32x4*>(bproj + head * DH + 8 * i4 + 4 * half);
#pragma unroll
        for (int i = 0; i < 4; ++i) pa[0][4 * i4 + i] = pa[1][4 * i4 + i] = bp[i];
    }
    if constexpr (PREW) {
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) mma<bf16>(pa[tb], wpf[c], frag_load<bf16>(xl + 32 * tb * LDO + 16 * c));
    } else {
#pragma unroll UNR
        for (int c = 0; c < KC; ++c) {
            const Frag<bf16> wp = frag_load<bf16>(wpl + 16 * c);
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) mma<bf16>(pa[tb], wp, frag_load<bf16>(xl + 32 * tb * LDO + 16 * c));
        }
    }
    const float sc = row_scale ? row_scale[img] : 1.0f;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int tq = 32 * tb + r;
        if (tq < L) {
            bf16* yo = out + pxv[tb] * C + head * DH + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 res = resv[tb][g];
                // (rounded where mtmp_gemm_nt's epilogue rounds: the projection, its scaled value, then the sum)
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = round_as<bf16>(round_as<bf16>(pa[tb][4 * g + i]) * sc) + res[i];
                store4<bf16>(yo + 8 * g, v[0], v[1], v[2], v[3]);
            }
        }
    }
}

}  // namespace

// y[rows,C] = LayerNorm(x rows; w, b, eps) in `dtype`; w,b fp32.  merge != 0: x is an NHWC map
// [n,H,W,C/4] and row (img,i,j) is the patch-merging concat of its 2x2 neighbourhood
// (swin_transformer.py:34-44, order x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2]); rows = n*(H/2)*(W/2).
// rows_live (every *_live entry; may be NULL): a DEVICE word with the rows in use (<= the row count argument) -- the frozen
// image encoder on a batch whose present images were moved to the front (mtmp_image_slots): buffers and grids keep the
// size of the whole batch (a captured hipGraph replays for any number of present images), rows past it are neither read nor
// written.
extern "C" int mtmp_layernorm_rows_live(int dtype, const void* x, const float* w, const float* b, void* y, long long rows,
                                        int C, float eps, int merge, int H, int W, const int32_t* rows_live, void* stream);
extern "C" int mtmp_layernorm_rows(int dtype, const void* x, const float* w, const float* b, void* y, long long rows,
                                   int C, float eps, int merge, int H, int W, void* stream) {
    return mtmp_layernorm_rows_live(dtype, x, w, b, y, rows, C, eps, merge, H, W, nullptr, stream);
}
extern "C" int mtmp_layernorm_rows_live(int dtype, const void* x, const float* w, const float* b, void* y, long long rows,
                                        int C, float eps, int merge, int H, int W, const int32_t* rows_live, void* stream) {
    MTMP_CHECK_ARG(x && w && b && y && rows > 0, "mtmp_layernorm_rows: bad pointer / rows");
    MTMP_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 1536 && (!merge || (C % 32 == 0 && H % 2 == 0 && W % 2 == 0)),
                   "mtmp_layernorm_rows: bad shape C=%d merge=%d H=%d W=%d", C, merge, H, W);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dtype == 0) rc = launch_ln_rows<float>(x, w, b, y, rows, C, eps, merge, H, W, rows_live, st);
    else if (dtype == 1) rc = launch_ln_rows<bf16>(x, w, b, y, rows, C, eps, merge, H, W, rows_live, st);
    else { mtmp_set_error("mtmp_layernorm_rows: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    if (rc) return rc;
    MTMP_CHECK_LAUNCH("mtmp_layernorm_rows");
    return MTMP_OK;
}

// out[n,H,W,C] = shifted-window attention of qkv[n,H,W,3C] (q|k|v, head h = 32 columns); window 7,
// head_dim 32 (C = 32*heads), H % 7 == W % 7 == 0; table [4][heads][64][64] in `dtype` (see header).
extern "C" int mtmp_swin_window_attn_live(int dtype, const void* qkv, const void* table, void* out, int n_img, int H, int W,
                                          int C, int heads, int shift, float scale, const int32_t* rows_live, void* stream);
extern "C" int mtmp_swin_window_attn(int dtype, const void* qkv, const void* table, void* out, int n_img, int H, int W,
                                     int C, int heads, int shift, float scale, void* stream) {
    return mtmp_swin_window_attn_live(dtype, qkv, table, out, n_img, H, W, C, heads, shift, scale, nullptr, stream);
}
// (rows_live counts token rows: live images = *rows_live / (H W))
extern "C" int mtmp_swin_window_attn_live(int dtype, const void* qkv, const void* table, void* out, int n_img, int H, int W,
                                          int C, int heads, int shift, float scale, const int32_t* rows_live, void* stream) {
    MTMP_CHECK_ARG(qkv && table && out, "mtmp_swin_window_attn: null pointer");
    MTMP_CHECK_ARG(n_img > 0 && H > 0 && W > 0 && H % WS == 0 && W % WS == 0 && heads > 0 && C == heads * DH && shift >= 0 &&
                       shift < WS, "mtmp_swin_window_attn: bad shape n=%d H=%d W=%d C=%d heads=%d shift=%d", n_img, H, W, C, heads, shift);
    const long long total = (long long)n_img * (H / WS) * (W / WS) * heads;
    const int nb = (int)((total + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL(swin_wattn_kernel<float>, dim3(nb), dim3(256), 4 * DH * LDV * sizeof(float), st, (const float*)qkv,
                           (const float*)table, (float*)out, n_img, H, W, C, heads, shift, scale, rows_live);
    else if (dtype == 1)
        hipLaunchKernelGGL(swin_wattn_kernel<bf16>, dim3(nb), dim3(256), 4 * DH * LDV * sizeof(bf16), st, (const bf16*)qkv,
                           (const bf16*)table, (bf16*)out, n_img, H, W, C, heads, shift, scale, rows_live);
    else { mtmp_set_error("mtmp_swin_window_attn: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_swin_window_attn");
    return MTMP_OK;
}

// y[M,C] = x + row_scale[row / rows_per_scale] * (gelu(LN(x; ln_w, ln_b, eps) W1^T + b1) W2^T + b2): the MLP half of a
// Swin block (swin_transformer.py:428-449; torchvision MLP keys mlp.0 / mlp.3) in one launch.  bf16 only (dtype 1),
// C = 96 or 192 (stages 1-2; wider stages use mtmp_layernorm_rows + mtmp_gemm_nt); w1 [4C,C], w2 [C,4C] bf16;
// ln_w, ln_b, b1, b2 fp32; row_scale (per-image StochasticDepth factor) may be NULL; y must not alias x.
extern "C" int mtmp_swin_mlp_live(int dtype, const void* x, const float* ln_w, const float* ln_b, const void* w1, const float* b1,
                                  const void* w2, const float* b2, const float* row_scale, int rows_per_scale, void* y,
                                  long long M, int C, float eps, const int32_t* rows_live, void* stream);
extern "C" int mtmp_swin_mlp(int dtype, const void* x, const float* ln_w, const float* ln_b, const void* w1, const float* b1,
                             const void* w2, const float* b2, const float* row_scale, int rows_per_scale, void* y,
                             long long M, int C, float eps, void* stream) {
    return mtmp_swin_mlp_live(dtype, x, ln_w, ln_b, w1, b1, w2, b2, row_scale, rows_per_scale, y, M, C, eps, nullptr, stream);
}
extern "C" int mtmp_swin_mlp_live(int dtype, const void* x, const float* ln_w, const float* ln_b, const void* w1, const float* b1,
                                  const void* w2, const float* b2, const float* row_scale, int rows_per_scale, void* y,
                                  long long M, int C, float eps, const int32_t* rows_live, void* stream) {
    MTMP_CHECK_ARG(x && ln_w && ln_b && w1 && b1 && w2 && b2 && y && x != y, "mtmp_swin_mlp: null / aliased pointer");
    MTMP_CHECK_ARG(dtype == 1 && (C == 96 || C == 192) && M > 0 && (!row_scale || rows_per_scale > 0),
                   "mtmp_swin_mlp: bf16 with C = 96 or 192 only (dtype=%d C=%d M=%lld)", dtype, C, M);
    hipStream_t st = (hipStream_t)stream;
    const int rc = C == 96 ? launch_swin_mlp<96, 64>(x, ln_w, ln_b, w1, b1, w2, b2, row_scale, rows_per_scale, y, M, eps, rows_live, st)
                           : launch_swin_mlp<192, 32>(x, ln_w, ln_b, w1, b1, w2, b2, row_scale, rows_per_scale, y, M, eps, rows_live, st);
    if (rc) return rc;
    MTMP_CHECK_LAUNCH("mtmp_swin_mlp");
    return MTMP_OK;
}

// y[M,N] = LayerNorm(x[M,C]; ln_w, ln_b, eps) W[N,C]^T + bias: norm1 + the qkv projection of a Swin block
// (swin_transformer.py:428-449, :115-225) in one launch.  bf16 only (dtype 1), C = 96 or 192, N % 32 == 0; W bf16,
// ln_w / ln_b / bias fp32 (bias may be NULL).
extern "C" int mtmp_swin_ln_linear_live(int dtype, const void* x, const float* ln_w, const float* ln_b, const void* w,
                                        const float* bias, void* y, long long M, int C, int N, float eps, const int32_t* rows_live,
                                        void* stream);
extern "C" int mtmp_swin_ln_linear(int dtype, const void* x, const float* ln_w, const float* ln_b, const void* w,
                                   const float* bias, void* y, long long M, int C, int N, float eps, void* stream) {
    return mtmp_swin_ln_linear_live(dtype, x, ln_w, ln_b, w, bias, y, M, C, N, eps, nullptr, stream);
}
extern "C" int mtmp_swin_ln_linear_live(int dtype, const void* x, const float* ln_w, const float* ln_b, const void* w,
                                        const float* bias, void* y, long long M, int C, int N, float eps, const int32_t* rows_live,
                                        void* stream) {
    MTMP_CHECK_ARG(x && ln_w && ln_b && w && y, "mtmp_swin_ln_linear: null pointer");
    MTMP_CHECK_ARG(dtype == 1 && (C == 96 || C == 192) && M > 0 && N > 0 && N % 32 == 0,
                   "mtmp_swin_ln_linear: bf16 with C = 96 or 192 and N %% 32 == 0 only (dtype=%d C=%d N=%d M=%lld)", dtype, C, N, M);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((M + 127) / 128));
    if (C == 96)
        hipLaunchKernelGGL(swin_ln_linear_kernel<96>, grid, dim3(256), 0, st, (const bf16*)x, ln_w, ln_b, (const bf16*)w, bias,
                           (bf16*)y, M, N, eps, rows_live);
    else
        hipLaunchKernelGGL(swin_ln_linear_kernel<192>, grid, dim3(256), 0, st, (const bf16*)x, ln_w, ln_b, (const bf16*)w, bias,
                           (bf16*)y, M, N, eps, rows_live);
    MTMP_CHECK_LAUNCH("mtmp_swin_ln_linear");
    return MTMP_OK;
}

// out = x + row_scale[image] * (proj(window_attention(qkv(norm1(x)))) + b_proj): the attention half of a Swin block
// (swin_transformer.py:428-449 with :115-225) in one launch.  bf16 only (dtype 1); x, out [n_img, H, W, C] (out != x), C = 96
// or 192 (heads = C / 32), H and W multiples of 7; wqkv [3C][C], wproj [C][C] bf16; ln_w, ln_b, bqkv [3C], bproj [C] fp32;
// table [4][heads][64][64] bf16 as mtmp_swin_window_attn's but with the KEY columns of every 16-key group in accumulator-register
// order (position 8 h + j of group g holds key 16 g + (j & 3) + 8 (j >> 2) + 4 h); row_scale: fp32[n_img] or NULL;
// rows_live: NULL or the device word of mtmp_image_slots (rows of the [n_img H W] map in use).
extern "C" int mtmp_swin_attn_block(int dtype, const void* x, const float* ln_w, const float* ln_b, float eps, const void* wqkv,
                                    const float* bqkv, const void* table, const void* wproj, const float* bproj,
                                    const float* row_scale, void* out, int n_img, int H, int W, int C, int heads, int shift,
                                    float scale, const int32_t* rows_live, void* stream) {
    MTMP_CHECK_ARG(x && ln_w && ln_b && wqkv && bqkv && table && wproj && bproj && out && out != x,
                   "mtmp_swin_attn_block: null pointer / in place");
    MTMP_CHECK_ARG(dtype == 1 && (C == 96 || C == 192) && heads * DH == C && n_img > 0 && H > 0 && W > 0 && H % WS == 0 &&
                       W % WS == 0 && shift >= 0 && shift < WS,
                   "mtmp_swin_attn_block: bf16 with C = 96 or 192, heads = C / 32, maps that are multiples of 7 only "
                   "(dtype=%d C=%d heads=%d n=%d H=%d W=%d shift=%d)", dtype, C, heads, n_img, H, W, shift);
    const long long nwg = (long long)n_img * (H / WS) * (W / WS);
    MTMP_CHECK_ARG(nwg < (1ll << 31), "mtmp_swin_attn_block: too many windows");
    hipStream_t st = (hipStream_t)stream;
    if (C == 96)
        hipLaunchKernelGGL(swin_attn_block_kernel<96>, dim3((unsigned)nwg), dim3(192), 0, st, (const bf16*)x, ln_w, ln_b, eps,
                           (const bf16*)wqkv, bqkv, (const bf16*)table, (const bf16*)wproj, bproj, row_scale, (bf16*)out, n_img, H,
                           W, shift, scale, rows_live);
    else
        hipLaunchKernelGGL(swin_attn_block_kernel<192>, dim3((unsigned)nwg), dim3(384), 0, st, (const bf16*)x, ln_w, ln_b, eps,
                           (const bf16*)wqkv, bqkv, (const bf16*)table, (const bf16*)wproj, bproj, row_scale, (bf16*)out, n_img, H,
                           W, shift, scale, rows_live);
    MTMP_CHECK_LAUNCH("mtmp_swin_attn_block");
    return MTMP_OK;
}
