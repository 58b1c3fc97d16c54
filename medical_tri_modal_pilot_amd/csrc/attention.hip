// Modality-aware multi-head attention for gfx950, forward + backward.
//
// Replaces builder/models/src/transformer/attention.py:24-84 (ScaledDotProductAttention +
// the head split/merge of MultiHeadAttention) together with the key-pad mask of
// builder/models/src/transformer/utils.py:79-125:
//     per (b, h):  O = softmax(Q K^T / sqrt(64) + keymask(kv_len[b])) V
// No N x N tensor is materialised; keys j >= kv_len[b] are never read.  A fully
// masked sample (kv_len == 0) reproduces the reference's masked_fill(-65504) +
// softmax result, i.e. the uniform average over all N keys.
//
// Layout: Q/K/V/O are [B, N, ld] with head h at columns [64h, 64h+64) ("head-major
// free": the reference's [H*B, N, 64] permute/contiguous copies are not observable).
// LSE is [B, H, N] fp32 in log2 units of the scaled score (internal format).
//
// Tiling (wave64, MFMA 32x32): one workgroup = 4 waves = 128 query rows (fwd, dQ)
// or 128 keys (dK/dV); K/V (or Q/dO) tiles of 64 rows are staged through LDS.
// The score tile is computed transposed (S^T = K Q^T) so that a query is a lane:
// row max / row sum are in-register, the online-softmax rescale of O^T is a
// per-lane multiply, and P^T is consumed by the P.V MFMA directly from the
// accumulator registers (common.cuh: acc -> Frag).
#include "common.cuh"
#include <math.h>

// Diagnostic build only (make stamp): s_memtime stamps around the phases of the forward loop.
#ifdef MTMP_STAMP
__device__ unsigned long long g_stamp[8];
#define STAMP(var)                                                                                   \
    {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    }
#define STAMP_DECL unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, sa = 0, sb = 0, sc = 0, sd = 0;
#define STAMP_ACC  { sa += ts1 - ts0; sb += ts2 - ts1; sc += ts3 - ts2; sd += ts4 - ts3; }
#else
#define STAMP(var)
#define STAMP_DECL
#define STAMP_ACC
#endif

namespace {

constexpr int DH = 64;       // head dim: d_model 256 / 4 heads (tri_mbt_vsltcls.py:29-30)
constexpr int KT = 64;       // rows per LDS tile
constexpr int LDT = DH + 8;  // padded LDS row, elements (keeps 16-byte alignment, spreads banks)
constexpr float LOG2E = 1.4426950408889634f;

template <typename T> struct AttnArgs {
    const T* q; const T* k; const T* v;
    T* o; const T* res; T* o_res;
    float* lse; const int* kv_len;
    int B, N, H, ld_qkv, ld_o;
    float scale;
};

// ---- LDS staging of a 64-row x 64-col tile by 256 threads, split into FETCH (global ->
// registers, issued one tile ahead so the loads fly under the MFMAs of the current tile) and
// PUT (registers -> LDS, after the barrier).  Rows >= limit are zero (masked at PUT time, so
// the loads carry no use while in flight).
//
// A tile is needed in two roles: as ROW operand (fragment = 8 consecutive columns of one row:
// ds_read_b128 from a row-major image, stride LDT) and as TRANSPOSED operand (fragment = 8
// consecutive ROWS of one column: V for P.V, K for dQ, Q / dO for dK / dV).
//   bf16: both images are row-major; the transposed fragments come from the hardware
//         transposing read ds_read_b64_tr_b16 (two per fragment) on an image with a 192-byte row
//         stride (conflict free for that read).  Global loads are fully coalesced: 8 lanes read
//         one 128-byte row, a wave-load covers 8 whole rows.
//   fp32 (parity build): there is no 32-bit transposing read; the transposed image is written
//         [col][row] with 8-byte LDS stores from a (row pair, column group) thread mapping.
template <typename T> struct Tile2 { Frag<T> a, b; bool oka, okb; };
constexpr int LDR = 96;      // row stride (elements) of the bf16 image that feeds ds_read_b64_tr_b16

template <typename T> constexpr int tr_elems() { return sizeof(T) == 2 ? KT * LDR : DH * LDT; }

template <typename T> MTMP_DEV void tile_map(int tid, int& ra, int& rb, int& col) {
    if (sizeof(T) == 2) { ra = tid >> 3; rb = ra + 32; col = (tid & 7) * 8; }
    else                { ra = (tid & 31) * 2; rb = ra + 1; col = (tid >> 5) * 8; }
}
template <typename T> MTMP_DEV Tile2<T> tile_fetch(const T* src, int ld, int row0, int limit, int tid) {
    int ra, rb, col;
    tile_map<T>(tid, ra, rb, col);
    Tile2<T> t;
    t.a = frag_load<T>(src + (size_t)min(row0 + ra, limit - 1) * ld + col);
    t.b = frag_load<T>(src + (size_t)min(row0 + rb, limit - 1) * ld + col);
    t.oka = row0 + ra < limit;
    t.okb = row0 + rb < limit;
    return t;
}
template <typename T> MTMP_DEV void put_rows(T* dst, const Tile2<T>& t, int tid) {
    int ra, rb, col;
    tile_map<T>(tid, ra, rb, col);
    if (wave_all(t.oka && t.okb)) {
        frag_store<T>(dst + ra * LDT + col, t.a);
        frag_store<T>(dst + rb * LDT + col, t.b);
    } else {
        frag_store<T>(dst + ra * LDT + col, frag_keep(t.a, t.oka));
        frag_store<T>(dst + rb * LDT + col, frag_keep(t.b, t.okb));
    }
}
// image for the TRANSPOSED role
MTMP_DEV void put_tr(bf16* dst, const Tile2<bf16>& t, int tid) {
    int ra, rb, col;
    tile_map<bf16>(tid, ra, rb, col);
    if (wave_all(t.oka && t.okb)) {
        frag_store<bf16>(dst + ra * LDR + col, t.a);
        frag_store<bf16>(dst + rb * LDR + col, t.b);
    } else {
        frag_store<bf16>(dst + ra * LDR + col, frag_keep(t.a, t.oka));
        frag_store<bf16>(dst + rb * LDR + col, frag_keep(t.b, t.okb));
    }
}
MTMP_DEV void put_tr(float* dst, const Tile2<float>& t, int tid) {
    float* d = dst + (tid >> 5) * 8 * LDT + (tid & 31) * 2;
    const Frag<float> a = frag_keep(t.a, t.oka), b = frag_keep(t.b, t.okb);
#pragma unroll
    for (int e = 0; e < 8; ++e) *reinterpret_cast<f32x2*>(d + e * LDT) = f32x2{a.v[e], b.v[e]};
}
// fragment of the transposed role: element j = tile[row0 + 8*half + j][col0 + r]   (r = lane & 31)
typedef short s16x4 __attribute__((ext_vector_type(4)));
MTMP_DEV Frag<bf16> frag_tr(const bf16* img, int row0, int col0, int lane) {
    // ds_read_b64_tr_b16: within a group of 16 lanes, lane 4q+p supplies the address of row q,
    // columns 4p..4p+3 of a 4 x 16 block; lane i receives column i of the 4 rows.
    const int G = lane >> 4, i = lane & 15;
    const bf16* a = img + (row0 + 8 * (G >> 1) + (i >> 2)) * LDR + col0 + 16 * (G & 1) + 4 * (i & 3);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * LDR));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    Frag<bf16> f;
    f.v = __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    return f;
}
MTMP_DEV Frag<float> frag_tr(const float* img, int row0, int col0, int lane) {
    return frag_load<float>(img + (col0 + (lane & 31)) * LDT + row0 + 8 * (lane >> 5));
}

// 32x32 tile: acc += A(rows through swz23 from an LDS row-major tile) * B(register fragments over dh = 64)
template <typename T>
MTMP_DEV void tile_qk(f32x16& acc, const T* lds_rows, int r, int half, const Frag<T> (&bf)[4]) {
    const T* arow = lds_rows + swz23(r) * LDT + 8 * half;
    acc = mma0<T>(frag_load<T>(arow), bf[0]);
#pragma unroll
    for (int c = 1; c < 4; ++c) mma<T>(acc, frag_load<T>(arow + 16 * c), bf[c]);
}

// =============================== forward ====================================
template <typename T>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 4 : 1)) void attn_fwd_kernel(AttnArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sK = reinterpret_cast<T*>(smem_raw);   // [KT][LDT]   keys x dh
    T* sVt = sK + KT * LDT;                   // V image for the transposed role (frag_tr)
    const int nqt = (p.N + 127) >> 7;
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = w % nqt, bh = w / nqt, hd = bh % p.H, b = bh / p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, half = lane >> 5;
    int kvl = p.kv_len ? min(p.kv_len[b], p.N) : p.N;
    const bool uniform = kvl <= 0;            // all keys masked -> reference gives the uniform average
    if (uniform) kvl = p.N;
    const size_t base = (size_t)b * p.N * p.ld_qkv + hd * DH;
    const T* Qb = p.q + base; const T* Kb = p.k + base; const T* Vb = p.v + base;
    const int qrow = qt * 128 + wave * 32 + r;
    Frag<T> qf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        qf[c] = frag_keep(frag_load<T>(Qb + (size_t)min(qrow, p.N - 1) * p.ld_qkv + 16 * c + 8 * half), qrow < p.N);
    f32x16 o0 = {0}, o1 = {0};
    float m = -INFINITY, l = 0.f;
    const float c2 = p.scale * LOG2E;
    const int ntiles = (kvl + KT - 1) / KT;
    Tile2<T> kreg = tile_fetch<T>(Kb, p.ld_qkv, 0, kvl, tid);
    Tile2<T> vreg = tile_fetch<T>(Vb, p.ld_qkv, 0, kvl, tid);
    STAMP_DECL
    for (int it = 0; it < ntiles; ++it) {
        const int k0 = it * KT;
        STAMP(ts0)
        __syncthreads();
        put_rows<T>(sK, kreg, tid);
        put_tr(sVt, vreg, tid);
        __syncthreads();
        STAMP(ts1)
#ifndef MTMP_ABLATE_FETCH                      // (ablation builds: tools/ablate_attn.sh -- never shipped)
        if (it + 1 < ntiles) {                 // next tile's loads fly under this tile's MFMAs
            kreg = tile_fetch<T>(Kb, p.ld_qkv, k0 + KT, kvl, tid);
            vreg = tile_fetch<T>(Vb, p.ld_qkv, k0 + KT, kvl, tid);
        }
#endif
        f32x16 st[2];
        if (!uniform) {
            tile_qk<T>(st[0], sK, r, half, qf);
            tile_qk<T>(st[1], sK + 32 * LDT, r, half, qf);
        } else {
            st[0] = f32x16{0};
            st[1] = f32x16{0};
        }
        if (k0 + KT > kvl) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if (k0 + 32 * kb + acc_row_swz(t, half) >= kvl) st[kb][t] = -INFINITY;
        }
        STAMP(ts2)
        float mx = st[0][0];
#pragma unroll
        for (int t = 1; t < 16; t += 2) mx = max3(mx, st[0][t], st[0][t + 1 < 16 ? t + 1 : t]);
#pragma unroll
        for (int t = 0; t < 16; t += 2) mx = max3(mx, st[1][t], st[1][t + 1]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c2;
        // Deferred rescale (exact): O, l and m move only when some row's maximum grew; while it has
        // not, p = exp2(s - m) <= 1 still holds.  Wave-uniform branch, rare after the first tiles.
        if (!wave_all(mx <= m)) {
            const float m_new = fmaxf(m, mx);
            const float alpha = fast_exp2(m - m_new);
            l *= alpha; o0 *= alpha; o1 *= alpha;
            m = m_new;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
#ifdef MTMP_ABLATE_EXP
                const float pv = fmaf(st[kb][t], c2, -m);
#else
                const float pv = fast_exp2(fmaf(st[kb][t], c2, -m));
#endif
                l += pv;
                st[kb][t] = pv;
            }
        STAMP(ts3)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const Frag<T> pf = frag_from_acc<T>(st[kb], s);
                // (requesting these eight V^T fragments before the softmax math was tried: +24 VGPRs cost a wave of
                //  occupancy and the forward got 3 % slower -- it is VALU-bound, not LDS-latency-bound like the backward)
                mma<T>(o0, frag_tr(sVt, 32 * kb + 16 * s, 0, lane), pf);
                mma<T>(o1, frag_tr(sVt, 32 * kb + 16 * s, 32, lane), pf);
            }
        STAMP(ts4)
        STAMP_ACC
    }
#ifdef MTMP_STAMP
    if (lane == 0) {
        atomicAdd(&g_stamp[0], sa); atomicAdd(&g_stamp[1], sb); atomicAdd(&g_stamp[2], sc); atomicAdd(&g_stamp[3], sd);
        atomicAdd(&g_stamp[4], (unsigned long long)ntiles);
    }
#endif
    l += __shfl_xor(l, 32, 64);
    // Epilogue.  A lane owns one query and 4-element pieces of its O row, so direct stores would be 8-byte pieces
    // at a row stride -- partial-line writes (PMC: 72 MB written per launch for a 33 MB output).  The wave's
    // 32 x 64 tile goes through a wave-private LDS tile instead and leaves as whole 128-byte head rows.
    __syncthreads();                                  // all waves are done with sK / sVt: reuse them as staging
    T* sO = reinterpret_cast<T*>(smem_raw) + wave * 32 * LDT;
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x16& o = dt ? o1 : o0;
            store4<T>(sO + r * LDT + 32 * dt + 8 * g + 4 * half, o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv,
                      o[4 * g + 3] * inv);
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: in-order LDS, no barrier needed
    constexpr int CH = DH * (int)sizeof(T) / 16, RPP = 64 / CH;       // 16-byte chunks per row, rows per pass
    const int q0w = qt * 128 + wave * 32;
#pragma unroll
    for (int ps = 0; ps < 32 / RPP; ++ps) {
        const int rl = ps * RPP + lane / CH, ch = lane % CH;
        if (q0w + rl < p.N) {
            const size_t off = ((size_t)b * p.N + q0w + rl) * p.ld_o + hd * DH + ch * (16 / (int)sizeof(T));
            const u32x4_t ov = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(sO + rl * LDT) + 16 * ch);
            *reinterpret_cast<u32x4_t*>(p.o + off) = ov;
            if (p.o_res) {
                // residual epilogue (encoder.py:27 "outputs += residual"): the residual adds the
                // value of O as stored (i.e. rounded to T), like the reference's tensor add.
                constexpr int E = 16 / (int)sizeof(T);
                T ob[E], rb[E], sb[E];
                __builtin_memcpy(ob, &ov, 16);
                const u32x4_t rvv = *reinterpret_cast<const u32x4_t*>(p.res + off);
                __builtin_memcpy(rb, &rvv, 16);
#pragma unroll
                for (int i = 0; i < E; ++i) sb[i] = from_f32<T>(to_f32(ob[i]) + to_f32(rb[i]));
                u32x4_t sv;
                __builtin_memcpy(&sv, sb, 16);
                *reinterpret_cast<u32x4_t*>(p.o_res + off) = sv;
            }
        }
    }
    if (qrow < p.N && half == 0) p.lse[((size_t)b * p.H + hd) * p.N + qrow] = m + log2f(l);
}

// =============================== backward ===================================
template <typename T> struct AttnBwdArgs {
    const T* q; const T* k; const T* v; const T* d_o;
    const float* lse; const float* delta; const int* kv_len;
    T* dq; T* dk; T* dv;
    int B, N, H, ld_qkv, ld_do, ld_dqkv;
    float scale;
};

// delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d]; one wave per token row (4 heads x 64).
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* o, const T* d_o, float* delta, int B, int N, int H,
                                                         int ld_o, int ld_do) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B * N) return;
    f32x4 a = load4<T>(o + (size_t)row * ld_o + 4 * lane);
    f32x4 g = load4<T>(d_o + (size_t)row * ld_do + 4 * lane);
    float s = a[0] * g[0] + a[1] * g[1] + a[2] * g[2] + a[3] * g[3];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((lane & 15) == 0) {
        const int b = row / N, q = row - b * N, hd = lane >> 4;
        if (hd < H) delta[((size_t)b * H + hd) * N + q] = s;
    }
}

// dQ: workgroup = 128 query rows, loops over key tiles (S^T and dP^T with the query on the lane).
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sK = reinterpret_cast<T*>(smem_raw);   // [KT][LDT]
    T* sV = sK + KT * LDT;                    // [KT][LDT]
    T* sKt = sV + KT * LDT;                   // K image for the transposed role (frag_tr)
    const int nqt = (p.N + 127) >> 7;
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = w % nqt, bh = w / nqt, hd = bh % p.H, b = bh / p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, half = lane >> 5;
    int kvl = p.kv_len ? min(p.kv_len[b], p.N) : p.N;
    const bool uniform = kvl <= 0;
    if (uniform) kvl = p.N;
    const size_t base = (size_t)b * p.N * p.ld_qkv + hd * DH;
    const T* Qb = p.q + base; const T* Kb = p.k + base; const T* Vb = p.v + base;
    const int qrow = qt * 128 + wave * 32 + r;
    Frag<T> qf[4], dof[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        qf[c] = frag_keep(frag_load<T>(Qb + (size_t)min(qrow, p.N - 1) * p.ld_qkv + 16 * c + 8 * half), qrow < p.N);
        dof[c] = frag_keep(frag_load<T>(p.d_o + ((size_t)b * p.N + min(qrow, p.N - 1)) * p.ld_do + hd * DH + 16 * c + 8 * half),
                           qrow < p.N);
    }
    const size_t sidx = ((size_t)b * p.H + hd) * p.N + qrow;
    const float L2 = (qrow < p.N) ? p.lse[sidx] : INFINITY;
    const float dl = (qrow < p.N) ? p.delta[sidx] : 0.f;
    const float c2 = p.scale * LOG2E;
    f32x16 dq0 = {0}, dq1 = {0};
    const int ntiles = uniform ? 0 : (kvl + KT - 1) / KT;   // uniform: scores are constants -> dQ = 0
    Tile2<T> kreg = tile_fetch<T>(Kb, p.ld_qkv, 0, kvl, tid);
    Tile2<T> vreg = tile_fetch<T>(Vb, p.ld_qkv, 0, kvl, tid);
    for (int it = 0; it < ntiles; ++it) {
        const int k0 = it * KT;
        __syncthreads();
        put_rows<T>(sK, kreg, tid);
        put_tr(sKt, kreg, tid);
        put_rows<T>(sV, vreg, tid);
        __syncthreads();
        if (it + 1 < ntiles) {
            kreg = tile_fetch<T>(Kb, p.ld_qkv, k0 + KT, kvl, tid);
            vreg = tile_fetch<T>(Vb, p.ld_qkv, k0 + KT, kvl, tid);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            // transposed K fragments of the dQ product requested before the score math (see the dK/dV kernel)
            Frag<T> trf[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                trf[s][0] = frag_tr(sKt, 32 * kb + 16 * s, 0, lane);
                trf[s][1] = frag_tr(sKt, 32 * kb + 16 * s, 32, lane);
            }
            f32x16 st = {0}, dp = {0};
            tile_qk<T>(st, sK + 32 * kb * LDT, r, half, qf);
            tile_qk<T>(dp, sV + 32 * kb * LDT, r, half, dof);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const bool valid = k0 + 32 * kb + acc_row_swz(t, half) < kvl;
                const float pv = valid ? fast_exp2(fmaf(st[t], c2, -L2)) : 0.f;
                st[t] = pv * (dp[t] - dl);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const Frag<T> dsf = frag_from_acc<T>(st, s);
                mma<T>(dq0, trf[s][0], dsf);
                mma<T>(dq1, trf[s][1], dsf);
            }
        }
    }
    if (qrow < p.N) {
        const size_t orow = ((size_t)b * p.N + qrow) * p.ld_dqkv + hd * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& o = dt ? dq1 : dq0;
                const int d0 = 32 * dt + 8 * g + 4 * half;
                store4<T>(p.dq + orow + d0, o[4 * g] * p.scale, o[4 * g + 1] * p.scale, o[4 * g + 2] * p.scale,
                          o[4 * g + 3] * p.scale);
            }
    }
}

// dK/dV: workgroup = 128 keys (a wave owns 32 and keeps dK, dV in registers), loops over query tiles
// (S and dP with the key on the lane; P^T and dS^T feed the dV / dK MFMAs from the accumulators).
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AttnBwdArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sQ = reinterpret_cast<T*>(smem_raw);   // [KT][LDT]  q x dh
    T* sdO = sQ + KT * LDT;                   // [KT][LDT]
    T* sQt = sdO + KT * LDT;                  // Q image for the transposed role (frag_tr)
    T* sdOt = sQt + tr_elems<T>();             // dO image for the transposed role
    float* sL = reinterpret_cast<float*>(sdOt + tr_elems<T>());   // [KT] lse (log2 units), +inf past N
    float* sD = sL + KT;                                     // [KT] delta
    const int nkt = (p.N + 127) >> 7;
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    const int kt = w % nkt, bh = w / nkt, hd = bh % p.H, b = bh / p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, half = lane >> 5;
    int kvl = p.kv_len ? min(p.kv_len[b], p.N) : p.N;
    const bool uniform = kvl <= 0;
    if (uniform) kvl = p.N;
    const size_t base = (size_t)b * p.N * p.ld_qkv + hd * DH;
    const T* Qb = p.q + base; const T* Kb = p.k + base; const T* Vb = p.v + base;
    const T* dOb = p.d_o + (size_t)b * p.N * p.ld_do + hd * DH;
    const float* Lb = p.lse + ((size_t)b * p.H + hd) * p.N;
    const float* Db = p.delta + ((size_t)b * p.H + hd) * p.N;
    const int kw0 = kt * 128 + wave * 32;      // first key of this wave
    const int key = kw0 + r;                   // this lane's key (column of S)
    const bool key_ok = key < kvl;
    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    if (kt * 128 < kvl) {                      // workgroup-uniform: keys past kv_len get zero gradients
        Frag<T> kf[4], vf[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            kf[c] = frag_keep(frag_load<T>(Kb + (size_t)min(key, kvl - 1) * p.ld_qkv + 16 * c + 8 * half), key_ok);
            vf[c] = frag_keep(frag_load<T>(Vb + (size_t)min(key, kvl - 1) * p.ld_qkv + 16 * c + 8 * half), key_ok);
        }
        const float c2 = p.scale * LOG2E;
        const int nq = (p.N + KT - 1) / KT;
        Tile2<T> qreg = tile_fetch<T>(Qb, p.ld_qkv, 0, p.N, tid);
        Tile2<T> oreg = tile_fetch<T>(dOb, p.ld_do, 0, p.N, tid);
        float lreg = (tid < KT) ? ((tid < p.N) ? Lb[tid] : INFINITY) : 0.f;
        float dreg = (tid < KT && tid < p.N) ? Db[tid] : 0.f;
        for (int it = 0; it < nq; ++it) {
            const int q0 = it * KT;
            __syncthreads();
            put_rows<T>(sQ, qreg, tid);
            put_tr(sQt, qreg, tid);
            put_rows<T>(sdO, oreg, tid);
            put_tr(sdOt, oreg, tid);
            if (tid < KT) { sL[tid] = lreg; sD[tid] = dreg; }
            __syncthreads();
#ifndef MTMP_DKDV_NOFETCH                      // (ablation builds: tools/ablate_dkdv.sh -- never shipped)
            if (it + 1 < nq) {
                qreg = tile_fetch<T>(Qb, p.ld_qkv, q0 + KT, p.N, tid);
                oreg = tile_fetch<T>(dOb, p.ld_do, q0 + KT, p.N, tid);
                if (tid < KT) {
                    lreg = (q0 + KT + tid < p.N) ? Lb[q0 + KT + tid] : INFINITY;
                    dreg = (q0 + KT + tid < p.N) ? Db[q0 + KT + tid] : 0.f;
                }
            }
#endif
            if (kw0 < kvl) {                   // wave-uniform
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    // Everything this query block reads from LDS is requested up front: the 8 transposed fragments of
                    // the dV / dK products and the lse / delta of its 16 + 16 query rows as four 16-byte reads each.
                    // (Ablations, tools/ablate_dkdv.sh: 64 scalar ds_read_b32 of lse/delta per tile cost 94 us of a
                    //  439 us backward, and transposed reads issued right in front of their MFMA another ~140 us of
                    //  exposed LDS latency; exp2 and the global loads cost nothing.)
                    Frag<T> trf[2][4];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int q16 = 32 * qb + 16 * s;
                        trf[s][0] = frag_tr(sdOt, q16, 0, lane);
                        trf[s][1] = frag_tr(sdOt, q16, 32, lane);
                        trf[s][2] = frag_tr(sQt, q16, 0, lane);
                        trf[s][3] = frag_tr(sQt, q16, 32, lane);
                    }
                    float lq[16], dq_[16];         // row t of the accumulators = query 32qb + 16(t>>3) + 8half + (t&7)
#pragma unroll
                    for (int h8 = 0; h8 < 2; ++h8)
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + 32 * qb + 16 * h8 + 8 * half + 4 * v);
                            const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + 32 * qb + 16 * h8 + 8 * half + 4 * v);
#pragma unroll
                            for (int i = 0; i < 4; ++i) { lq[8 * h8 + 4 * v + i] = l4[i]; dq_[8 * h8 + 4 * v + i] = d4[i]; }
                        }
                    f32x16 st = {0}, dp = {0};
                    if (!uniform) tile_qk<T>(st, sQ + 32 * qb * LDT, r, half, kf);
                    tile_qk<T>(dp, sdO + 32 * qb * LDT, r, half, vf);
                    f32x16 ds;
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        // uniform case: st == 0 and the forward stored lse = log2(N), so pv = 1/N
                        const float pv = key_ok ? fast_exp2(fmaf(st[t], c2, -lq[t])) : 0.f;
                        st[t] = pv;
                        ds[t] = uniform ? 0.f : pv * (dp[t] - dq_[t]);
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const Frag<T> pf = frag_from_acc<T>(st, s);
                        const Frag<T> dsf = frag_from_acc<T>(ds, s);
                        mma<T>(dv0, pf, trf[s][0]);
                        mma<T>(dv1, pf, trf[s][1]);
                        mma<T>(dk0, dsf, trf[s][2]);
                        mma<T>(dk1, dsf, trf[s][3]);
                    }
                }
            }
        }
    }
    // rows = keys (registers), cols = dh (lanes)
    const size_t obase = (size_t)b * p.N * p.ld_dqkv + hd * DH;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int krow = kw0 + acc_row(t, half);
        if (krow < p.N) {
            T* dkp = p.dk + obase + (size_t)krow * p.ld_dqkv;
            T* dvp = p.dv + obase + (size_t)krow * p.ld_dqkv;
            dkp[r] = from_f32<T>(dk0[t] * p.scale);
            dkp[32 + r] = from_f32<T>(dk1[t] * p.scale);
            dvp[r] = from_f32<T>(dv0[t]);
            dvp[32 + r] = from_f32<T>(dv1[t]);
        }
    }
}

template <typename T> size_t fwd_smem() { return (size_t)(KT * LDT + tr_elems<T>()) * sizeof(T); }
template <typename T> size_t dq_smem() { return (size_t)(2 * KT * LDT + tr_elems<T>()) * sizeof(T); }
template <typename T> size_t dkdv_smem() { return (size_t)(2 * KT * LDT + 2 * tr_elems<T>()) * sizeof(T) + 2 * KT * sizeof(float); }

template <typename K> int set_smem(K kern, size_t bytes) {
    if (bytes > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bytes) != hipSuccess) {
            mtmp_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", bytes);
            return MTMP_ERR_LAUNCH;
        }
    }
    return MTMP_OK;
}

template <typename T>
int launch_fwd(const void* q, const void* k, const void* v, void* o, const void* res, void* o_res, float* lse,
               const int* kv_len, int B, int N, int H, int ld_qkv, int ld_o, float scale, hipStream_t st) {
    AttnArgs<T> a{(const T*)q, (const T*)k, (const T*)v, (T*)o, (const T*)res, (T*)o_res, lse, kv_len,
                  B, N, H, ld_qkv, ld_o, scale};
    const int nwg = ((N + 127) / 128) * H * B;
    const size_t sm = fwd_smem<T>();
    if (int e = set_smem(attn_fwd_kernel<T>, sm)) return e;
    hipLaunchKernelGGL(attn_fwd_kernel<T>, dim3(nwg), dim3(256), sm, st, a);
    MTMP_CHECK_LAUNCH("mtmp_attn_fwd");
    return MTMP_OK;
}

template <typename T>
int launch_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
               const int* kv_len, void* dq, void* dk, void* dv, float* delta, int B, int N, int H, int ld_qkv,
               int ld_o, int ld_do, int ld_dqkv, float scale, hipStream_t st) {
    hipLaunchKernelGGL(attn_delta_kernel<T>, dim3((B * N + 3) / 4), dim3(256), 0, st, (const T*)o, (const T*)d_o,
                       delta, B, N, H, ld_o, ld_do);
    MTMP_CHECK_LAUNCH("mtmp_attn_bwd(delta)");
    AttnBwdArgs<T> a{(const T*)q, (const T*)k, (const T*)v, (const T*)d_o, lse, delta, kv_len,
                     (T*)dq, (T*)dk, (T*)dv, B, N, H, ld_qkv, ld_do, ld_dqkv, scale};
    const int nwg = ((N + 127) / 128) * H * B;
    if (int e = set_smem(attn_bwd_dkdv_kernel<T>, dkdv_smem<T>())) return e;
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel<T>, dim3(nwg), dim3(256), dkdv_smem<T>(), st, a);
    MTMP_CHECK_LAUNCH("mtmp_attn_bwd(dkdv)");
    if (int e = set_smem(attn_bwd_dq_kernel<T>, dq_smem<T>())) return e;
    hipLaunchKernelGGL(attn_bwd_dq_kernel<T>, dim3(nwg), dim3(256), dq_smem<T>(), st, a);
    MTMP_CHECK_LAUNCH("mtmp_attn_bwd(dq)");
    return MTMP_OK;
}

bool attn_shape_ok(int B, int N, int H, int ld_a, int ld_b) {
    return B > 0 && N > 0 && H > 0 && H <= 4 && ld_a >= H * DH && ld_b >= H * DH && (ld_a % 8) == 0 && (ld_b % 8) == 0;
}

}  // namespace

extern "C" int mtmp_attn_fwd(int dtype, const void* q, const void* k, const void* v, void* o, const void* res,
                             void* o_res, float* lse, const int32_t* kv_len, int B, int N, int H, int ld_qkv,
                             int ld_o, float scale, void* stream) {
    MTMP_CHECK_ARG(q && k && v && o && lse, "mtmp_attn_fwd: null pointer");
    MTMP_CHECK_ARG((res == nullptr) == (o_res == nullptr), "mtmp_attn_fwd: res and o_res must be given together");
    MTMP_CHECK_ARG(attn_shape_ok(B, N, H, ld_qkv, ld_o), "mtmp_attn_fwd: bad shape B=%d N=%d H=%d ld=%d/%d", B, N, H,
                   ld_qkv, ld_o);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) return launch_fwd<float>(q, k, v, o, res, o_res, lse, kv_len, B, N, H, ld_qkv, ld_o, scale, st);
    if (dtype == 1) return launch_fwd<bf16>(q, k, v, o, res, o_res, lse, kv_len, B, N, H, ld_qkv, ld_o, scale, st);
    mtmp_set_error("mtmp_attn_fwd: unknown dtype %d", dtype);
    return MTMP_ERR_ARG;
}

extern "C" int mtmp_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                             const float* lse, const int32_t* kv_len, void* dq, void* dk, void* dv, float* delta_ws,
                             int B, int N, int H, int ld_qkv, int ld_o, int ld_do, int ld_dqkv, float scale,
                             void* stream) {
    MTMP_CHECK_ARG(q && k && v && o && d_o && lse && dq && dk && dv && delta_ws, "mtmp_attn_bwd: null pointer");
    MTMP_CHECK_ARG(attn_shape_ok(B, N, H, ld_qkv, ld_o) && attn_shape_ok(B, N, H, ld_do, ld_dqkv),
                   "mtmp_attn_bwd: bad shape B=%d N=%d H=%d", B, N, H);
    MTMP_CHECK_ARG(H == 4, "mtmp_attn_bwd: the delta pass assumes 4 heads x 64 (d_model 256), got H=%d", H);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        return launch_bwd<float>(q, k, v, o, d_o, lse, kv_len, dq, dk, dv, delta_ws, B, N, H, ld_qkv, ld_o, ld_do,
                                 ld_dqkv, scale, st);
    if (dtype == 1)
        return launch_bwd<bf16>(q, k, v, o, d_o, lse, kv_len, dq, dk, dv, delta_ws, B, N, H, ld_qkv, ld_o, ld_do,
                                ld_dqkv, scale, st);
    mtmp_set_error("mtmp_attn_bwd: unknown dtype %d", dtype);
    return MTMP_ERR_ARG;
}

#ifdef MTMP_STAMP
// diagnostic build: read and clear the accumulated phase cycles {puts+barriers, S, softmax, PV, tiles}
extern "C" int mtmp_debug_stamps(unsigned long long* out8) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamp), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long z[8] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)) != hipSuccess;
}
#endif
