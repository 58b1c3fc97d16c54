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
// accumulator registers (common.hip.h: acc -> Frag).
#include "common.hip.h"
#include <math.h>
#include <type_traits>

namespace {

constexpr int DH = 64;       // head dim: d_model 256 / 4 heads (tri_mbt_vsltcls.py:29-30)
constexpr int KT = 64;       // rows per LDS tile
constexpr int LDT = DH + 8;  // padded LDS row, elements (keeps 16-byte alignment, spreads banks)
constexpr float LOG2E = 1.4426950408889634f;

template <typename T> struct AttnArgs {
    const T* q; const T* k; const T* v;
    T* o; const T* res; T* o_res;
    float* lse; const int* kv_len;
    const float* knorm;      // [ceil(B N / 32)][H]: max ||k_h||_2 over each 32-row block of the [B N] token rows, or NULL
    int B, N, H, ld_qkv, ld_o;
    float scale;
    // PACKED stream (NULL = the padded [B, N] layout): sample b's tokens are rows row_start[b] .. row_start[b] + kv_len[b] of the
    // q / k / v / o / res buffers -- there are no pad rows, so kv_len[b] is also its query count.  N stays the stride of lse
    // and sizes the grid (workgroups past a sample's rows return at once).  The array is mtmp_row_starts' int32[2 B + 1]:
    // row_start[B + 1 + i] = the sample that the grid's i-th sample slot works on (length-balanced over the XCDs).
    const int* row_start = nullptr;
};

// ---- LDS staging of a 64-row x 64-col tile by 256 threads, split into FETCH (global ->
// registers, issued one tile ahead so the loads fly under the MFMAs of the current tile) and
// PUT (registers -> LDS, after the barrier).  Full tiles are fetched through running pointers; rows
// of a ragged last tile are clamped to the last valid row (never read past it) and made inert by the
// kernels' masks / row constants.
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
constexpr int LDR = 96;      // row stride (elements) of the bf16 image that feeds ds_read_b64_tr_b16

template <typename T> constexpr int tr_elems() { return sizeof(T) == 2 ? KT * LDR : DH * LDT; }

template <typename T> MTMP_DEV void tile_map(int tid, int& ra, int& rb, int& col) {
    if (sizeof(T) == 2) { ra = tid >> 3; rb = ra + 32; col = (tid & 7) * 8; }
    else                { ra = (tid & 31) * 2; rb = ra + 1; col = (tid >> 5) * 8; }
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
// Workgroup = 4 waves = 256 query rows; a wave owns 64 of them (two 32-query blocks, query = lane) and walks the
// keys in 64-key tiles staged through a double-buffered LDS pair (one barrier per tile; the next tile's global loads
// are issued two tiles ahead into registers).  Per tile a wave runs four 32 x 32 "units" (query block qb, key block kb):
//     S^T = K Q^T (4 MFMA)  ->  softmax numerators p, row sum  ->  O^T += V^T P^T (4 MFMA, P^T straight from the
//     accumulator registers)
// K / V fragments of a key block are read from LDS once and serve both query blocks (half the LDS bytes per MFMA of the
// 32-query waves this kernel had through round 2: with the vector work cut down, LDS was the next co-limiter).
//
// At d_h = 64 the loop is bound by the vector port, not by the matrix pipe (round-2 counters: 13 vector instructions per
// MFMA, vector port 61 % busy, matrix pipe 32 %), so there are TWO bodies:
//
//   bounded ("fast") body -- softmax is shift invariant and the running maximum exists only to keep exp() in range.
//     When |s| <= ||q|| ||k|| is known to stay below FAST_BOUND (log2 units) for every query of the wave, exp2(s) can
//     neither overflow nor lose the row's largest terms, so the body is  p = exp2(s), l += p  and nothing else: no row
//     maximum (24 v_max3 + cross-half swap), no s*c2 - m (32 fma), no rescale branch.  The softmax scale and log2(e)
//     are folded into the Q fragments once per wave.  LSE = log2(l) (reference point 0) stays exact.
//     The units of a tile are software-pipelined inside the wave: the score MFMAs of unit u+1 and the P.V MFMAs of unit
//     u-1 are issued around the exponentials of unit u (sched_group_barrier lays out 1 MFMA : 2 exp : 3 VALU).
//     The bound comes from p.knorm: max ||k_h|| per 32-token block, a by-product of the Q/K/V projection's epilogue
//     (mtmp_ln_gemm_qkv) or of mtmp_key_norms; ||q|| is computed here from the fragments the lane holds.
//   online body -- the classic running maximum with a deferred (exact) rescale; taken per WAVE when the bound is
//     missing (knorm == NULL), too large, or not finite.  Both bodies have the same barrier structure, so the waves
//     of one workgroup may differ.
template <typename T> struct TileR { Frag<T> a, b; };

template <typename T> MTMP_DEV void put_rows_r(T* dst, const TileR<T>& t, int tid) {
    int ra, rb, col;
    tile_map<T>(tid, ra, rb, col);
    frag_store<T>(dst + ra * LDT + col, t.a);
    frag_store<T>(dst + rb * LDT + col, t.b);
}
MTMP_DEV void put_tr_r(bf16* dst, const TileR<bf16>& t, int tid) {
    int ra, rb, col;
    tile_map<bf16>(tid, ra, rb, col);
    frag_store<bf16>(dst + ra * LDR + col, t.a);
    frag_store<bf16>(dst + rb * LDR + col, t.b);
}
MTMP_DEV void put_tr_r(float* dst, const TileR<float>& t, int tid) {
    float* d = dst + (tid >> 5) * 8 * LDT + (tid & 31) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) *reinterpret_cast<f32x2*>(d + e * LDT) = f32x2{t.a.v[e], t.b.v[e]};
}
template <typename T> MTMP_DEV Frag<T> frag_scale(const Frag<T>& f, float s) {
    Frag<T> r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[j] = from_f32<T>(to_f32(f.v[j]) * s);
    return r;
}
// |scaled score| (log2 units) up to which the bounded body is taken: exp2(+-64) and sums of a few thousand such terms
// (times |v|) are far inside the f32 / bf16 exponent range (2^+-126), and a row whose scores are all near -64 still
// keeps 60 binades below its largest term.
constexpr float FAST_BOUND = 64.0f;
constexpr int FWD_QW = 64;                      // query rows per wave
constexpr int FWD_QWG = 4 * FWD_QW;             // ... per workgroup

template <typename T> constexpr int fwd_stage_elems() { return KT * LDT + tr_elems<T>(); }     // one K + V^T image pair

template <typename T>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void attn_fwd_kernel(Grouped<AttnArgs<T>> grp) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sbase = reinterpret_cast<T*>(smem_raw);       // 2 x { K rows [KT][LDT] | V image for the transposed role }
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = grp_find(grp, wg);
    const AttnArgs<T>& p = grp.seg[seg];
    const int w = wg - grp.first[seg];
    const int nqt = (p.N + FWD_QWG - 1) / FWD_QWG;
    const int qt = w % nqt, bh = w / nqt, hd = bh % p.H;
    const int b = p.row_start ? p.row_start[p.B + 1 + bh / p.H] : bh / p.H;      // packed: samples in mtmp_row_starts' balanced order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, half = lane >> 5;
    int kvl = p.kv_len ? min(p.kv_len[b], p.N) : p.N;
    const int row0 = p.row_start ? p.row_start[b] : b * p.N;     // first token row of sample b
    const int Nq = p.row_start ? max(kvl, 0) : p.N;              // its query rows
    if (qt * FWD_QWG >= Nq) return;                              // (packed stream; workgroup-uniform, before any barrier)
    // all keys masked -> the reference's masked_fill(-65504) + softmax gives the uniform average over all N
    // keys: run the ordinary loop with Q = 0 (every score 0, every p = 1).
    const bool uniform = kvl <= 0;
    if (uniform) kvl = Nq;
    const size_t base = (size_t)row0 * p.ld_qkv + hd * DH;
    const T* Qb = p.q + base; const T* Kb = p.k + base; const T* Vb = p.v + base;
    const int q0w = qt * FWD_QWG + wave * FWD_QW;          // first query row of this wave
    // Prologue: EVERY load of the workgroup's first phase is issued before anything waits -- this lane's entry of the key-norm
    // table (clamped address, unconditional), the Q fragments, the first key tile -- so the wave pays one memory latency, not
    // three in a row (Q -> ||q|| -> table -> first tile, as the code stood through round 3).
    const int blk0 = row0 >> 5, blk1 = (row0 + Nq - 1) >> 5;
    float km = 0.f;
    if (p.knorm) km = p.knorm[(size_t)min(blk0 + lane, blk1) * p.H + hd];
    Frag<T> qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qrow = q0w + 32 * qb + r;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            qf[qb][c] = frag_keep(frag_load<T>(Qb + (size_t)min(qrow, Nq - 1) * p.ld_qkv + 16 * c + 8 * half),
                                  qrow < Nq && !uniform);
    }
    const float c2 = p.scale * LOG2E;
    f32x16 o[2][2] = {{{0}, {0}}, {{0}, {0}}};                // [query block][dh half]
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    const int nfull = kvl / KT, ntiles = (kvl + KT - 1) / KT;
    int ra, rb, col;
    tile_map<T>(tid, ra, rb, col);
    const size_t tstep = (size_t)KT * p.ld_qkv;
    const T* kpa = Kb + (size_t)ra * p.ld_qkv + col;     // this thread's two rows of the NEXT full tile to fetch
    const T* kpb = Kb + (size_t)rb * p.ld_qkv + col;
    const ptrdiff_t kv_off = Vb - Kb;
    TileR<T> kreg, vreg;
    // global -> registers, two tiles ahead of the MFMAs; rows of the ragged last tile are clamped to kv_len - 1: keys
    // past kv_len are never read, their scores are masked and their P is 0, so whatever finite row stands in for
    // them adds nothing.
    auto fetch = [&](int t) {
        if (t < nfull) {
            kreg.a = frag_load<T>(kpa); kreg.b = frag_load<T>(kpb);
            vreg.a = frag_load<T>(kpa + kv_off); vreg.b = frag_load<T>(kpb + kv_off);
            kpa += tstep; kpb += tstep;
        } else {
            const T* ka = Kb + (size_t)min(t * KT + ra, kvl - 1) * p.ld_qkv + col;
            const T* kb_ = Kb + (size_t)min(t * KT + rb, kvl - 1) * p.ld_qkv + col;
            kreg.a = frag_load<T>(ka); kreg.b = frag_load<T>(kb_);
            vreg.a = frag_load<T>(ka + kv_off); vreg.b = frag_load<T>(kb_ + kv_off);
        }
    };
    auto put = [&](int t) {
        T* sK = sbase + (t & 1) * fwd_stage_elems<T>();
        put_rows_r<T>(sK, kreg, tid);
        put_tr_r(sK + KT * LDT, vreg, tid);
    };
    fetch(0);
    // ---- which body?  (wave-uniform)
    bool fast = false;
    if (p.knorm) {
        float qs = 0.f;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) s = fmaf(to_f32(qf[qb][c].v[j]), to_f32(qf[qb][c].v[j]), s);
            qs = fmaxf(qs, s);
        }
        qs = half_sum(qs) ;                                   // >= either row's ||q||^2 (the half-lanes hold half rows)
        for (int i = blk0 + lane + 64; i <= blk1; i += 64) km = fmaxf(km, p.knorm[(size_t)i * p.H + hd]);   // (samples of more than 2048 rows)
        km = wave_max(km);
        // 1.02: the key norms may have been taken before the keys were rounded to bf16, q * c2 is rounded again below
        const float bound = sqrtf(qs) * km * (c2 * 1.02f);
        fast = wave_all(bound <= FAST_BOUND);                 // (a NaN anywhere fails the comparison: online body)
    }
    put(0);
    if (ntiles > 1) fetch(1);
    // start of tile `it`: its images become visible, the other buffer is free for tile it + 1
    auto tile_begin = [&](int it) {
        __syncthreads();
        if (it + 1 < ntiles) put(it + 1);
        if (it + 2 < ntiles) fetch(it + 2);
    };
    auto load_k = [&](const T* sK, int kb, Frag<T> (&ka)[4]) {
        const T* arow = sK + (32 * kb + swz23(r)) * LDT + 8 * half;
#pragma unroll
        for (int c = 0; c < 4; ++c) ka[c] = frag_load<T>(arow + 16 * c);
    };
    auto load_v = [&](const T* sV, int kb, Frag<T> (&vt)[2][2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            vt[s][0] = frag_tr(sV, 32 * kb + 16 * s, 0, lane);
            vt[s][1] = frag_tr(sV, 32 * kb + 16 * s, 32, lane);
        }
    };
    auto scores = [&](f32x16& st, const Frag<T> (&ka)[4], int qb) {
        st = mma0<T>(ka[0], qf[qb][0]);
#pragma unroll
        for (int c = 1; c < 4; ++c) mma<T>(st, ka[c], qf[qb][c]);
    };
    // Ragged last tile: keys >= kv_len drop out THROUGH THE MATRIX PIPE.  Keys are accumulator rows of S^T = K Q^T, so the
    // mask is a per-register constant: the tail tile's score products start from C = -inf in the rows of the missing keys
    // (0 elsewhere) instead of from the inline 0 -- exp2(-inf + finite) = 0 with no compare / select per score (the select
    // form cost 135 vector instructions more than a full tile for the same 32 MFMAs).  Built once per key block of the tail.
    auto tail_c = [&](int k0) {
        f32x16 c;
#pragma unroll
        for (int t = 0; t < 16; ++t) c[t] = (k0 + acc_row_swz(t, half) >= kvl) ? -INFINITY : 0.f;
        return c;
    };
    auto scores_c = [&](f32x16& st, const Frag<T> (&ka)[4], int qb, const f32x16& c) {
        st = mma_c<T>(ka[0], qf[qb][0], c);
#pragma unroll
        for (int c4 = 1; c4 < 4; ++c4) mma<T>(st, ka[c4], qf[qb][c4]);
    };
    auto pv = [&](int qb, const Frag<T> (&pf)[2], const Frag<T> (&vt)[2][2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            mma<T>(o[qb][0], vt[s][0], pf[s]);
            mma<T>(o[qb][1], vt[s][1], pf[s]);
        }
    };
    if (fast) {
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int c = 0; c < 4; ++c) qf[qb][c] = frag_scale<T>(qf[qb][c], c2);
        auto soft = [&](f32x16& st, int qb, Frag<T> (&pf)[2]) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float e = fast_exp2(st[t]);
                l[qb] += e;
                st[t] = e;
            }
            pf[0] = frag_from_acc<T>(st, 0);
            pf[1] = frag_from_acc<T>(st, 1);
        };
        // a slot's eight MFMAs: four exponentials behind each of the first four, six plain vector instructions (4 adds + 2
        // conversions) behind each of the last four -- nothing reads an exponential that was issued a moment ago
        // (75.8 us against 78.3 us for 1 MFMA : 2 exp : 3 VALU eight times, tools/attn_lab.py)
        auto interleave = [&]() {
            if (sizeof(T) == 2) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                }
            }
        };
        // The unit pipeline runs ACROSS the tile boundary (round 4).  Through round 3 it drained at every barrier -- a tile's
        // first slot was four score MFMAs with nothing beside them, its last four P.V MFMAs likewise (6 slots for 4 units of
        // vector work) -- and the K fragments of the next tile were waited for with an empty matrix pipe.  Here the last unit's
        // softmax and the last two units' P.V products of tile t run beside the first score products of tile t + 1, after
        // that tile's barrier (they read registers only: P, and the V fragments fetched one slot before the barrier): every
        // slot is 8 MFMAs beside one unit of softmax, 4 slots per tile.  The state crossing a barrier is s11 (raw scores of
        // unit (1,1)), p01 and the key block's V fragments; it starts as "-inf scores, zero P" so that the first tile needs no
        // special body (its eight idle P.V MFMAs add zeros: 1.5 % of a 16-tile sample).
        auto body_rot = [&](int it, auto tail_tag, f32x16& s11, Frag<T> (&p01)[2], Frag<T> (&vtp)[2][2]) {
            constexpr bool TAIL = decltype(tail_tag)::value;
            __syncthreads();                                 // tile `it` visible; the other buffer is free for tile it + 1
            const T* sK = sbase + (it & 1) * fwd_stage_elems<T>();
            const T* sV = sK + KT * LDT;
            const int k0 = it * KT;
            f32x16 s00, s10, s01;
            Frag<T> ka0[4], ka1[4], vt0[2][2], p00[2], p10[2], p11[2];
            load_k(sK, 0, ka0);                              // (ahead of the staging stores in this wave's LDS queue)
            if (it + 1 < ntiles) put(it + 1);
            if (it + 2 < ntiles) fetch(it + 2);
            f32x16 c0, c1;
            if (TAIL) { c0 = tail_c(k0); c1 = tail_c(k0 + 32); }
            pv(0, p01, vtp);                                 // slot A: PV(prev 0,1) + S(0,0) || softmax(prev 1,1)
            if (TAIL) scores_c(s00, ka0, 0, c0); else scores(s00, ka0, 0);
            soft(s11, 1, p11);
            load_v(sV, 0, vt0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            pv(1, p11, vtp);                                 // slot B: PV(prev 1,1) + S(1,0) || softmax(0,0)
            if (TAIL) scores_c(s10, ka0, 1, c0); else scores(s10, ka0, 1);
            soft(s00, 0, p00);
            load_k(sK, 1, ka1);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            pv(0, p00, vt0);                                 // slot C: PV(0,0) + S(0,1) || softmax(1,0)
            if (TAIL) scores_c(s01, ka1, 0, c1); else scores(s01, ka1, 0);
            soft(s10, 1, p10);
            load_v(sV, 1, vtp);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            pv(1, p10, vt0);                                 // slot D: PV(1,0) + S(1,1) || softmax(0,1)
            if (TAIL) scores_c(s11, ka1, 1, c1); else scores(s11, ka1, 1);
            soft(s01, 0, p01);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
        };
        {
            f32x16 s11;
            Frag<T> p01[2] = {frag_zero<T>(), frag_zero<T>()}, vtp[2][2] = {{frag_zero<T>(), frag_zero<T>()}, {frag_zero<T>(), frag_zero<T>()}};
#pragma unroll
            for (int t = 0; t < 16; ++t) s11[t] = -INFINITY;
            for (int it = 0; it < nfull; ++it) body_rot(it, std::false_type{}, s11, p01, vtp);
            if (ntiles > nfull) body_rot(nfull, std::true_type{}, s11, p01, vtp);
            Frag<T> p11[2];                                  // drain: PV(last 0,1) || softmax(last 1,1), then PV(last 1,1)
            pv(0, p01, vtp);
            soft(s11, 1, p11);
            __builtin_amdgcn_sched_barrier(0);
            pv(1, p11, vtp);
        }
        m[0] = m[1] = 0.f;
    } else {
        auto body = [&](int it, auto tail_tag) {
            constexpr bool TAIL = decltype(tail_tag)::value;
            tile_begin(it);
            const T* sK = sbase + (it & 1) * fwd_stage_elems<T>();
            const T* sV = sK + KT * LDT;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                Frag<T> ka[4], vt[2][2];
                load_k(sK, kb, ka);
                load_v(sV, kb, vt);
                f32x16 ct;
                if (TAIL) ct = tail_c(it * KT + 32 * kb);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    f32x16 st;
                    if (TAIL) scores_c(st, ka, qb, ct); else scores(st, ka, qb);
                    // v_max3 is inline asm, which the compiler's hazard recogniser does not see: an asm instruction
                    // that reads an MFMA result too early gets whatever the register holds.  `seed` is an ordinary
                    // instruction on the LAST accumulator register written, so the required wait states are inserted
                    // in front of it, and every asm below depends on it.
                    const float seed = fmaxf(st[15], st[14]);
                    float mxa = max3(seed, st[0], st[1]), mxb = max3(seed, st[2], st[3]);
#pragma unroll
                    for (int t = 4; t < 12; t += 4) {
                        mxa = max3(mxa, st[t], st[t + 1]);
                        mxb = max3(mxb, st[t + 2], st[t + 3]);
                    }
                    const float mx = half_max(max3(mxa, mxb, fmaxf(st[12], st[13]))) * c2;
                    // Deferred rescale (exact): O, l and m move only when some row's maximum grew; while it has
                    // not, p = exp2(s - m) <= 1 still holds.  Wave-uniform branch, rare after the first tiles.
                    if (!wave_all(mx <= m[qb])) {
                        const float m_new = fmaxf(m[qb], mx);
                        const float alpha = fast_exp2(m[qb] - m_new);
                        l[qb] *= alpha; o[qb][0] *= alpha; o[qb][1] *= alpha;
                        m[qb] = m_new;
                    }
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const float e = fast_exp2(fmaf(st[t], c2, -m[qb]));
                        l[qb] += e;
                        st[t] = e;
                    }
                    Frag<T> pf[2] = {frag_from_acc<T>(st, 0), frag_from_acc<T>(st, 1)};
                    pv(qb, pf, vt);
                }
            }
        };
        for (int it = 0; it < nfull; ++it) body(it, std::false_type{});
        if (ntiles > nfull) body(nfull, std::true_type{});
    }
    // Epilogue.  A lane owns one query and 4-element pieces of its O row, so direct stores would be 8-byte pieces
    // at a row stride -- partial-line writes (PMC: 72 MB written per launch for a 33 MB output).  The wave's
    // 64 x 64 tile goes through a wave-private LDS tile instead and leaves as whole 128-byte head rows.
    // Memory order (round 4): the residual pieces are REQUESTED first -- all passes at once, clamped rows, no branch around
    // a load -- then O is normalised and staged while they fly, then all stores go out back to back.  Through round 3 every
    // pass was "store O, load the residual, wait for it (vmcnt(0): for the store just issued as well), add, store": eight
    // (fp32: sixteen) memory round trips in a row per wave.
    constexpr int E = 16 / (int)sizeof(T);                            // elements per 16-byte piece
    constexpr int CH = DH * (int)sizeof(T) / 16, RPP = 64 / CH, NPS = FWD_QW / RPP;   // pieces per row, rows per pass, passes
    const int rsub = lane / CH, ch = lane % CH;
    u32x4_t rv[NPS];
    if (p.o_res) {
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int rowc = min(q0w + ps * RPP + rsub, Nq - 1);
            rv[ps] = *reinterpret_cast<const u32x4_t*>(p.res + ((size_t)row0 + rowc) * p.ld_o + hd * DH + ch * E);
        }
    }
    __syncthreads();                                  // all waves are done with the K / V images: reuse them as staging
    T* sO = reinterpret_cast<T*>(smem_raw) + wave * FWD_QW * LDT;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        l[qb] = half_sum(l[qb]);
        const float inv = 1.0f / l[qb];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& ov = o[qb][dt];
                store4<T>(sO + (32 * qb + r) * LDT + 32 * dt + 8 * g + 4 * half, ov[4 * g] * inv, ov[4 * g + 1] * inv,
                          ov[4 * g + 2] * inv, ov[4 * g + 3] * inv);
            }
        const int qrow = q0w + 32 * qb + r;
        if (qrow < Nq && half == 0) p.lse[((size_t)b * p.H + hd) * p.N + qrow] = m[qb] + log2f(l[qb]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: in-order LDS, no barrier needed
    u32x4_t ovv[NPS];
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps)
        ovv[ps] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(sO + (ps * RPP + rsub) * LDT) + 16 * ch);
    T* orow = p.o + ((size_t)row0 + q0w + rsub) * p.ld_o + hd * DH + ch * E;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps)
        if (q0w + ps * RPP + rsub < Nq) *reinterpret_cast<u32x4_t*>(orow + (size_t)ps * RPP * p.ld_o) = ovv[ps];
    if (p.o_res) {
        // residual epilogue (encoder.py:27 "outputs += residual"): the residual adds the
        // value of O as stored (i.e. rounded to T), like the reference's tensor add.
        T* rrow = p.o_res + ((size_t)row0 + q0w + rsub) * p.ld_o + hd * DH + ch * E;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            T ob[E], rbv[E], sb[E];
            __builtin_memcpy(ob, &ovv[ps], 16);
            __builtin_memcpy(rbv, &rv[ps], 16);
#pragma unroll
            for (int i = 0; i < E; ++i) sb[i] = from_f32<T>(to_f32(ob[i]) + to_f32(rbv[i]));
            u32x4_t sv;
            __builtin_memcpy(&sv, sb, 16);
            if (q0w + ps * RPP + rsub < Nq) *reinterpret_cast<u32x4_t*>(rrow + (size_t)ps * RPP * p.ld_o) = sv;
        }
    }
}

// Max ||k_h||_2 per 32-token block and head of a [M, ld] key matrix (M = B N rows, head h = columns [64h, 64h + 64)):
// the stand-alone producer of AttnArgs::knorm (the Q/K/V projection writes the same table from its epilogue).
// One wave per block; 8 lanes read one 64-element head row.
template <typename T> __global__ __launch_bounds__(256) void key_norms_kernel(const T* k, float* out, int M, int H, int ld) {
    const int lane = threadIdx.x & 63, blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk * 32 >= M) return;
    const int rsub = lane >> 3, ch = lane & 7;
    for (int h = 0; h < H; ++h) {
        float best = 0.f;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = min(blk * 32 + ps * 8 + rsub, M - 1);
            const Frag<T> f = frag_load<T>(k + (size_t)row * ld + h * DH + 8 * ch);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s = fmaf(to_f32(f.v[j]), to_f32(f.v[j]), s);
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            best = fmaxf(best, s);
        }
        best = fmaxf(best, __shfl_xor(best, 8, 64));
        best = fmaxf(best, __shfl_xor(best, 16, 64));
        best = fmaxf(best, __shfl_xor(best, 32, 64));
        if (lane == 0) out[(size_t)blk * H + h] = sqrtf(best);
    }
}

// ---- helpers of the backward kernels ----

// 32x32 tile: C + A(rows through swz23 from an LDS row-major tile) * B(register fragments over dh = 64)
// (the row constants -LSE / -delta are the C operand: D != C, no copy of the shared C registers)

// A staged 64-row x 64-column tile that is read in BOTH roles: row fragments (8 consecutive columns of one row: the A
// operand of the score products) and transposed fragments (8 consecutive ROWS of one column: dQ's K, dK / dV's Q and dO).
//   bf16: ONE unpadded 8 KiB image.  Byte offset of 16-byte chunk ch of row `row` (cdna_hip_programming.md T10, image (a),
//         for 128-byte rows): 1024 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3)) -- 8-row x
//         32-column subtiles of 512 B whose 16-byte chunks are XOR-ed with bits 2-3 of the row.  ds_read_b128 row reads (rows
//         through swz23 or not) and ds_read_b64_tr_b16 reads are both conflict-free on it (bank simulation,
//         tools/dbg/banksim.py; the two padded images of round 1-2 were each conflict-free for ONE role only), so a tile is
//         written once instead of twice and takes 8 KiB instead of 21.5.  A thread stages chunks c and 4 + c of ONE row
//         (row = tid >> 2, c = tid & 3): the eight lanes of a ds_write_b128 service group then cover 2 rows x 4 chunks of one
//         subtile -- no write conflicts (8 chunks of one row would put chunks c and c + 4 on the same banks).
//   fp32 (parity build): no 32-bit transposing read exists: a row image [64][LDT] plus a transposed image written
//         [col][row] with 8-byte stores, as before.
template <typename T> struct DualTile;
template <> struct DualTile<bf16> {
    static constexpr int BYTES = 8192;
    static MTMP_DEV void map(int tid, int& ra, int& rb, int& ca, int& cb) { ra = rb = tid >> 2; ca = 8 * (tid & 3); cb = 32 + 8 * (tid & 3); }
    static MTMP_DEV int off(int row, int ch) {
        return 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
    }
    static MTMP_DEV void put(char* img, const TileR<bf16>& t, int tid) {
        const int row = tid >> 2, c = tid & 3;
        *reinterpret_cast<bf16x8*>(img + off(row, c)) = t.a.v;
        *reinterpret_cast<bf16x8*>(img + off(row, 4 + c)) = t.b.v;
    }
    // k-step c of the row operand: row `row`, elements 16 c + 8 half .. + 7
    static MTMP_DEV Frag<bf16> row_frag(const char* img, int row, int c, int half) {
        Frag<bf16> f;
        f.v = *reinterpret_cast<const bf16x8*>(img + off(row, 2 * c + half));
        return f;
    }
    // transposed operand: element j = tile[row0 + 8 * (lane >> 5) + j][col0 + (lane & 31)]
    static MTMP_DEV Frag<bf16> tr_frag(const char* img, int row0, int col0, int lane) {
        const int G = lane >> 4, i = lane & 15;
        const int row = row0 + 8 * (G >> 1) + (i >> 2), col = col0 + 16 * (G & 1) + 4 * (i & 3);
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off(row, col >> 3) + 2 * (col & 7)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + off(row + 4, col >> 3) + 2 * (col & 7)));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        Frag<bf16> f;
        f.v = __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
        return f;
    }
};
template <> struct DualTile<float> {
    static constexpr int BYTES = (KT * LDT + DH * LDT) * 4;
    static MTMP_DEV void map(int tid, int& ra, int& rb, int& ca, int& cb) { ra = (tid & 31) * 2; rb = ra + 1; ca = cb = (tid >> 5) * 8; }
    static MTMP_DEV void put(char* img, const TileR<float>& t, int tid) {
        float* rows = reinterpret_cast<float*>(img);
        put_rows_r<float>(rows, t, tid);
        put_tr_r(rows + KT * LDT, t, tid);
    }
    static MTMP_DEV Frag<float> row_frag(const char* img, int row, int c, int half) {
        return frag_load<float>(reinterpret_cast<const float*>(img) + row * LDT + 16 * c + 8 * half);
    }
    static MTMP_DEV Frag<float> tr_frag(const char* img, int row0, int col0, int lane) {
        return frag_tr(reinterpret_cast<const float*>(img) + KT * LDT, row0, col0, lane);
    }
};

// One thread's share (two 8-element pieces) of a stream of 64-row x 64-column tiles over rows [0, limit) of a [limit][ld]
// matrix, in DualTile<T>'s thread mapping.  Full tiles are fetched through two running pointers (no address arithmetic in the
// loop); the ragged last tile clamps its rows to limit - 1 (a finite stand-in row; the kernels make such rows inert through
// their row constants or masks, never by reading past `limit`).
template <typename T> struct TileStream {
    const T* base; const T* pa; const T* pb;
    size_t step; int ld, limit, ra, rb, ca, cb, nfull;
    MTMP_DEV void init(const T* src, int ld_, int limit_, int tid) {
        DualTile<T>::map(tid, ra, rb, ca, cb);
        base = src; ld = ld_; limit = limit_; nfull = limit_ / KT; step = (size_t)KT * ld_;
        pa = src + (size_t)ra * ld_ + ca; pb = src + (size_t)rb * ld_ + cb;
    }
    MTMP_DEV TileR<T> fetch(int t) {                 // t must run 0, 1, 2, ... (running pointers)
        TileR<T> x;
        if (t < nfull) {
            x.a = frag_load<T>(pa); x.b = frag_load<T>(pb);
            pa += step; pb += step;
        } else {
            x.a = frag_load<T>(base + (size_t)min(t * KT + ra, limit - 1) * ld + ca);
            x.b = frag_load<T>(base + (size_t)min(t * KT + rb, limit - 1) * ld + cb);
        }
        return x;
    }
    // rows >= limit of tile t -> 0 (ragged key tiles of the dQ kernel)
    MTMP_DEV TileR<T> zero_tail(const TileR<T>& x, int t) const {
        TileR<T> y;
        y.a = frag_keep(x.a, t * KT + ra < limit);
        y.b = frag_keep(x.b, t * KT + rb < limit);
        return y;
    }
};
// =============================== backward ===================================
template <typename T> struct AttnBwdArgs {
    const T* q; const T* k; const T* v; const T* d_o;
    const float* lse; float* delta; const int* kv_len;     // delta: written by the dQ kernel, read by the dK/dV kernel
    T* dq; T* dk; T* dv;
    int B, N, H, ld_qkv, ld_do, ld_dqkv;
    float scale;
    const T* o; int ld_o;
    const int* row_start = nullptr;    // packed stream, as in AttnArgs (lse / delta keep the [B, H, N] layout)
};

// Both backward kernels fold the row constants into the matrix pipe (cdna_hip_programming.md, 'Attention
// backward': row constants as the initial accumulator): with the softmax scale folded into one operand
// (K in dK/dV, Q in dQ) and  -LSE  /  -delta  loaded as the C operand of the two score products,
//     S' = (c2 K) Q^T - LSE     -> p  = exp2(S')          (no fma, no subtraction)
//     dP' = V dO^T - delta      -> dS = p * dP'           (one multiply)
// which leaves exp2 + mul + the bf16 conversions as the only per-score vector work.
// Round 3: K / V (dQ) and Q / dO (dK/dV) tiles live in double-buffered DualTile images -- ONE barrier per tile, the next
// tile's global loads two tiles ahead in registers, each tile written to LDS once -- and the two 32-row units of a tile are
// software-pipelined inside the wave in both kernels (round-2 counters: vector and matrix pipe busy together only 9 % / 15 %
// of the time, 40 % of the wave-cycles in issue stalls).

// dQ: workgroup = 128 query rows, loops over key tiles (S^T and dP^T with the query on the lane).
template <typename T> constexpr int dq_stage_bytes() { return 2 * DualTile<T>::BYTES; }          // K | V
template <typename T>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void attn_bwd_dq_kernel(Grouped<AttnBwdArgs<T>> grp) {
    using DT = DualTile<T>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = grp_find(grp, wg);
    const AttnBwdArgs<T>& p = grp.seg[seg];
    const int w = wg - grp.first[seg];
    const int nqt = (p.N + 127) >> 7;
    const int qt = w % nqt, bh = w / nqt, hd = bh % p.H;
    const int b = p.row_start ? p.row_start[p.B + 1 + bh / p.H] : bh / p.H;      // packed: samples in mtmp_row_starts' balanced order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, half = lane >> 5;
    int kvl = p.kv_len ? min(p.kv_len[b], p.N) : p.N;
    const int row0 = p.row_start ? p.row_start[b] : b * p.N;     // (packed stream: AttnArgs::row_start)
    const int Nq = p.row_start ? max(kvl, 0) : p.N;
    if (qt * 128 >= Nq) return;
    const bool uniform = kvl <= 0;            // scores are constants -> dQ = 0
    if (uniform) kvl = Nq;
    const size_t base = (size_t)row0 * p.ld_qkv + hd * DH;
    const T* Qb = p.q + base; const T* Kb = p.k + base; const T* Vb = p.v + base;
    const int qrow = qt * 128 + wave * 32 + r;
    const float c2 = p.scale * LOG2E;
    // Prologue: every load of the first phase is issued before anything waits (round 4; it used to be three memory latencies in
    // a row: Q / dO / O, then -- behind the delta store -- LSE, then the first key tile): LSE (clamped row, unconditional), the
    // three row fragments, the first K / V tile.
    const size_t sidx = ((size_t)b * p.H + hd) * p.N + min(qrow, Nq - 1);
    const float lse_v = p.lse[sidx];
    Frag<T> qf[4], dof[4], of[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        qf[c] = frag_load<T>(Qb + (size_t)min(qrow, Nq - 1) * p.ld_qkv + 16 * c + 8 * half);
        dof[c] = frag_load<T>(p.d_o + ((size_t)row0 + min(qrow, Nq - 1)) * p.ld_do + hd * DH + 16 * c + 8 * half);
        of[c] = frag_load<T>(p.o + ((size_t)row0 + min(qrow, Nq - 1)) * p.ld_o + hd * DH + 16 * c + 8 * half);
    }
    const int ntiles = uniform ? 0 : (kvl + KT - 1) / KT;
    const int nfull = kvl / KT;
    TileStream<T> ks, vs;
    ks.init(Kb, p.ld_qkv, kvl, tid);
    vs.init(Vb, p.ld_qkv, kvl, tid);
    TileR<T> kreg, vreg;
    auto fetch = [&](int t) { kreg = ks.fetch(t); vreg = vs.fetch(t); };
    auto put = [&](int t) {                    // dQ += dS K: keys past kv_len are staged as zero rows (they add exactly 0)
        char* st_ = smem_raw + (t & 1) * dq_stage_bytes<T>();
        DT::put(st_, t >= nfull ? ks.zero_tail(kreg, t) : kreg, tid);
        DT::put(st_ + DT::BYTES, vreg, tid);
    };
    if (ntiles > 0) fetch(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        qf[c] = frag_keep(frag_scale<T>(qf[c], c2), qrow < Nq);
        dof[c] = frag_keep(dof[c], qrow < Nq);
    }
    // delta[q] = sum_d dO[q,d] O[q,d] (the softmax backward's row constant) is computed here, from the dO fragments
    // this lane already holds and the matching O fragments, and handed to the dK/dV kernel through p.delta -- the
    // separate delta pass (one launch, 66 MB of traffic per call) is gone; the dQ kernel therefore runs FIRST.
    float dsum = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int jx = 0; jx < 8; ++jx) dsum = fmaf(to_f32(dof[c].v[jx]), to_f32(of[c].v[jx]), dsum);
    dsum = half_sum(dsum);                                 // the two half-lanes hold the two halves of the row
    if (qrow < Nq && half == 0) p.delta[sidx] = dsum;
    // rows past N: -LSE = -inf makes p = 0 whatever the (zero) fragments give
    const float nL = (qrow < Nq) ? -lse_v : -INFINITY;
    const float nD = (qrow < Nq) ? -dsum : 0.f;
    f32x16 cL, cD;                            // the row constants in every register: C operands
#pragma unroll
    for (int t = 0; t < 16; ++t) { cL[t] = nL; cD[t] = nD; }
    f32x16 dq0 = {0}, dq1 = {0};
    if (ntiles > 0) put(0);
    if (ntiles > 1) fetch(1);
    struct Unit { f32x16 st, dp; };
    // One tile = two key blocks (units), four phases.  Every phase's LDS fragments are requested one phase AHEAD of the MFMAs that
    // read them (round 4): as the code stood, each phase began with its own ds_reads and its MFMAs waited for them one by one -- the
    // matrix pipe idled through an LDS latency per MFMA in the score phases and through the transposed reads in front of each
    // gradient phase (the disassembly showed `ds_read, s_waitcnt lgkmcnt(0), v_mfma` eight times in a row).
    //   after the barrier:  row fragments of key block 0            | staging stores of tile it + 1, loads of tile it + 2
    //   phase 0:  S, dP (block 0)                                   | row fragments of block 1
    //   phase 1:  S, dP (block 1)  ||  exp / mul / cvt (block 0)    | transposed K fragments of block 0
    //   phase 2:  dQ += (block 0)  ||  exp / mul / cvt (block 1)    | transposed K fragments of block 1
    //   phase 3:  dQ += (block 1)
    auto body = [&](int it, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const int k0 = it * KT;
        __syncthreads();                       // tile `it` visible; the other stage is free for tile it + 1
        const char* sK = smem_raw + (it & 1) * dq_stage_bytes<T>();
        const char* sV = sK + DT::BYTES;
        auto load_rows = [&](int kb, Frag<T> (&ka)[4], Frag<T> (&va)[4]) {
            const int row = 32 * kb + swz23(r);
#pragma unroll
            for (int c = 0; c < 4; ++c) { ka[c] = DT::row_frag(sK, row, c, half); va[c] = DT::row_frag(sV, row, c, half); }
        };
        auto load_tr = [&](int kb, Frag<T> (&trf)[2][2]) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                trf[s][0] = DT::tr_frag(sK, 32 * kb + 16 * s, 0, lane);
                trf[s][1] = DT::tr_frag(sK, 32 * kb + 16 * s, 32, lane);
            }
        };
        auto scores = [&](int kb, Unit& u, const Frag<T> (&ka)[4], const Frag<T> (&va)[4]) {
            if (TAIL) {
                // ragged last tile: -LSE becomes -inf in the accumulator rows of the keys >= kv_len (keys are rows of S^T), so
                // their p -- and with it dS = p * dP' -- is exactly 0 with no select per score
                f32x16 cLt;
#pragma unroll
                for (int t = 0; t < 16; ++t) cLt[t] = (k0 + 32 * kb + acc_row_swz(t, half) < kvl) ? nL : -INFINITY;
                u.st = mma_c<T>(ka[0], qf[0], cLt);
            } else
            u.st = mma_c<T>(ka[0], qf[0], cL);
            // (the score chain first, the dP chain behind it: the exponentials that follow read `st`)
#pragma unroll
            for (int c = 1; c < 4; ++c) mma<T>(u.st, ka[c], qf[c]);
            u.dp = mma_c<T>(va[0], dof[0], cD);
#pragma unroll
            for (int c = 1; c < 4; ++c) mma<T>(u.dp, va[c], dof[c]);
        };
        auto soft = [&](Unit& u, Frag<T> (&dsf)[2]) {
#pragma unroll
            for (int t = 0; t < 16; ++t) u.st[t] = fast_exp2(u.st[t]) * u.dp[t];
            dsf[0] = frag_from_acc<T>(u.st, 0);
            dsf[1] = frag_from_acc<T>(u.st, 1);
        };
        auto grad = [&](const Frag<T> (&dsf)[2], const Frag<T> (&trf)[2][2]) {       // dQ^T += K^T dS^T (transposed K fragments of the same image)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                mma<T>(dq0, trf[s][0], dsf[s]);
                mma<T>(dq1, trf[s][1], dsf[s]);
            }
        };
        Unit u0, u1;
        Frag<T> d0[2], d1[2], ka0[4], va0[4], ka1[4], va1[4], tr0[2][2], tr1[2][2];
        load_rows(0, ka0, va0);                // (ahead of the staging stores in this wave's LDS queue)
        if (it + 1 < ntiles) put(it + 1);
        if (it + 2 < ntiles) fetch(it + 2);
        __builtin_amdgcn_sched_barrier(0);
        load_rows(1, ka1, va1);                // phase 0
        scores(0, u0, ka0, va0);
        if (sizeof(T) == 2) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_tr(0, tr0);                       // phase 1
        scores(1, u1, ka1, va1);
        soft(u0, d0);
        if (sizeof(T) == 2) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_tr(1, tr1);                       // phase 2
        grad(d0, tr0);
        soft(u1, d1);
        if (sizeof(T) == 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        grad(d1, tr1);                         // phase 3
    };
    for (int it = 0; it < min(nfull, ntiles); ++it) body(it, std::false_type{});
    if (ntiles > nfull) body(nfull, std::true_type{});
    if (qrow < Nq) {
        const size_t orow = ((size_t)row0 + qrow) * p.ld_dqkv + hd * DH;
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
// No masks in the loop: a lane whose key is past kv_len computes garbage that stays in its own output
// rows (the contraction runs over queries), and those rows are written as zeros; query rows past N carry
// -LSE = -inf, so their p and dS are exactly 0.
template <typename T> constexpr int dkdv_stage_bytes() { return 2 * DualTile<T>::BYTES + 2 * KT * (int)sizeof(float); }   // Q | dO | -lse | -delta
template <typename T>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void attn_bwd_dkdv_kernel(Grouped<AttnBwdArgs<T>> grp) {
    using DT = DualTile<T>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = grp_find(grp, wg);
    const AttnBwdArgs<T>& p = grp.seg[seg];
    const int w = wg - grp.first[seg];
    const int nkt = (p.N + 127) >> 7;
    const int kt = w % nkt, bh = w / nkt, hd = bh % p.H;
    const int b = p.row_start ? p.row_start[p.B + 1 + bh / p.H] : bh / p.H;      // packed: samples in mtmp_row_starts' balanced order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, half = lane >> 5;
    int kvl = p.kv_len ? min(p.kv_len[b], p.N) : p.N;
    const int row0 = p.row_start ? p.row_start[b] : b * p.N;     // (packed stream: AttnArgs::row_start)
    const int Nq = p.row_start ? max(kvl, 0) : p.N;              // rows of this sample: its queries, and the keys that get a gradient row
    if (kt * 128 >= Nq) return;
    const bool uniform = kvl <= 0;            // forward = uniform average: p = 1/N (K = 0 below), dS = 0
    if (uniform) kvl = Nq;
    const size_t base = (size_t)row0 * p.ld_qkv + hd * DH;
    const T* Qb = p.q + base; const T* Kb = p.k + base; const T* Vb = p.v + base;
    const T* dOb = p.d_o + (size_t)row0 * p.ld_do + hd * DH;
    const float* Lb = p.lse + ((size_t)b * p.H + hd) * p.N;
    const float* Db = p.delta + ((size_t)b * p.H + hd) * p.N;
    const int kw0 = kt * 128 + wave * 32;      // first key of this wave
    const int key = kw0 + r;                   // this lane's key (column of S)
    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    if (kt * 128 < kvl) {                      // workgroup-uniform: keys past kv_len get zero gradients
        const float c2 = p.scale * LOG2E;
        Frag<T> kf[4], vf[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            kf[c] = frag_keep(frag_scale<T>(frag_load<T>(Kb + (size_t)min(key, kvl - 1) * p.ld_qkv + 16 * c + 8 * half), c2),
                              key < kvl && !uniform);
            vf[c] = frag_keep(frag_load<T>(Vb + (size_t)min(key, kvl - 1) * p.ld_qkv + 16 * c + 8 * half), key < kvl);
        }
        const int nq = (Nq + KT - 1) / KT;
        TileStream<T> qs, os;
        qs.init(Qb, p.ld_qkv, Nq, tid);
        os.init(dOb, p.ld_do, Nq, tid);
        TileR<T> qreg, oreg;
        // lse / delta of the tile's 64 query rows ride along in wave 0.  The loads are unconditional (clamped
        // address) and their values are not touched before the put: a guarded load, or any arithmetic on the
        // loaded value here, makes the compiler wait for it (vmcnt(0): ALL the tile loads just issued) on the
        // spot, wave 0 then reaches the next barrier a full memory latency late and the other three wait.
        const int lrow = tid & (KT - 1);
        float lreg, dreg;
        auto fetch = [&](int t) {
            qreg = qs.fetch(t);
            oreg = os.fetch(t);
            const int qn = min(t * KT + lrow, Nq - 1);
            lreg = Lb[qn];
            dreg = Db[qn];
        };
        auto put = [&](int t) {
            char* st_ = smem_raw + (t & 1) * dkdv_stage_bytes<T>();
            DT::put(st_, qreg, tid);
            DT::put(st_ + DT::BYTES, oreg, tid);
            if (tid < KT) {                    // C operands of the score products: -lse (-inf past N), -delta
                float* sL = reinterpret_cast<float*>(st_ + 2 * DT::BYTES);
                sL[tid] = (t * KT + tid < Nq) ? -lreg : -INFINITY;
                sL[KT + tid] = (t * KT + tid < Nq) ? -dreg : 0.f;
            }
        };
        fetch(0);
        put(0);
        if (nq > 1) fetch(1);
        for (int it = 0; it < nq; ++it) {
            __syncthreads();                   // tile `it` visible; the other stage is free for tile it + 1
            if (it + 1 < nq) put(it + 1);
            if (it + 2 < nq) fetch(it + 2);
            if (kw0 < kvl) {                   // wave-uniform
                const char* sQ = smem_raw + (it & 1) * dkdv_stage_bytes<T>();
                const char* sdO = sQ + DT::BYTES;
                const float* sL = reinterpret_cast<const float*>(sQ + 2 * DT::BYTES);
                const float* sD = sL + KT;
                // A 64-query tile = two 32-query blocks, software-pipelined INSIDE the wave so that the matrix pipe
                // and the vector ALU run side by side:
                //   phase 0:  S,dP(block 0)                          8 MFMA   (+ LDS reads of block 1's operands)
                //   phase 1:  S,dP(block 1)  ||  exp/mul/cvt(block 0)  8 MFMA beside 48 VALU
                //   phase 2:  dV,dK(block 0) ||  exp/mul/cvt(block 1)  8 MFMA beside 48 VALU
                //   phase 3:  dV,dK(block 1)                         8 MFMA
                // sched_group_barrier lays the instruction order out (1 MFMA : 2 transcendental : 4 VALU per gap =
                // 32 issue cycles beside a 32-cycle MFMA); sched_barrier(0) fences the phases.
                auto load_rowconst = [&](int qb, f32x16& cL, f32x16& cD) {   // row t = query 32qb + 16(t>>3) + 8half + (t&7)
#pragma unroll
                    for (int h8 = 0; h8 < 2; ++h8)
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + 32 * qb + 16 * h8 + 8 * half + 4 * v);
                            const f32x4 d4 = *reinterpret_cast<const f32x4*>(sD + 32 * qb + 16 * h8 + 8 * half + 4 * v);
#pragma unroll
                            for (int i = 0; i < 4; ++i) { cL[8 * h8 + 4 * v + i] = l4[i]; cD[8 * h8 + 4 * v + i] = d4[i]; }
                        }
                };
                auto load_rows = [&](int qb, Frag<T> (&qa)[4], Frag<T> (&oa)[4]) {
                    const int row = 32 * qb + swz23(r);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { qa[c] = DT::row_frag(sQ, row, c, half); oa[c] = DT::row_frag(sdO, row, c, half); }
                };
                auto load_tr = [&](int qb, Frag<T> (&trf)[2][4]) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int q16 = 32 * qb + 16 * s;
                        trf[s][0] = DT::tr_frag(sdO, q16, 0, lane);
                        trf[s][1] = DT::tr_frag(sdO, q16, 32, lane);
                        trf[s][2] = DT::tr_frag(sQ, q16, 0, lane);
                        trf[s][3] = DT::tr_frag(sQ, q16, 32, lane);
                    }
                };
                auto scores = [&](f32x16& st, f32x16& ds, const f32x16& cL, const f32x16& cD, const Frag<T> (&qa)[4],
                                  const Frag<T> (&oa)[4]) {
                    // the score chain first, the dP chain behind it: the exponentials of the next phase read `st`, whose last MFMA
                    // has then been out for four MFMA times instead of one
                    st = mma_c<T>(qa[0], kf[0], cL);
#pragma unroll
                    for (int c = 1; c < 4; ++c) mma<T>(st, qa[c], kf[c]);
                    ds = mma_c<T>(oa[0], vf[0], cD);
#pragma unroll
                    for (int c = 1; c < 4; ++c) mma<T>(ds, oa[c], vf[c]);
                };
                auto probs = [&](f32x16& st, f32x16& ds, Frag<T> (&pf)[2], Frag<T> (&dsf)[2]) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const float pv = fast_exp2(st[t]);
                        st[t] = pv;
                        ds[t] *= pv;
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) { pf[s] = frag_from_acc<T>(st, s); dsf[s] = frag_from_acc<T>(ds, s); }
                };
                // dV^T += dO^T P, dK^T += Q^T dS: the transposed tile as the A operand, so the head dimension lands on the accumulator
                // registers and the KEY on the lane -- a lane then stores 4 consecutive head-dimension elements of its own key row
                // (8-byte pieces, 32 per lane) instead of 64 two-byte elements of 16 different rows (round 3: a 3.8 us store tail per
                // workgroup, four rounds of workgroups per launch)
                auto grads = [&](const Frag<T> (&pf)[2], const Frag<T> (&dsf)[2], const Frag<T> (&trf)[2][4]) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        mma<T>(dv0, trf[s][0], pf[s]);
                        mma<T>(dv1, trf[s][1], pf[s]);
                        mma<T>(dk0, trf[s][2], dsf[s]);
                        mma<T>(dk1, trf[s][3], dsf[s]);
                    }
                };
                f32x16 st0, ds0, st1, ds1, cL1, cD1;
                Frag<T> qa1[4], oa1[4], pf0[2], dsf0[2], pf1[2], dsf1[2];
                {   // phase 0
                    f32x16 cL0, cD0;
                    Frag<T> qa0[4], oa0[4];
                    load_rowconst(0, cL0, cD0);
                    load_rows(0, qa0, oa0);
                    load_rowconst(1, cL1, cD1);
                    load_rows(1, qa1, oa1);
                    scores(st0, ds0, cL0, cD0, qa0, oa0);
                }
                __builtin_amdgcn_sched_barrier(0);
                {   // phase 1
                    scores(st1, ds1, cL1, cD1, qa1, oa1);
                    probs(st0, ds0, pf0, dsf0);
                    if (sizeof(T) == 2) {
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                {   // phase 2
                    Frag<T> trf0[2][4];
                    load_tr(0, trf0);
                    grads(pf0, dsf0, trf0);
                    probs(st1, ds1, pf1, dsf1);
                    if (sizeof(T) == 2) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
                        __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);     // under the LDS latency
                        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (g < 6) __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                            if (g < 6) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                {   // phase 3
                    Frag<T> trf1[2][4];
                    load_tr(1, trf1);
                    grads(pf1, dsf1, trf1);
                }
            }
        }
    }
    // dK^T / dV^T: rows = head dimension (registers), cols = this wave's keys (lanes); keys past kv_len (and the whole dK of a
    // fully masked sample) are written as zeros
    if (key < Nq) {
        const bool live = key < kvl;
        const float ks = (live && !uniform) ? p.scale : 0.f, vs_ = live ? 1.f : 0.f;
        const size_t orow = ((size_t)row0 + key) * p.ld_dqkv + hd * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& ak = dt ? dk1 : dk0;
                const f32x16& av = dt ? dv1 : dv0;
                const int d0 = 32 * dt + 8 * g + 4 * half;
                // (a select, not a product with 0: a key past kv_len may hold NaN / inf garbage in its own column)
                store4<T>(p.dk + orow + d0, ks != 0.f ? ak[4 * g] * ks : 0.f, ks != 0.f ? ak[4 * g + 1] * ks : 0.f,
                          ks != 0.f ? ak[4 * g + 2] * ks : 0.f, ks != 0.f ? ak[4 * g + 3] * ks : 0.f);
                store4<T>(p.dv + orow + d0, vs_ != 0.f ? av[4 * g] : 0.f, vs_ != 0.f ? av[4 * g + 1] : 0.f,
                          vs_ != 0.f ? av[4 * g + 2] : 0.f, vs_ != 0.f ? av[4 * g + 3] : 0.f);
            }
    }
}

// ---- dK/dV, 64 keys per wave (bf16 build, round 5): ONE wave per SIMD, the accumulators in the AGPR half of the register file.
// Why (round-4 ablations): in the 32-key kernel a wave's 32 MFMAs per tile stand against ~290 other instructions
// (64 LDS fragment reads, 150 vector, 64 scalar, 18 waits); the MFMAs alone and everything else alone each take ~100 of its 139 us.
// Here every Q / dO fragment read from LDS serves TWO key blocks (64 MFMAs per tile against the same 64 reads), and the structure
// the compiler could not be talked into (round 4: 176-192 spills in VGPR form, 492 v_accvgpr moves per tile in AGPR form) is
// written down by hand:
//   * dV^T / dK^T of the wave's 64 keys (128 registers) and its K / V operand fragments (64) are `"+a"` / `"a"` operands of
//     inline-asm MFMAs: they live in a[0:191] for the whole kernel, the compiler allocates and tracks them (no literal register
//     names), and no vector instruction touches them before the epilogue.
//   * S' / dP' are written by asm MFMAs into VGPRs (B operand from the AGPRs, C = the -LSE / -delta rows) and read by the
//     exponentials one SLOT later.  The compiler does not know these statements are MFMAs, so the MFMA -> VALU read distance
//     (11 wait states for an 8-pass MFMA) is guaranteed by construction: a slot issues its score products FIRST, then the eight
//     gradient products, and every step ends in sched_barrier(0) -- nothing moves across, the first reader stands >= 8 MFMAs
//     behind the last writer.  P / dS leave the vector unit one slot before the gradient products read them.
//   * One slot = one 32 x 32 unit (query block, key block) in three stages of a software pipeline that never drains:
//         S', dP' of unit u + 1 (8 MFMA)  |  exp2, mul, cvt of unit u (one score per MFMA gap)  |  dV, dK of unit u - 1 (8 MFMA)
//     16 MFMAs beside 48 vector instructions and (every other slot) 32 LDS reads: per gap 1 v_exp + 1 v_mul + 1 v_cvt_pk + 2
//     ds_read = ~24 issue cycles beside a 32-cycle MFMA (MI355X_MICROARCH: <= 24 hide at one wave per SIMD).
//   * Operand registers roll: a Q / dO row fragment, a row-constant group or a transposed fragment is reloaded for the NEXT slot
//     that needs it right behind the MFMA that read it last (A / B operands have no write-after-read hazard; the row constants
//     are reloaded one step later, and an LDS return is > 50 cycles away anyway).
//   * Three LDS stages, one barrier per tile in FRONT of the tile's last slot: that slot already loads the next tile's first
//     fragments, so no MFMA ever waits for an LDS round trip behind a barrier.
// Streams under BWD64_MIN_ROWS rows (config 2: 1005 / 54 / 133 tokens) and the fp32 parity build keep the 32-key kernel.
#define MTMP_MFMA_FIRST(d, a, b, c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c))
#define MTMP_MFMA_NEXT(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b))
#define MTMP_MFMA_ACC(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

#ifdef MTMP_LAB_CLOCK                                         // lab builds only (tools/dbg/dkdv64_clock.py): s_memtime stamps of the first workgroups
__device__ long long mtmp_dbg_stamps[64 * 4 * 160];
#define STAMP(k) do { if (wg < 64 && lane == 0 && (k) < 160) mtmp_dbg_stamps[(wg * 4 + wave) * 160 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
constexpr int DKV64_KEYS = 256;                              // keys per workgroup
// Launches whose longest stream is shorter keep the 32-key kernel.  Measured (round 5, B 64, H 4, one box, dQ + dK/dV in us): N 600:
// 127 vs 119 for the 32-key kernel, N 1005: 261 vs 260 (in the step 316 vs 302), N 2005: 922 vs 960.  One wave per SIMD exposes
// every workgroup's prologue + epilogue (~8.7 us per workgroup, 35 us of a 135 us launch at N 1005: four rounds of 256 workgroups
// with nothing beside them on the CU); the tile loop itself holds 70 % of the MFMA rate (the 32-key kernel: 41 % over the launch).
// The fixed part is amortised over N / 64 tiles, so the form pays from ~1500 rows on (configs[4]: TIE-len 2000).
constexpr int BWD64_MIN_ROWS = 1536;
constexpr int BWD64_STAGES = 3;
constexpr int DKV64_OUT_BYTES = 2 * 64 * LDT * 2;            // per wave: dK | dV of its 64 keys as [key][LDT] bf16 rows (epilogue staging)

// a value DEFINED in the accumulator file: later "a" operands then need no copy (an ordinary value given to an "a" operand is
// copied into a fresh AGPR tuple in front of every statement that reads it)
MTMP_DEV bf16x8 to_agpr(bf16x8 x) {
    bf16x8 y;
    asm volatile("" : "=a"(y) : "0"(x));
    return y;
}
struct PackedUnit { u32x4_t p[2], d[2]; };                   // P and dS of a unit as the B fragments of k-steps 0, 1
MTMP_DEV unsigned cvt_pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv64_kernel(Grouped<AttnBwdArgs<bf16>> grp) {
    using T = bf16;
    using DT = DualTile<T>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = grp_find(grp, wg);
    const AttnBwdArgs<T>& p = grp.seg[seg];
    const int w = wg - grp.first[seg];
    const int nkt = (p.N + DKV64_KEYS - 1) / DKV64_KEYS;
    const int kt = w % nkt, bh = w / nkt, hd = bh % p.H;
    const int b = p.row_start ? p.row_start[p.B + 1 + bh / p.H] : bh / p.H;      // packed: samples in mtmp_row_starts' balanced order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, half = lane >> 5;
    int kvl = p.kv_len ? min(p.kv_len[b], p.N) : p.N;
    const int row0 = p.row_start ? p.row_start[b] : b * p.N;
    const int Nq = p.row_start ? max(kvl, 0) : p.N;
    if (kt * DKV64_KEYS >= Nq) return;
    STAMP(0);
    const bool uniform = kvl <= 0;            // forward = uniform average: p = 1/N (K = 0 below), dS = 0
    if (uniform) kvl = Nq;
    const size_t base = (size_t)row0 * p.ld_qkv + hd * DH;
    const T* Qb = p.q + base; const T* Kb = p.k + base; const T* Vb = p.v + base;
    const T* dOb = p.d_o + (size_t)row0 * p.ld_do + hd * DH;
    const float* Lb = p.lse + ((size_t)b * p.H + hd) * p.N;
    const float* Db = p.delta + ((size_t)b * p.H + hd) * p.N;
    const int kw0 = kt * DKV64_KEYS + wave * 64;             // first key of this wave
    f32x16 dv00 = {0}, dv01 = {0}, dk00 = {0}, dk01 = {0}, dv10 = {0}, dv11 = {0}, dk10 = {0}, dk11 = {0};   // d<k|v><key block><dh half>
    if (kt * DKV64_KEYS < kvl) {              // workgroup-uniform: keys past kv_len get zero gradients
        const float c2 = p.scale * LOG2E;
        Frag<T> kx[2][4], vx[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int key = kw0 + 32 * kb + r;
                kx[kb][c] = frag_load<T>(Kb + (size_t)min(key, kvl - 1) * p.ld_qkv + 16 * c + 8 * half);
                vx[kb][c] = frag_load<T>(Vb + (size_t)min(key, kvl - 1) * p.ld_qkv + 16 * c + 8 * half);
            }
        const int nq = (Nq + KT - 1) / KT;
        TileStream<T> qs, os;
        qs.init(Qb, p.ld_qkv, Nq, tid);
        os.init(dOb, p.ld_do, Nq, tid);
        TileR<T> qreg, oreg;
        const int lrow = tid & (KT - 1);
        float lreg, dreg;
        auto stage = [&](int t) { return smem_raw + (t % BWD64_STAGES) * dkdv_stage_bytes<T>(); };
        // Staging of a tile (global -> registers one tile ahead, registers -> LDS) in six pieces, so that an active wave can issue
        // them between the MFMAs of a slot that has no LDS reads of its own (as one block in front of the slot they cost ~800
        // cycles per tile with an empty matrix pipe: in-kernel stamps, round 5).  Pieces 0-2 put tile tp, 3-5 fetch tile tf; the
        // loads are unconditional (clamped addresses) and their values untouched before the put (see the 32-key kernel).
        auto stage_piece = [&](int k, int tp, int tf) {
            if (k < 3) {
                if (tp >= nq) return;
                char* st_ = stage(tp);
                if (k == 0) DT::put(st_, qreg, tid);
                else if (k == 1) DT::put(st_ + DT::BYTES, oreg, tid);
                else if (tid < KT) {               // C operands of the score products: -lse (-inf past N), -delta
                    float* sL = reinterpret_cast<float*>(st_ + 2 * DT::BYTES);
                    sL[tid] = (tp * KT + tid < Nq) ? -lreg : -INFINITY;
                    sL[KT + tid] = (tp * KT + tid < Nq) ? -dreg : 0.f;
                }
            } else {
                if (tf >= nq) return;
                if (k == 3) qreg = qs.fetch(tf);
                else if (k == 4) oreg = os.fetch(tf);
                else {
                    const int qn = min(tf * KT + lrow, Nq - 1);
                    lreg = Lb[qn];
                    dreg = Db[qn];
                }
            }
        };
        auto stage_all = [&](int tp, int tf) {
#pragma unroll
            for (int k = 0; k < 6; ++k) stage_piece(k, tp, tf);
        };
        stage_all(nq, 0);                      // fetch tile 0
        bf16x8 kf[2][4], vf[2][4];             // "a" operands from here on
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool live = kw0 + 32 * kb + r < kvl;
                kf[kb][c] = to_agpr(frag_keep(frag_scale<T>(kx[kb][c], c2), live && !uniform).v);
                vf[kb][c] = to_agpr(frag_keep(vx[kb][c], live).v);
            }
        stage_all(0, 1);                       // put tile 0, fetch tile 1
        __syncthreads();                       // tile 0 visible
        const bool active = kw0 < kvl;         // wave-uniform: a wave whose keys are all past kv_len only stages tiles
        // rolling operand registers (see the header): rows / row constants of ONE query block, transposed fragments of ONE query block
        bf16x8 qa[4], oa[4], tr[2][4];
        f32x16 cL, cD;
        // LDS addressing: every fragment address is  stage base + a LANE pattern + a compile-time constant.  The five lane patterns
        // (DualTile<bf16>::off at the lane's row / chunk) are computed once; per tile one add each gives the stage's bases, and
        // everything else is the ds_read's immediate offset (the form the compiler found by itself rebuilt ~90 addresses per tile).
        struct LaneOffs { unsigned rowE, rowO, cst, trA, trB; };
        LaneOffs lo;
        {
            const int sr = swz23(r), G = lane >> 4, i16 = lane & 15;
            lo.rowE = DT::off(sr, half);                               // row fragments, k-steps 0 / 2 (+512)
            lo.rowO = DT::off(sr, 2 + half);                           //                k-steps 1 / 3 (+512)
            lo.cst = 2 * DT::BYTES + 32 * half;                        // row constants (floats 8 half ..)
            const int trow = 8 * (G >> 1) + (i16 >> 2), tch = 2 * (G & 1) + ((i16 & 3) >> 1);
            lo.trA = DT::off(trow, tch) + 8 * (i16 & 1);               // transposed fragments: rows q16 + ..
            lo.trB = DT::off(trow + 4, tch) + 8 * (i16 & 1);           //                       rows q16 + 4 + ..
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem_raw;
            lo.rowE += lds0; lo.rowO += lds0; lo.cst += lds0; lo.trA += lds0; lo.trB += lds0;
        }
        typedef __attribute__((address_space(3))) const char lds_cchar;
        auto lds_at = [](unsigned ofs) { return (lds_cchar*)(size_t)ofs; };
        auto bases = [&](int t) {
            const unsigned sb = (unsigned)((t % BWD64_STAGES) * dkdv_stage_bytes<T>());
            LaneOffs x{lo.rowE + sb, lo.rowO + sb, lo.cst + sb, lo.trA + sb, lo.trB + sb};
            asm volatile("" : "+v"(x.rowE), "+v"(x.rowO), "+v"(x.cst), "+v"(x.trA), "+v"(x.trB));      // opaque: keep base + immediate
            return x;
        };
        auto load_rows = [&](const LaneOffs& bs, int qb, int c, bool dO) {
            typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
            const bf16x8 v = *(lds_bf16x8*)(lds_at((c & 1) ? bs.rowO : bs.rowE) + 512 * (c >> 1) + 4096 * qb + (dO ? DT::BYTES : 0));
            if (dO) oa[c] = v; else qa[c] = v;
        };
        auto load_const = [&](const LaneOffs& bs, int qb, int k4, bool dlt) {   // registers 4 k4 .. 4 k4 + 3: queries 32 qb + 16 (k4 >> 1) + 8 half + 4 (k4 & 1) ..
            typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
            const f32x4 v4 = *(lds_f32x4*)(lds_at(bs.cst) + (dlt ? 4 * KT : 0) + 4 * (32 * qb + 16 * (k4 >> 1) + 4 * (k4 & 1)));
#pragma unroll
            for (int i = 0; i < 4; ++i) { if (dlt) cD[4 * k4 + i] = v4[i]; else cL[4 * k4 + i] = v4[i]; }
        };
        auto load_tr = [&](const LaneOffs& bs, int qb, int s2, int j) {         // j: 0, 1 = dO^T (dh halves), 2, 3 = Q^T
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            const int imm = 4096 * qb + 2048 * s2 + 512 * (j & 1) + (j < 2 ? DT::BYTES : 0);
            const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_at(bs.trA) + imm));
            const s16x4 b4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds_at(bs.trB) + imm));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            tr[s2][j] = __builtin_bit_cast(bf16x8, s16x8{a[0], a[1], a[2], a[3], b4[0], b4[1], b4[2], b4[3]});
        };
        // One slot (see the header).  (sN, dN): the unit whose scores are produced; (sC, dC) -> out: the unit whose probabilities are
        // taken; in: the unit whose gradients are accumulated into the key block's four accumulators.  MODE 1: this slot reloads
        // the row operands from (st_rows, qb_rows) and the transposed fragments from (st_tr, qb_tr) behind their last readers;
        // MODE 2: it carries the staging pieces of tiles (tp, tf) instead.
        auto slot = [&](auto mode_tag, f32x16& sN, f32x16& dN, const bf16x8 (&kfb)[4], const bf16x8 (&vfb)[4], f32x16& sC, f32x16& dC,
                        PackedUnit& out, const PackedUnit& in, f32x16& dv0, f32x16& dv1, f32x16& dk0, f32x16& dk1,
                        const LaneOffs& st_rows, int qb_rows, const LaneOffs& st_tr, int qb_tr, int tp, int tf) {
            constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // ---- matrix pipe
                if (i == 0) MTMP_MFMA_FIRST(sN, qa[0], kfb[0], cL);
                else if (i < 4) MTMP_MFMA_NEXT(sN, qa[i], kfb[i]);
                else if (i == 4) MTMP_MFMA_FIRST(dN, oa[0], vfb[0], cD);
                else if (i < 8) MTMP_MFMA_NEXT(dN, oa[i - 4], vfb[i - 4]);
                else {
                    const int s2 = (i - 8) >> 2, j = (i - 8) & 3;
                    if (j == 0) MTMP_MFMA_ACC(dv0, tr[s2][0], in.p[s2]);
                    else if (j == 1) MTMP_MFMA_ACC(dv1, tr[s2][1], in.p[s2]);
                    else if (j == 2) MTMP_MFMA_ACC(dk0, tr[s2][2], in.d[s2]);
                    else MTMP_MFMA_ACC(dk1, tr[s2][3], in.d[s2]);
                }
                // ---- vector unit: one score of the current unit (the product one step behind its exponential: a transcendental
                // result read by the very next vector instruction costs a wait state)
                const float e = fast_exp2(sC[i]);
                if (i > 0) dC[i - 1] *= sC[i - 1];
                sC[i] = e;
                if (i & 1) out.p[i >> 3][(i & 7) >> 1] = cvt_pk_bf16(sC[i - 1], e);
                if (i >= 2 && !(i & 1)) out.d[(i - 2) >> 3][((i - 2) & 7) >> 1] = cvt_pk_bf16(dC[i - 2], dC[i - 1]);
                if (i == 15) {
                    dC[15] *= e;
                    out.d[1][3] = cvt_pk_bf16(dC[14], dC[15]);
                }
                // ---- LDS: operands of the next slots into the registers this step's MFMA has just read
                if (MODE == 1) {
                    if (i < 4) load_rows(st_rows, qb_rows, i, false);
                    else if (i < 8) load_rows(st_rows, qb_rows, i - 4, true);
                    else load_tr(st_tr, qb_tr, (i - 8) >> 2, (i - 8) & 3);
                    if (i >= 1 && i < 5) load_const(st_rows, qb_rows, i - 1, false);
                    else if (i >= 5 && i < 9) load_const(st_rows, qb_rows, i - 5, true);
                }
                // ---- staging of the next tiles, a piece every other step
                if (MODE == 2 && (i & 1) && i < 12) stage_piece(i >> 1, tp, tf);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        f32x16 sA, dA, sB, dB;                 // the two units in flight
        PackedUnit fX, fY;
#pragma unroll
        for (int t = 0; t < 16; ++t) { sB[t] = -INFINITY; dB[t] = 0.f; }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            fX.p[s2] = u32x4_t{0, 0, 0, 0}; fX.d[s2] = u32x4_t{0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) tr[s2][j] = frag_zero<T>().v;
        }
        if (active) {
            const LaneOffs b0 = bases(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                load_rows(b0, 0, c, false); load_rows(b0, 0, c, true);
                load_const(b0, 0, c, false); load_const(b0, 0, c, true);
            }
        }
        const std::integral_constant<int, 0> plain{};
        const std::integral_constant<int, 1> loads{};
        const std::integral_constant<int, 2> staging{};
        STAMP(1);
        for (int it = 0; it < nq; ++it) {
            STAMP(2 + 8 * it);
            const LaneOffs cur = bases(it);
            const LaneOffs nxt = bases(it + 1 < nq ? it + 1 : it);     // (last tile: harmless reloads of its own rows)
            if (active) {
                __builtin_amdgcn_sched_barrier(0);
                STAMP(3 + 8 * it);
                // A: scores (0,0) | probabilities of the previous tile's (1,1) | gradients of its (1,0); staging: put tile it + 1, fetch tile it + 2
                slot(staging, sA, dA, kf[0], vf[0], sB, dB, fY, fX, dv00, dv01, dk00, dk01, cur, 0, cur, 0, it + 1, it + 2);
                STAMP(4 + 8 * it);
                // B: scores (0,1) | probabilities (0,0) | gradients of the previous (1,1); reload: rows of block 1, transposed block 0
                slot(loads, sB, dB, kf[1], vf[1], sA, dA, fX, fY, dv10, dv11, dk10, dk11, cur, 1, cur, 0, 0, 0);
                STAMP(5 + 8 * it);
                // C: scores (1,0) | probabilities (0,1) | gradients (0,0)
                slot(plain, sA, dA, kf[0], vf[0], sB, dB, fY, fX, dv00, dv01, dk00, dk01, cur, 0, cur, 0, 0, 0);
                STAMP(6 + 8 * it);
            } else {
                stage_all(it + 1, it + 2);
            }
            __syncthreads();                   // tile it + 1 visible; stage (it + 2) % 3 is free (its last readers: slot D of tile it - 1)
            if (active) {
                STAMP(7 + 8 * it);
                // D: scores (1,1) | probabilities (1,0) | gradients (0,1); reload: rows of the NEXT tile's block 0, transposed block 1
                slot(loads, sB, dB, kf[1], vf[1], sA, dA, fX, fY, dv10, dv11, dk10, dk11, nxt, 0, cur, 1, 0, 0);
                STAMP(8 + 8 * it);
            }
        }
        if (active) {                          // drain: probabilities of the last (1,1) beside the gradients of the last (1,0), then its own
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i < 8) {
                    const int s2 = i >> 2, j = i & 3;
                    if (j == 0) MTMP_MFMA_ACC(dv00, tr[s2][0], fX.p[s2]);
                    else if (j == 1) MTMP_MFMA_ACC(dv01, tr[s2][1], fX.p[s2]);
                    else if (j == 2) MTMP_MFMA_ACC(dk00, tr[s2][2], fX.d[s2]);
                    else MTMP_MFMA_ACC(dk01, tr[s2][3], fX.d[s2]);
                }
                const float e = fast_exp2(sB[i]);
                dB[i] *= e;
                sB[i] = e;
                if (i & 1) {
                    fY.p[i >> 3][(i & 7) >> 1] = cvt_pk_bf16(sB[i - 1], e);
                    fY.d[i >> 3][(i & 7) >> 1] = cvt_pk_bf16(dB[i - 1], dB[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                MTMP_MFMA_ACC(dv10, tr[s2][0], fY.p[s2]);
                MTMP_MFMA_ACC(dv11, tr[s2][1], fY.p[s2]);
                MTMP_MFMA_ACC(dk10, tr[s2][2], fY.d[s2]);
                MTMP_MFMA_ACC(dk11, tr[s2][3], fY.d[s2]);
            }
        }
        STAMP(150);
        // the compiler reads the accumulators below without knowing that MFMAs wrote them: cover the last products' passes here
        asm volatile("s_nop 15\n\ts_nop 15" : "+a"(dv00), "+a"(dv01), "+a"(dk00), "+a"(dk01), "+a"(dv10), "+a"(dv11), "+a"(dk10), "+a"(dk11));
    }
    // Epilogue.  dK^T / dV^T: rows = head dimension (registers), columns = this wave's keys (lanes): a lane owns 4-element pieces of
    // its key's rows, so direct stores are 64 eight-byte pieces per lane at a row stride (store-issue bound: ~9.8k cycles per
    // workgroup, stamps).  The wave's two 64 x 64 tiles go through a wave-private LDS region of their own (no barrier: the
    // stages may still be read by slower waves) and leave as whole 128-byte rows, 16 sixteen-byte stores per lane.  Keys past
    // kv_len (and the whole dK of a fully masked sample) are written as zeros -- a select, not a product with 0: such a key
    // may hold NaN / inf garbage in its own column.
    T* sOut = reinterpret_cast<T*>(smem_raw + BWD64_STAGES * dkdv_stage_bytes<T>() + wave * DKV64_OUT_BYTES);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int key = kw0 + 32 * kb + r;
        const bool live = key < kvl;
        const bool wk = live && !uniform;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& ak = kb ? (dt ? dk11 : dk10) : (dt ? dk01 : dk00);
                const f32x16& av = kb ? (dt ? dv11 : dv10) : (dt ? dv01 : dv00);
                const int d0 = 32 * dt + 8 * g + 4 * half;
                store4<T>(sOut + (32 * kb + r) * LDT + d0, wk ? ak[4 * g] * p.scale : 0.f, wk ? ak[4 * g + 1] * p.scale : 0.f,
                          wk ? ak[4 * g + 2] * p.scale : 0.f, wk ? ak[4 * g + 3] * p.scale : 0.f);
                store4<T>(sOut + (64 + 32 * kb + r) * LDT + d0, live ? av[4 * g] : 0.f, live ? av[4 * g + 1] : 0.f,
                          live ? av[4 * g + 2] : 0.f, live ? av[4 * g + 3] : 0.f);
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: in-order LDS, no barrier needed
    {
        const int rsub = lane >> 3, ch = lane & 7;
        u32x4_t ok[8], ov[8];
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            ok[ps] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(sOut + (8 * ps + rsub) * LDT) + 16 * ch);
            ov[ps] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(sOut + (64 + 8 * ps + rsub) * LDT) + 16 * ch);
        }
        T* krow = p.dk + ((size_t)row0 + kw0 + rsub) * p.ld_dqkv + hd * DH + 8 * ch;
        T* vrow = p.dv + ((size_t)row0 + kw0 + rsub) * p.ld_dqkv + hd * DH + 8 * ch;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps)
            if (kw0 + 8 * ps + rsub < Nq) {
                *reinterpret_cast<u32x4_t*>(krow + (size_t)8 * ps * p.ld_dqkv) = ok[ps];
                *reinterpret_cast<u32x4_t*>(vrow + (size_t)8 * ps * p.ld_dqkv) = ov[ps];
            }
    }
    STAMP(151);
}

template <typename T> size_t fwd_smem() { return (size_t)2 * fwd_stage_elems<T>() * sizeof(T); }   // >= 4 x 64 x LDT staging rows
template <typename T> size_t dq_smem() { return (size_t)2 * dq_stage_bytes<T>(); }
template <typename T> size_t dkdv_smem() { return (size_t)2 * dkdv_stage_bytes<T>(); }

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

// segments -> one grid: first[i] = first block of segment i (largest work first is the caller's business)
template <typename A, typename F> Grouped<A> make_group(int n, const A* segs, F blocks_of, int& total) {
    Grouped<A> g;
    total = 0;
    for (int i = 0; i < GRP_MAX; ++i) {
        g.seg[i] = segs[i < n ? i : 0];
        g.first[i] = total;
        if (i < n) total += blocks_of(segs[i]);
    }
    g.first[GRP_MAX] = total;
    return g;
}

template <typename T> int launch_fwd(int n, const AttnArgs<T>* segs, hipStream_t st) {
    int nwg;
    const Grouped<AttnArgs<T>> g = make_group(n, segs, [](const AttnArgs<T>& a) { return ((a.N + FWD_QWG - 1) / FWD_QWG) * a.H * a.B; }, nwg);
    const size_t sm = fwd_smem<T>();
    if (int e = set_smem(attn_fwd_kernel<T>, sm)) return e;
    hipLaunchKernelGGL(attn_fwd_kernel<T>, dim3(nwg), dim3(256), sm, st, g);
    MTMP_CHECK_LAUNCH("mtmp_attn_fwd");
    return MTMP_OK;
}

template <typename T> int launch_bwd(int n, const AttnBwdArgs<T>* segs, hipStream_t st) {
    int nwg;
    const Grouped<AttnBwdArgs<T>> g = make_group(n, segs, [](const AttnBwdArgs<T>& a) { return ((a.N + 127) / 128) * a.H * a.B; }, nwg);
    if (int e = set_smem(attn_bwd_dq_kernel<T>, dq_smem<T>())) return e;
    hipLaunchKernelGGL(attn_bwd_dq_kernel<T>, dim3(nwg), dim3(256), dq_smem<T>(), st, g);      // also writes delta
    MTMP_CHECK_LAUNCH("mtmp_attn_bwd(dq)");
    if constexpr (sizeof(T) == 2) {
        // long streams (the vital-sign stream): 64 keys per wave, 256 per workgroup; short ones keep the 32-key kernel
        int nmax = 0;
        for (int i = 0; i < n; ++i) nmax = segs[i].N > nmax ? segs[i].N : nmax;
        if (nmax >= BWD64_MIN_ROWS) {
            int nwg64;
            const Grouped<AttnBwdArgs<T>> g64 =
                make_group(n, segs, [](const AttnBwdArgs<T>& a) { return ((a.N + DKV64_KEYS - 1) / DKV64_KEYS) * a.H * a.B; }, nwg64);
            const size_t sm64 = (size_t)BWD64_STAGES * dkdv_stage_bytes<T>() + 4 * DKV64_OUT_BYTES;
            if (int e = set_smem(attn_bwd_dkdv64_kernel, sm64)) return e;
            hipLaunchKernelGGL(attn_bwd_dkdv64_kernel, dim3(nwg64), dim3(256), sm64, st, g64);
            MTMP_CHECK_LAUNCH("mtmp_attn_bwd(dkdv64)");
            return MTMP_OK;
        }
    }
    if (int e = set_smem(attn_bwd_dkdv_kernel<T>, dkdv_smem<T>())) return e;
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel<T>, dim3(nwg), dim3(256), dkdv_smem<T>(), st, g);
    MTMP_CHECK_LAUNCH("mtmp_attn_bwd(dkdv)");
    return MTMP_OK;
}

bool attn_shape_ok(int B, int N, int H, int ld_a, int ld_b) {
    return B > 0 && N > 0 && H > 0 && H <= 4 && ld_a >= H * DH && ld_b >= H * DH && (ld_a % 8) == 0 && (ld_b % 8) == 0;
}

}  // namespace

template <typename T>
int fwd_entry(int n, const void* const* q, const void* const* k, const void* const* v, void* const* o, const void* const* res,
              void* const* o_res, float* const* lse, const int32_t* const* kv_len, const int32_t* const* row_start,
              const float* const* key_norms, const int* N, const int* ld_qkv, const int* ld_o, int B, int H, float scale,
              hipStream_t st) {
    AttnArgs<T> a[GRP_MAX];
    for (int i = 0; i < n; ++i)
        a[i] = AttnArgs<T>{(const T*)q[i], (const T*)k[i], (const T*)v[i], (T*)o[i], res ? (const T*)res[i] : nullptr,
                           o_res ? (T*)o_res[i] : nullptr, lse[i], kv_len ? kv_len[i] : nullptr, key_norms ? key_norms[i] : nullptr,
                           B, N[i], H, ld_qkv[i], ld_o[i], scale, row_start ? row_start[i] : nullptr};
    return launch_fwd<T>(n, a, st);
}

// n <= 3 streams in ONE launch (all arrays are HOST arrays of n entries; res / o_res / kv_len / row_start / key_norms may be NULL
// as a whole or per entry).  mtmp_attn_fwd is the n = 1 form.  row_start[i] != NULL: stream i is PACKED -- int32[B] device, sample
// b's kv_len[i][b] tokens are rows row_start[i][b] .. of the q / k / v / o / res buffers (mtmp_row_starts; AttnArgs::row_start),
// N[i] is the longest sample the buffers and lse were sized for.
extern "C" int mtmp_attn_fwd_grouped(int dtype, int n, const void* const* q, const void* const* k, const void* const* v,
                                     void* const* o, const void* const* res, void* const* o_res, float* const* lse,
                                     const int32_t* const* kv_len, const int32_t* const* row_start, const float* const* key_norms,
                                     const int* N, const int* ld_qkv, const int* ld_o, int B, int H, float scale, void* stream) {
    MTMP_CHECK_ARG(n >= 1 && n <= GRP_MAX && q && k && v && o && lse && N && ld_qkv && ld_o, "mtmp_attn_fwd: bad group (n=%d)", n);
    for (int i = 0; i < n; ++i) {
        MTMP_CHECK_ARG(q[i] && k[i] && v[i] && o[i] && lse[i], "mtmp_attn_fwd: null pointer (stream %d)", i);
        MTMP_CHECK_ARG((!res || !res[i]) == (!o_res || !o_res[i]), "mtmp_attn_fwd: res and o_res must be given together");
        MTMP_CHECK_ARG(!(row_start && row_start[i]) || (kv_len && kv_len[i]), "mtmp_attn_fwd: a packed stream needs kv_len");
        MTMP_CHECK_ARG(attn_shape_ok(B, N[i], H, ld_qkv[i], ld_o[i]), "mtmp_attn_fwd: bad shape B=%d N=%d H=%d ld=%d/%d", B, N[i], H,
                       ld_qkv[i], ld_o[i]);
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) return fwd_entry<float>(n, q, k, v, o, res, o_res, lse, kv_len, row_start, key_norms, N, ld_qkv, ld_o, B, H, scale, st);
    if (dtype == 1) return fwd_entry<bf16>(n, q, k, v, o, res, o_res, lse, kv_len, row_start, key_norms, N, ld_qkv, ld_o, B, H, scale, st);
    mtmp_set_error("mtmp_attn_fwd: unknown dtype %d", dtype);
    return MTMP_ERR_ARG;
}

extern "C" int mtmp_attn_fwd(int dtype, const void* q, const void* k, const void* v, void* o, const void* res,
                             void* o_res, float* lse, const int32_t* kv_len, const float* key_norms, int B, int N, int H,
                             int ld_qkv, int ld_o, float scale, void* stream) {
    return mtmp_attn_fwd_grouped(dtype, 1, &q, &k, &v, &o, &res, &o_res, &lse, &kv_len, nullptr, &key_norms, &N, &ld_qkv, &ld_o, B, H,
                                 scale, stream);
}

template <typename T>
int bwd_entry(int n, const void* const* q, const void* const* k, const void* const* v, const void* const* o, const void* const* d_o,
              const float* const* lse, const int32_t* const* kv_len, const int32_t* const* row_start, void* const* dq,
              void* const* dk, void* const* dv, float* const* delta_ws, const int* N, const int* ld_qkv, const int* ld_o,
              const int* ld_do, const int* ld_dqkv, int B, int H, float scale, hipStream_t st) {
    AttnBwdArgs<T> a[GRP_MAX];
    for (int i = 0; i < n; ++i)
        a[i] = AttnBwdArgs<T>{(const T*)q[i], (const T*)k[i], (const T*)v[i], (const T*)d_o[i], lse[i], delta_ws[i],
                              kv_len ? kv_len[i] : nullptr, (T*)dq[i], (T*)dk[i], (T*)dv[i], B, N[i], H, ld_qkv[i], ld_do[i],
                              ld_dqkv[i], scale, (const T*)o[i], ld_o[i], row_start ? row_start[i] : nullptr};
    return launch_bwd<T>(n, a, st);
}

extern "C" int mtmp_attn_bwd_grouped(int dtype, int n, const void* const* q, const void* const* k, const void* const* v,
                                     const void* const* o, const void* const* d_o, const float* const* lse,
                                     const int32_t* const* kv_len, const int32_t* const* row_start, void* const* dq,
                                     void* const* dk, void* const* dv, float* const* delta_ws, const int* N, const int* ld_qkv,
                                     const int* ld_o, const int* ld_do, const int* ld_dqkv, int B, int H, float scale, void* stream) {
    MTMP_CHECK_ARG(n >= 1 && n <= GRP_MAX && q && k && v && o && d_o && lse && dq && dk && dv && delta_ws && N && ld_qkv && ld_o &&
                       ld_do && ld_dqkv, "mtmp_attn_bwd: bad group (n=%d)", n);
    for (int i = 0; i < n; ++i) {
        MTMP_CHECK_ARG(q[i] && k[i] && v[i] && o[i] && d_o[i] && lse[i] && dq[i] && dk[i] && dv[i] && delta_ws[i],
                       "mtmp_attn_bwd: null pointer (stream %d)", i);
        MTMP_CHECK_ARG(!(row_start && row_start[i]) || (kv_len && kv_len[i]), "mtmp_attn_bwd: a packed stream needs kv_len");
        MTMP_CHECK_ARG(attn_shape_ok(B, N[i], H, ld_qkv[i], ld_o[i]) && attn_shape_ok(B, N[i], H, ld_do[i], ld_dqkv[i]),
                       "mtmp_attn_bwd: bad shape B=%d N=%d H=%d", B, N[i], H);
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        return bwd_entry<float>(n, q, k, v, o, d_o, lse, kv_len, row_start, dq, dk, dv, delta_ws, N, ld_qkv, ld_o, ld_do, ld_dqkv, B, H, scale, st);
    if (dtype == 1)
        return bwd_entry<bf16>(n, q, k, v, o, d_o, lse, kv_len, row_start, dq, dk, dv, delta_ws, N, ld_qkv, ld_o, ld_do, ld_dqkv, B, H, scale, st);
    mtmp_set_error("mtmp_attn_bwd: unknown dtype %d", dtype);
    return MTMP_ERR_ARG;
}

extern "C" int mtmp_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                             const float* lse, const int32_t* kv_len, void* dq, void* dk, void* dv, float* delta_ws,
                             int B, int N, int H, int ld_qkv, int ld_o, int ld_do, int ld_dqkv, float scale,
                             void* stream) {
    return mtmp_attn_bwd_grouped(dtype, 1, &q, &k, &v, &o, &d_o, &lse, &kv_len, nullptr, &dq, &dk, &dv, &delta_ws, &N, &ld_qkv, &ld_o,
                                 &ld_do, &ld_dqkv, B, H, scale, stream);
}

// ------------------------------------------------------------------------------------------------------------------------
// Attention of ONE query per sample -- the CLS token of the LAST fusion layer of the vital-sign stream.  tri_mbt_vsltcls.py:248
// reads nothing of the encoder's result but outputs[0][:, 0, :]: in the last layer only the CLS row's attention output, FFN and
// residuals feed the loss (every other query row of that layer is dead code; its keys and values are not -- K / V come from
// all rows).  Forward: s_k = scale q_cls.k_k over the kv_len[b] valid keys, p = softmax(s), o = sum p_k v_k, r1 = o + residual
// (o rounded to T first, like the dense kernel's epilogue).  Backward: dO arrives for the CLS query only, so dQ is one row and
// dK / dV are rank one per (sample, head): dk_k = scale dS_k q, dv_k = p_k dO, dq = scale sum dS_k k_k with dS = p (dP - delta).
// One workgroup per (sample, head); plain fp32 FMAs (a few hundred thousand MACs per workgroup: no MFMA needed); the scores
// live in LDS.  Same packed / padded addressing as the dense kernels (AttnArgs::row_start).
namespace {
template <typename T> struct AttnClsArgs {
    const T* q; const T* k; const T* v; int ld_qkv;
    const T* res; int ld_res;              // forward: the layer input z, row-aligned with q (residual of encoder.py:27)
    T* o_cls; T* r1_cls; float* lse;       // [B, H * 64], [B, H * 64], [B, H] (log2 units of the scaled scores)
    const T* d_o;                          // backward: gradient w.r.t. o_cls, [B, H * 64]
    T* dq; T* dk; T* dv; int ld_dqkv;      // backward: DENSE gradient rows (every row of the sample is written)
    const int* kv_len; const int* row_start;
    int B, N, H, cls_tok;
    float scale;
};

// 8 consecutive elements of a row as floats (one 16-byte load for bf16, two for fp32)
template <typename T> MTMP_DEV void load8f(const T* ptr, float (&o)[8]) {
    const f32x4 a = load4<T>(ptr), b = load4<T>(ptr + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
MTMP_DEV float block_reduce(float v, float* red, bool take_max) {      // 256 threads; red: 4 floats of LDS scratch
    v = take_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return take_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

// One workgroup (256 threads) per (sample, head).  Scores: a thread per key (the whole 128-byte key row against q held in
// registers).  P V: thread = (8-dim group dg = tid & 7, key lane kl = tid >> 3): 32 key lanes walk the keys with 16-byte loads,
// their partial sums meet in LDS.
template <typename T> __global__ __launch_bounds__(256) void attn_cls_fwd_kernel(AttnClsArgs<T> p) {
    extern __shared__ float cls_lds[];
    float* s = cls_lds;                                     // [N] scores -> probabilities
    float* red = s + p.N;                                   // [4]
    float* part = red + 4;                                  // [32][64] partial outputs
    const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H, tid = threadIdx.x;
    const int kv_raw = p.kv_len ? p.kv_len[b] : p.N;
    if (p.row_start && kv_raw <= p.cls_tok) {               // packed stream: this sample has no such row (never the case in the fused
        if (tid < DH) {                                     // path, where kv_len >= prefix + CLS): defined outputs, nothing read
            const size_t oi = ((size_t)b * p.H + hd) * DH + tid;
            p.o_cls[oi] = from_f32<T>(0.f);
            p.r1_cls[oi] = from_f32<T>(0.f);
            if (tid == 0) p.lse[b * p.H + hd] = 0.f;
        }
        return;
    }
    // all keys masked (padded layout): the reference's masked_fill(-65504) + softmax = the uniform average over all N keys, as in
    // the dense kernel (`uniform`): every score 0 (q scaled by 0), all N keys taken
    const bool uniform = kv_raw <= 0;
    const int kvl = uniform ? p.N : min(kv_raw, p.N);
    const size_t row0 = p.row_start ? (size_t)p.row_start[b] : (size_t)b * p.N;
    const T* Kb = p.k + row0 * p.ld_qkv + hd * DH;
    const T* Vb = p.v + row0 * p.ld_qkv + hd * DH;
    const T* qrow = p.q + (row0 + p.cls_tok) * p.ld_qkv + hd * DH;
    const float c2 = uniform ? 0.f : p.scale * LOG2E;
    float q[DH];
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) {
        float t[8];
        load8f<T>(qrow + 8 * c, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) q[8 * c + j] = t[j] * c2;
    }
    float mx = -INFINITY;
    for (int k = tid; k < kvl; k += 256) {
        const T* kr = Kb + (size_t)k * p.ld_qkv;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            float t[8];
            load8f<T>(kr + 8 * c, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) a = fmaf(q[8 * c + j], t[j], a);
        }
        s[k] = a;
        mx = fmaxf(mx, a);
    }
    mx = block_reduce(mx, red, true);
    float l = 0.f;
    for (int k = tid; k < kvl; k += 256) {
        const float e = fast_exp2(s[k] - mx);
        s[k] = e;
        l += e;
    }
    l = block_reduce(l, red, false);                        // (its barriers also publish the probabilities)
    const int dg = tid & 7, kl = tid >> 3;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = kl; k < kvl; k += 32) {
        float t[8];
        load8f<T>(Vb + (size_t)k * p.ld_qkv + 8 * dg, t);
        const float pk = s[k];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(pk, t[j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[kl * DH + 8 * dg + j] = acc[j];
    __syncthreads();
    if (tid < DH) {
        float o = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) o += part[i * DH + tid];
        const T ot = from_f32<T>(o / l);
        const size_t oi = ((size_t)b * p.H + hd) * DH + tid;
        p.o_cls[oi] = ot;
        p.r1_cls[oi] = from_f32<T>(to_f32(ot) + to_f32(p.res[(row0 + p.cls_tok) * p.ld_res + hd * DH + tid]));
        if (tid == 0) p.lse[b * p.H + hd] = mx + log2f(l);
    }
}

template <typename T> MTMP_DEV void store8(T* ptr, const float (&v)[8]) {
    store4<T>(ptr, v[0], v[1], v[2], v[3]);
    store4<T>(ptr + 4, v[4], v[5], v[6], v[7]);
}

template <typename T> __global__ __launch_bounds__(256) void attn_cls_bwd_kernel(AttnClsArgs<T> p) {
    extern __shared__ float cls_lds[];
    float* sP = cls_lds;                                    // [N] p_k
    float* sD = sP + p.N;                                   // [N] dS_k
    float* red = sD + p.N;                                  // [4]
    float* part = red + 4;                                  // [32][64] partial dq
    const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H, tid = threadIdx.x;
    const int kv_raw = p.kv_len ? p.kv_len[b] : p.N;
    if (p.row_start && kv_raw <= p.cls_tok) return;         // packed stream without that row: no gradient rows either (see the forward)
    const bool uniform = kv_raw <= 0;                       // forward = uniform average: constant scores, dS = 0, dV = dO / N
    const int kvl = uniform ? p.N : min(kv_raw, p.N);
    const int nrows = p.row_start ? kvl : p.N;              // rows of this sample in the dense gradient buffers
    const size_t row0 = p.row_start ? (size_t)p.row_start[b] : (size_t)b * p.N;
    const T* Kb = p.k + row0 * p.ld_qkv + hd * DH;
    const T* Vb = p.v + row0 * p.ld_qkv + hd * DH;
    const T* qrow = p.q + (row0 + p.cls_tok) * p.ld_qkv + hd * DH;
    const size_t oi = ((size_t)b * p.H + hd) * DH;
    const float c2 = uniform ? 0.f : p.scale * LOG2E, lse = p.lse[b * p.H + hd];
    float q[DH], g[DH];                                     // q (scaled to log2 units) and dO: whole rows in every thread
    float dl = 0.f;
#pragma unroll
    for (int c = 0; c < DH / 8; ++c) {
        float t[8], d[8], o[8];
        load8f<T>(qrow + 8 * c, t);
        load8f<T>(p.d_o + oi + 8 * c, d);
        load8f<T>(p.o_cls + oi + 8 * c, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) { q[8 * c + j] = t[j] * c2; g[8 * c + j] = d[j]; dl = fmaf(d[j], o[j], dl); }
    }
    for (int k = tid; k < kvl; k += 256) {                  // a thread per key: p_k and dS_k = p_k (dO.v_k - dO.o)
        const T* kr = Kb + (size_t)k * p.ld_qkv;
        const T* vr = Vb + (size_t)k * p.ld_qkv;
        float a = 0.f, dp = 0.f;
#pragma unroll
        for (int c = 0; c < DH / 8; ++c) {
            float t[8], u[8];
            load8f<T>(kr + 8 * c, t);
            load8f<T>(vr + 8 * c, u);
#pragma unroll
            for (int j = 0; j < 8; ++j) { a = fmaf(q[8 * c + j], t[j], a); dp = fmaf(g[8 * c + j], u[j], dp); }
        }
        const float pk = fast_exp2(a - lse);
        sP[k] = pk;
        sD[k] = uniform ? 0.f : pk * (dp - dl);
    }
    __syncthreads();
    // thread = (8-dim group dg, key lane kl): the gradient rows leave as 16-byte pieces; dq's partial sums meet in LDS
    const int dg = tid & 7, kl = tid >> 3;
    float qd[8], gd[8], zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    load8f<T>(qrow + 8 * dg, qd);                            // (re-read: indexing the register copies by dg would put them in scratch)
    load8f<T>(p.d_o + oi + 8 * dg, gd);
#pragma unroll
    for (int j = 0; j < 8; ++j) qd[j] *= p.scale;
    T* dQb = p.dq + row0 * p.ld_dqkv + hd * DH + 8 * dg;
    T* dKb = p.dk + row0 * p.ld_dqkv + hd * DH + 8 * dg;
    T* dVb = p.dv + row0 * p.ld_dqkv + hd * DH + 8 * dg;
    for (int k = kl; k < nrows; k += 32) {
        const size_t ro = (size_t)k * p.ld_dqkv;
        if (k < kvl) {
            const float ds = sD[k], pk = sP[k];
            float t[8], dk[8], dv[8];
            load8f<T>(Kb + (size_t)k * p.ld_qkv + 8 * dg, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[j] = fmaf(ds, t[j], acc[j]); dk[j] = ds * qd[j]; dv[j] = pk * gd[j]; }
            store8<T>(dKb + ro, dk);
            store8<T>(dVb + ro, dv);
        } else {
            store8<T>(dKb + ro, zero);
            store8<T>(dVb + ro, zero);
        }
        if (k != p.cls_tok) store8<T>(dQb + ro, zero);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[kl * DH + 8 * dg + j] = acc[j];
    __syncthreads();
    if (tid < DH) {
        float o = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) o += part[i * DH + tid];
        p.dq[(row0 + p.cls_tok) * p.ld_dqkv + hd * DH + tid] = from_f32<T>(o * p.scale);
    }
}

bool attn_cls_ok(int B, int N, int H, int cls_tok, int ld) {
    return B > 0 && N > 0 && N <= 8192 && H > 0 && H <= 4 && cls_tok >= 0 && cls_tok < N && ld >= H * DH && ld % 4 == 0;
}
}  // namespace

// o_cls, r1_cls [B, H * 64] (dtype), lse float[B, H]: attention output of query row `cls_tok` of every sample, and that output plus
// the residual row res[.., cls_tok] (encoder.py:27).  q / k / v [rows, ld_qkv], res [rows, ld_res]; kv_len (may be NULL), row_start
// (may be NULL: padded [B, N] layout) as in mtmp_attn_fwd_grouped.  Replaces attention.py:24-84 for the one query row that
// tri_mbt_vsltcls.py:248 reads of the last layer.
extern "C" int mtmp_attn_cls_fwd(int dtype, const void* q, const void* k, const void* v, int ld_qkv, const void* res, int ld_res,
                                 void* o_cls, void* r1_cls, float* lse, const int32_t* kv_len, const int32_t* row_start, int B, int N,
                                 int H, int cls_tok, float scale, void* stream) {
    MTMP_CHECK_ARG(q && k && v && res && o_cls && r1_cls && lse, "mtmp_attn_cls_fwd: null pointer");
    MTMP_CHECK_ARG(attn_cls_ok(B, N, H, cls_tok, ld_qkv) && ld_res >= H * DH && (!row_start || kv_len),
                   "mtmp_attn_cls_fwd: bad argument (B=%d N=%d H=%d cls=%d ld=%d)", B, N, H, cls_tok, ld_qkv);
    const size_t sm = ((size_t)N + 4 + 32 * DH) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        AttnClsArgs<float> a{(const float*)q, (const float*)k, (const float*)v, ld_qkv, (const float*)res, ld_res, (float*)o_cls,
                             (float*)r1_cls, lse, nullptr, nullptr, nullptr, nullptr, 0, kv_len, row_start, B, N, H, cls_tok, scale};
        if (int e = set_smem(attn_cls_fwd_kernel<float>, sm)) return e;
        hipLaunchKernelGGL(attn_cls_fwd_kernel<float>, dim3(B * H), dim3(256), sm, st, a);
    } else if (dtype == 1) {
        AttnClsArgs<bf16> a{(const bf16*)q, (const bf16*)k, (const bf16*)v, ld_qkv, (const bf16*)res, ld_res, (bf16*)o_cls,
                            (bf16*)r1_cls, lse, nullptr, nullptr, nullptr, nullptr, 0, kv_len, row_start, B, N, H, cls_tok, scale};
        if (int e = set_smem(attn_cls_fwd_kernel<bf16>, sm)) return e;
        hipLaunchKernelGGL(attn_cls_fwd_kernel<bf16>, dim3(B * H), dim3(256), sm, st, a);
    } else { mtmp_set_error("mtmp_attn_cls_fwd: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_attn_cls_fwd");
    return MTMP_OK;
}

// Backward of the above: d_o [B, H * 64] (gradient w.r.t. o_cls) -> dq, dk, dv written for EVERY row of every sample (dq is zero
// except in row cls_tok; rows past kv_len are zero): the dense [rows, ld_dqkv] buffers the layer's remaining backward
// (mtmp_gemm_tn, mtmp_gemm_lnbwd) reads.
extern "C" int mtmp_attn_cls_bwd(int dtype, const void* q, const void* k, const void* v, int ld_qkv, const void* o_cls, const void* d_o,
                                 const float* lse, const int32_t* kv_len, const int32_t* row_start, void* dq, void* dk, void* dv,
                                 int ld_dqkv, int B, int N, int H, int cls_tok, float scale, void* stream) {
    MTMP_CHECK_ARG(q && k && v && o_cls && d_o && lse && dq && dk && dv, "mtmp_attn_cls_bwd: null pointer");
    MTMP_CHECK_ARG(attn_cls_ok(B, N, H, cls_tok, ld_qkv) && ld_dqkv >= H * DH && (!row_start || kv_len),
                   "mtmp_attn_cls_bwd: bad argument (B=%d N=%d H=%d cls=%d ld=%d)", B, N, H, cls_tok, ld_qkv);
    const size_t sm = ((size_t)2 * N + 4 + 32 * DH) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        AttnClsArgs<float> a{(const float*)q, (const float*)k, (const float*)v, ld_qkv, nullptr, 0, (float*)o_cls, nullptr,
                             (float*)lse, (const float*)d_o, (float*)dq, (float*)dk, (float*)dv, ld_dqkv, kv_len, row_start, B, N, H,
                             cls_tok, scale};
        if (int e = set_smem(attn_cls_bwd_kernel<float>, sm)) return e;
        hipLaunchKernelGGL(attn_cls_bwd_kernel<float>, dim3(B * H), dim3(256), sm, st, a);
    } else if (dtype == 1) {
        AttnClsArgs<bf16> a{(const bf16*)q, (const bf16*)k, (const bf16*)v, ld_qkv, nullptr, 0, (bf16*)o_cls, nullptr, (float*)lse,
                            (const bf16*)d_o, (bf16*)dq, (bf16*)dk, (bf16*)dv, ld_dqkv, kv_len, row_start, B, N, H, cls_tok, scale};
        if (int e = set_smem(attn_cls_bwd_kernel<bf16>, sm)) return e;
        hipLaunchKernelGGL(attn_cls_bwd_kernel<bf16>, dim3(B * H), dim3(256), sm, st, a);
    } else { mtmp_set_error("mtmp_attn_cls_bwd: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_attn_cls_bwd");
    return MTMP_OK;
}

#ifdef MTMP_LAB_CLOCK
extern "C" int mtmp_dbg_read_stamps(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mtmp_dbg_stamps), (size_t)n * sizeof(long long));
}
#endif
extern "C" long long mtmp_key_norms_floats(long long rows, int H) { return ((rows + 31) / 32) * H; }

extern "C" int mtmp_key_norms(int dtype, const void* k, float* out, long long rows, int H, int ld, void* stream) {
    MTMP_CHECK_ARG(k && out, "mtmp_key_norms: null pointer");
    MTMP_CHECK_ARG(rows > 0 && rows < (1ll << 31) && H > 0 && H <= 4 && ld >= H * DH && (ld % 8) == 0,
                   "mtmp_key_norms: bad shape rows=%lld H=%d ld=%d", rows, H, ld);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)((rows + 31) / 32);
    if (dtype == 0) hipLaunchKernelGGL(key_norms_kernel<float>, dim3((nblk + 3) / 4), dim3(256), 0, st, (const float*)k, out, (int)rows, H, ld);
    else if (dtype == 1) hipLaunchKernelGGL(key_norms_kernel<bf16>, dim3((nblk + 3) / 4), dim3(256), 0, st, (const bf16*)k, out, (int)rows, H, ld);
    else { mtmp_set_error("mtmp_key_norms: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_key_norms");
    return MTMP_OK;
}

