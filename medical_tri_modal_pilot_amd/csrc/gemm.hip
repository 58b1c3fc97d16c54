// MFMA projection kernels of the encoder layer (SURVEY K5+K6, K8) for gfx950.
//
//  mtmp_ln_gemm : Y = act( LN(X) W^T + b ),  X [M,256]  -- the custom LayerNorm of
//      builder/models/src/transformer/module.py:138-144 (unbiased std, eps added to the
//      std) fused as the prologue of
//        * the Q/K/V projections, attention.py:60-62,68-70 (W = [Wq;Wk;Wv], N = 768), and
//        * the first position-wise FFN conv + ReLU, module.py:74-77 (N = 1024).
//      K = d_model = 256 is the whole row, so a wave keeps its 32 normalised rows as MFMA
//      fragments in registers (two lanes share a row: statistics need one cross-lane add)
//      and only the weight tiles go through LDS.  Also writes LN(X) (the backward's GEMM
//      operand) and the per-row (mean, 1/(std+eps)).
//  mtmp_gemm_nt : Y = act( A W^T + b ) (+ R) for any K % 64 == 0 -- second FFN conv with the
//      residual add of encoder.py:32 (K = 1024, N = 256); also dX = dY (W^T)^T.
//  mtmp_gemm_tn : dW[N,K] = dY[M,N]^T X[M,K] (+ column sums of dY = bias gradient): the
//      weight-gradient product, whose contraction index is the token index M (64 320 at
//      config 2) while the output is tiny -- split over M into partial slabs + one reduce pass.
//
// "NT" operands are contiguous along the contraction index, which is exactly the MFMA fragment
// shape (common.hip.h).  The product is computed as W-rows x token-columns so that a TOKEN is a
// lane: each lane then owns 4 consecutive output features per accumulator group and stores
// 8/16-byte pieces of its own row.  Weights arrive in the compute dtype (bf16 shadow copy or
// fp32 master), gamma/beta/bias always fp32.
#include "common.hip.h"
#include <type_traits>
template <int N> using template_int = std::integral_constant<int, N>;

namespace {

constexpr int BM = 128, BN = 128, BK = 64, LDW = BK + 8;

template <typename T> struct GemmArgs {
    const T* a; const T* w; const float* bias; const T* res; T* y;
    const float* gamma; const float* beta; T* xn; float* stats;
    int M, N, K, lda, ldy, ldr;
    float eps;
    float drop_p;        // nn.Dropout probability applied after act (0 = off), module.py:77-79
    unsigned seed;       // per-call seed of the counter-based mask (common.hip.h: dropout_keep4)
    const unsigned* seed_dev;   // optional device word XOR-ed into seed (advanced by the host/graph every step,
                                // so a replayed hipGraph does not repeat its masks)
    const T* gate;       // optional [M,N]: y = gate > 0 ? y * gate_scale : 0  (ReLU/dropout backward)
    float gate_scale;
    int act;             // 0 none, 1 ReLU, 2 exact GELU (Swin MLP, swin_transformer.py:439)
    const float* row_scale;   // optional per-sample factor (row-mode StochasticDepth): y *= row_scale[row / rows_per_scale]
    int rows_per_scale;
    unsigned short* signs = nullptr;   // row-panel kernels (bf16): 1 bit per output, "y > 0" -- written by the forward, read as the gate
    float* knorm = nullptr;            // Q/K/V projection (N = 768): [ceil(M / 32)][4] max ||k_h|| per 32-row block (attention.hip, AttnArgs::knorm)
    const int* m_live = nullptr;       // packed stream: device word holding the rows in use (<= M); common.hip.h live_rows
};

// 128 x 64 tile of a row-major matrix -> registers (4 x 16 B per thread).  The loads are
// unconditional from a CLAMPED (always valid) address and carry no use until tile_commit, so they
// stay in flight under the MFMAs of the current tile; columns >= kmax (K tail of a 64-wide chunk) are
// zeroed by a mask at commit time (a `cond ? load : 0` at the fetch site makes hipcc branch around --
// and wait for -- every single load).  Rows >= limit REPLICATE row limit-1: a replicated token row
// recomputes, and rewrites, exactly the output of row M-1, so the epilogue needs no row predicates;
// replicated weight rows produce columns >= N, which are never stored.
template <typename T> struct TileRegs { Frag<T> f[4]; unsigned ok; };

template <typename T, int ROWS = 128>
MTMP_DEV void tile_fetch(TileRegs<T>& t, const T* src, int ld, int row0, int limit, int k0, int tid, int kmax) {
    const int kc = k0 + (tid & 7) * 8;
    const int kcc = min(kc, kmax - 8);
    t.ok = 0;
#pragma unroll
    for (int ps = 0; ps < ROWS / 32; ++ps) {
        const int row = row0 + (tid >> 3) + 32 * ps;
        t.f[ps] = frag_load<T>(src + (size_t)min(row, limit - 1) * ld + kcc);
        t.ok |= (kc < kmax) ? (1u << ps) : 0u;
    }
}
template <typename T, int ROWS = 128> MTMP_DEV void tile_commit(T* dst, const TileRegs<T>& t, int tid) {
    if (wave_all(t.ok == (1u << (ROWS / 32)) - 1u)) {      // no K tail in this chunk (every chunk when K % 64 == 0): no masking
#pragma unroll
        for (int ps = 0; ps < ROWS / 32; ++ps) frag_store<T>(dst + ((tid >> 3) + 32 * ps) * LDW + (tid & 7) * 8, t.f[ps]);
        return;
    }
#pragma unroll
    for (int ps = 0; ps < ROWS / 32; ++ps)
        frag_store<T>(dst + ((tid >> 3) + 32 * ps) * LDW + (tid & 7) * 8, frag_keep(t.f[ps], (t.ok >> ps) & 1u));
}

// Epilogue of a 128 (tokens) x 128 (features) block, in two phases so that HBM sees whole rows:
//  1. every wave applies bias / activation / dropout to its accumulators (acc[nt]: rows = features
//     n0 + 32nt + acc_row(t, half), column = this lane's token) and parks them, as T, in an LDS
//     staging tile [128 tokens][LDO];
//  2. all 256 threads walk the tile row-wise -- 16 lanes x 16 bytes = one 256-byte row segment per
//     pass -- apply gate / row scale / residual from equally coalesced loads and store.
// (Storing straight from the accumulator layout writes 8-byte pieces at a row stride: partial-line
//  writes that made the K = 256, write-heavy projections run at half their HBM bound.)
constexpr int LDO = BN + 8;

// TM = token rows per workgroup: 128 (4 waves x 32 rows, 128 features each) or 64 (2 x 2 waves: 32 rows x 64
// features each) -- the smaller tile is used when a launch would otherwise have fewer workgroups than ~2 per CU
// (frozen-encoder stages 3-4, the image / text streams): such launches are bound by the latency chain of ONE
// workgroup per CU, and more, smaller workgroups overlap their chains.
template <int TM> struct NtGeom {
    // RT x NT 32x32 tiles per wave, WR x WC waves.  TM = 128: 4 x 1 waves of 32 tokens x 128 features (-DMTMP_NT_RT128=2:
    // 2 x 2 waves of 64 x 64, one operand fragment fewer per 4 MFMAs -- measured 5-10 % SLOWER on every shape of
    // tools/bench_kernels.py: 7 spilled VGPRs under the 128-register cap); TM = 64: 2 x 2 waves of 32 x 64.
    static constexpr int RT = 1;
    static constexpr int WR = TM / (32 * RT), WC = 4 / WR, NT = 4 / WC;
};

template <typename T, bool RELU, int TM, bool DROP>
MTMP_DEV void epilogue(const f32x16 (&acc)[NtGeom<TM>::RT][NtGeom<TM>::NT], const GemmArgs<T>& p, const int M, T* sOut, int m0, int n0, int tid) {
    using G = NtGeom<TM>;
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    const int wr = wave % G::WR, foff = (wave / G::WR) * 32 * G::NT;
    const unsigned thr = dropout_threshold(p.drop_p);
    const float keep_scale = 1.0f / (1.0f - p.drop_p);
    const unsigned seed_eff = p.seed ^ ((p.drop_p > 0.f && p.seed_dev) ? *p.seed_dev : 0u);
    // phase-2 operands first: independent 16-byte loads per thread stay in flight under phase 1
    // (loading them one by one inside the store loop cost one full memory latency per pass)
    constexpr int PS = TM / 16;
    const int c8 = (tid & 15) * 8, gcol = n0 + c8, gcolc = min(gcol, p.N - 8);
    Frag<T> gv[PS], rv[PS];
    if (p.gate) {
#pragma unroll
        for (int ps = 0; ps < PS; ++ps)
            gv[ps] = frag_load<T>(p.gate + (size_t)min(m0 + (tid >> 4) + 16 * ps, M - 1) * p.N + gcolc);
    }
    if (p.res) {
#pragma unroll
        for (int ps = 0; ps < PS; ++ps)
            rv[ps] = frag_load<T>(p.res + (size_t)min(m0 + (tid >> 4) + 16 * ps, M - 1) * p.ldr + gcolc);
    }
#pragma unroll
    for (int rt = 0; rt < G::RT; ++rt) {
        const int rl = 32 * (G::RT * wr + rt) + r, row = min(m0 + rl, M - 1);
#pragma unroll
        for (int nt = 0; nt < G::NT; ++nt) {
            if (n0 + foff + 32 * nt >= p.N) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = foff + 32 * nt + 8 * g + 4 * half, col = n0 + cl;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + col);
                float v[4];
                unsigned fld[4];
                if (DROP) dropout_fields4(seed_eff, ((unsigned)row * (unsigned)p.N + (unsigned)col) >> 2, fld);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[rt][nt][4 * g + i] + bv[i];
                if (RELU || p.act == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu1(v[i]);
                }
                if (p.act == 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = gelu<T>(v[i]);
                }
                if (DROP) {                      // (template parameter: no per-element selects when dropout is off)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fld[i] >= thr ? v[i] * keep_scale : 0.f;
                }
                store4<T>(sOut + rl * LDO + cl, v[0], v[1], v[2], v[3]);
            }
        }
    }
    __syncthreads();
    if (gcol < p.N) {
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const int rl = (tid >> 4) + 16 * ps, grow = min(m0 + rl, M - 1);
            Frag<T> o = frag_load<T>(sOut + rl * LDO + c8);
            if (p.gate || p.row_scale || p.res) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = to_f32(o.v[i]);
                if (p.gate) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = to_f32(gv[ps].v[i]) > 0.f ? v[i] * p.gate_scale : 0.f;
                }
                if (p.row_scale) {
                    const float rsv = p.row_scale[grow / p.rows_per_scale];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= rsv;
                }
                if (p.res) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = round_as<T>(v[i]) + to_f32(rv[ps].v[i]);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) o.v[i] = from_f32<T>(v[i]);
            }
            frag_store<T>(p.y + (size_t)grow * p.ldy + gcol, o);
        }
    }
}

// ---------------------------------------------------------------------------
// A-stationary row-panel kernel (K = 256).  A workgroup owns 128 token rows (32 per wave, held as 16
// normalised MFMA fragments in registers for the whole kernel) and walks over its share of the output
// features in PANELS of NP features: the [NP][256] weight panel is the only thing that goes through LDS.
//
// The loop is written for memory latency, which is what bounded the first version (its s_waitcnt vmcnt(0)
// at the top of every k-chunk also waited for the previous tile's output STORES -- gfx9 counts loads and
// stores in one in-order counter):
//   * two LDS panel buffers, one barrier per panel;
//   * the global loads of panel j+2 are issued BEFORE the stores of panel j, and are only waited for after
//     the MFMAs of panel j+1, so a wait for loads never covers younger stores;
//   * every load and store is unconditional: rows >= M read (and therefore recompute and rewrite) row M-1,
//     which keeps the in-flight counts static so that hipcc emits exact vmcnt(N) waits;
//   * the epilogue parks 32 tokens x 32 features per wave in a wave-private LDS tile (no barrier) and
//     writes 64-byte row pieces, 16 rows per store instruction.
// gridDim.y splits the panels of one row block over several workgroups when M alone cannot fill 256 CUs
// (image / text streams: 29 and 69 row blocks); split 0 writes xn / stats.
template <typename T> struct Panel {
    static constexpr int NP = sizeof(T) == 2 ? 64 : 32;          // features per panel (32 KiB of weights)
    static constexpr int G = NP / 32;                            // 32-feature MFMA groups per panel
    static constexpr int LDP = 256 + 16 / (int)sizeof(T);        // panel row + 16 B pad (conflict-free ds_read_b128)
    static constexpr int CPR = 256 * (int)sizeof(T) / 16;        // 16-byte chunks per weight row
    static constexpr int LOADS = NP * CPR / 256;                 // 16-byte loads per thread per panel (= 8)
    static constexpr int FS = 32 + 16 / (int)sizeof(T);          // staging row: 32 features + 16 B pad
    static constexpr int LPT = 32 * (int)sizeof(T) / 16;         // lanes per token row in the store phase
    static constexpr int PASSES = 32 * LPT / 64;                 // store instructions per 32 x 32 tile
    static constexpr size_t panel_bytes = (size_t)NP * LDP * sizeof(T);
    static constexpr size_t stage_bytes_wave = (size_t)32 * FS * sizeof(T);
    static constexpr size_t lds_bytes = 2 * panel_bytes + 4 * stage_bytes_wave;
};
struct PanelRegs { u32x4_t q[8]; };

template <typename T> MTMP_DEV void panel_fetch(PanelRegs& w, const T* wsrc, int n0, int N, int tid) {
    using P = Panel<T>;
#pragma unroll
    for (int i = 0; i < P::LOADS; ++i) {
        const int id = i * 256 + tid, row = id / P::CPR, ch = id % P::CPR;
        w.q[i] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(wsrc + (size_t)min(n0 + row, N - 1) * 256) + 16 * ch);
    }
}
template <typename T> MTMP_DEV void panel_commit(T* dst, const PanelRegs& w, int tid) {
    using P = Panel<T>;
#pragma unroll
    for (int i = 0; i < P::LOADS; ++i) {
        const int id = i * 256 + tid, row = id / P::CPR, ch = id % P::CPR;
        *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(dst + row * P::LDP) + 16 * ch) = w.q[i];
    }
}
// bias of one panel in accumulator order (lane half h: features 32g + 8*i4 + 4h .. +3), loaded straight into the
// register block that the panel's FIRST MFMA takes as its C operand (no accumulator-initialising moves).  A null bias
// reads the first N*4 bytes of W instead -- both are kernel-argument, i.e. global, pointers; a select between the bias and
// a __device__ constant would turn these into FLAT loads, whose out-of-order counters force s_waitcnt 0 everywhere -- and
// the caller zeroes the block behind a uniform branch (bias_or_zero), so the load count per iteration stays static.
template <typename T>
MTMP_DEV void bias_fetch(f32x16 (&b)[Panel<T>::G], const float* bias, const T* w, int n0, int half) {
    const float* src = bias ? bias : reinterpret_cast<const float*>(w);
#pragma unroll
    for (int g = 0; g < Panel<T>::G; ++g)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + n0 + 32 * g + 8 * i4 + 4 * half);
            b[g][4 * i4] = v[0]; b[g][4 * i4 + 1] = v[1]; b[g][4 * i4 + 2] = v[2]; b[g][4 * i4 + 3] = v[3];
        }
}
template <typename T> MTMP_DEV void bias_or_zero(f32x16 (&b)[Panel<T>::G], const float* bias) {
    if (!bias) {
#pragma unroll
        for (int g = 0; g < Panel<T>::G; ++g) b[g] = f32x16{0};
    }
}
// LDS traffic of ONE wave to its private staging tile needs no barrier (a wave's LDS instructions execute in
// order); this only stops the compiler from moving memory operations across the hand-over point.
MTMP_DEV void wave_lds_handover() { asm volatile("" ::: "memory"); }

template <typename T, bool RELU, bool DROP>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void ln_gemm_kernel(GemmArgs<T> p) {
    using P = Panel<T>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sP = reinterpret_cast<T*>(smem_raw);                                       // [2][NP][LDP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    T* sS = reinterpret_cast<T*>(smem_raw + 2 * P::panel_bytes) + wave * 32 * P::FS;   // wave-private [32][FS]
    float* sG = reinterpret_cast<float*>(smem_raw + 2 * P::panel_bytes);          // gamma|beta, prologue only (aliases sS)
    const int npanels = p.N / P::NP;
    const int per = (npanels + gridDim.y - 1) / gridDim.y;
    const int j0 = blockIdx.y * per, j1 = min(npanels, j0 + per);
    if (j0 >= j1) return;
    const int m_wave = blockIdx.x * BM + wave * 32;
    const int row = min(m_wave + r, p.M - 1);
    PanelRegs wreg;
    panel_fetch<T>(wreg, p.w, j0 * P::NP, p.N, tid);
    if (p.gamma) {
        sG[tid] = p.gamma[tid];
        sG[256 + tid] = p.beta[tid];
    }
    // ---- LayerNorm prologue, in registers: lane (r, half) holds k = 16c + 8*half + j of row r
    Frag<T> af[16];
    float s1 = 0.f;
    const T* arow = p.a + (size_t)row * p.lda + 8 * half;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        af[c] = frag_load<T>(arow + 16 * c);
#pragma unroll
        for (int j = 0; j < 8; ++j) s1 += to_f32(af[c].v[j]);
    }
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * (1.0f / 256.0f);
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = to_f32(af[c].v[j]) - mean; s2 = fmaf(d, d, s2); }
    s2 += __shfl_xor(s2, 32, 64);
    const float sigma = sqrtf(s2 * (1.0f / 255.0f));             // torch.std: Bessel-corrected
    const float rs = 1.0f / (sigma + p.eps);
    __syncthreads();                                             // sG ready
    if (p.gamma) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int k = 16 * c + 8 * half;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                af[c].v[j] = from_f32<T>(fmaf(sG[k + j], (to_f32(af[c].v[j]) - mean) * rs, sG[256 + k + j]));
            if (p.xn && blockIdx.y == 0) frag_store<T>(p.xn + (size_t)row * 256 + k, af[c]);
        }
        if (p.stats && half == 0 && blockIdx.y == 0) {
            p.stats[2 * (size_t)row] = mean;
            p.stats[2 * (size_t)row + 1] = rs;
        }
    }
    panel_commit<T>(sP, wreg, tid);
    f32x16 binit[P::G];
    bias_fetch<T>(binit, p.bias, p.w, j0 * P::NP, half);
    panel_fetch<T>(wreg, p.w, min(j0 + 1, j1 - 1) * P::NP, p.N, tid);
    __syncthreads();                                             // panel j0 visible; sG dead (sS may be written)
    const unsigned thr = dropout_threshold(p.drop_p);
    const float keep_scale = 1.0f / (1.0f - p.drop_p);
    const unsigned seed_eff = p.seed ^ ((p.drop_p > 0.f && p.seed_dev) ? *p.seed_dev : 0u);
    const int tok = lane / P::LPT, ch = lane % P::LPT;           // store phase: this lane's token (+ 64/LPT per pass), 16 B chunk
    for (int j = j0; j < j1; ++j) {
        const int n0 = j * P::NP;
        T* cur = sP + ((j - j0) & 1) * P::NP * P::LDP;
        T* nxt = sP + (((j - j0) & 1) ^ 1) * P::NP * P::LDP;
        f32x16 acc[P::G];                                        // the first MFMA takes the bias block as its C operand
        bias_or_zero<T>(binit, p.bias);
#pragma unroll
        for (int g = 0; g < P::G; ++g)
            acc[g] = mma_c<T>(frag_load<T>(cur + (32 * g + r) * P::LDP + 8 * half), af[0], binit[g]);
        constexpr int CEND = 16;
#pragma unroll
        for (int c = 1; c < CEND; ++c)
#pragma unroll
            for (int g = 0; g < P::G; ++g)
                mma<T>(acc[g], frag_load<T>(cur + (32 * g + r) * P::LDP + 16 * c + 8 * half), af[c]);
        panel_commit<T>(nxt, wreg, tid);                         // panel j+1 (or a harmless repeat of the last one)
        bias_fetch<T>(binit, p.bias, p.w, min(j + 1, j1 - 1) * P::NP, half);
        panel_fetch<T>(wreg, p.w, min(j + 2, j1 - 1) * P::NP, p.N, tid);
#pragma unroll
        for (int g = 0; g < P::G; ++g) {
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int col = n0 + 32 * g + 8 * i4 + 4 * half;
                unsigned fld[4];
                if (DROP) dropout_fields4(seed_eff, ((unsigned)row * (unsigned)p.N + (unsigned)col) >> 2, fld);
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[g][4 * i4 + i];
                    if (RELU) v[i] = relu1(v[i]);
                    if (DROP) v[i] = fld[i] >= thr ? v[i] * keep_scale : 0.f;
                }
                store4<T>(sS + r * P::FS + 8 * i4 + 4 * half, v[0], v[1], v[2], v[3]);
            }
            wave_lds_handover();
#pragma unroll
            for (int ps = 0; ps < P::PASSES; ++ps) {
                const int t = tok + ps * (64 / P::LPT);
                const u32x4_t d = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(sS + t * P::FS) + 16 * ch);
                T* dst = p.y + (size_t)min(m_wave + t, p.M - 1) * p.ldy + n0 + 32 * g;
                *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(dst) + 16 * ch) = d;
            }
            wave_lds_handover();
        }
        __syncthreads();                                         // nxt visible to all waves, cur free for the next commit
    }
}

// ---------------------------------------------------------------------------
// bf16 build of the same row-panel kernel with the weight panels moved by LDS-DMA (global_load_lds_dwordx4: LDS[M0 + 16 * lane]
// <- 16 bytes from the lane's own global address, tools/dbg/dma_probe).  What that buys over ln_gemm_kernel<bf16>:
//   * no panel prefetch registers (32 VGPRs) and no ds_write_b128 commit (13 LDS-path cycles each, 8 per thread per panel);
//   * the bias block lives in LDS (a broadcast ds_read_b128 into the accumulators at the top of a panel) instead of 32 more
//     registers that stayed live across the epilogue -- the first version ran at 256 VGPRs with spills and its MFMA phase began
//     with ten "ds_read_b128 -> s_waitcnt lgkmcnt(0) -> v_mfma" round trips because nothing was left to prefetch into.
// A DMA instruction writes 1 KiB of CONTIGUOUS LDS, i.e. two unpadded 512-byte weight rows, so bank conflicts are avoided by
// permuting the 16-byte chunks inside a row instead of padding it: chunk c of panel row R sits at position c ^ (R & 15) (the
// lanes of one ds_read_b128 service group hold 16 rows that are distinct mod 16).  The permutation is applied on the SOURCE
// side (each lane picks the global chunk that belongs at its fixed LDS slot), the reader XORs its chunk index.
// A panel is two accumulator groups of 32 features; the loop is rotated so that the epilogue of one group (vector work + the
// staging round trip) is issued between the MFMAs of the other: [MFMA g1(j) | epilogue g0(j)] -> s_waitcnt vmcnt(4) (DMA(j+1)
// is older than exactly the four row-piece stores of the last two epilogues) -> barrier -> DMA(j+2) -> [MFMA g0(j+1) |
// epilogue g1(j)].
// K = row length (contraction index) = 256, the fusion layers' d_model.
template <int K = 256> struct PanelDmaK {
    static constexpr int NP = 64, G = 2, FS = 40, LPT = 4, PASSES = 2, MAXP = 16;   // MAXP panels (1024 features) per workgroup
    static constexpr int KC = K / 16, ROWB = 2 * K;                // 16-wide k-steps; bytes per weight row
    static constexpr int ND = NP * ROWB / 1024 / 4;               // 1 KiB DMA pieces per wave and panel
    static constexpr unsigned panel_bytes = NP * ROWB, group_bytes = 32 * ROWB, stage_off = 2 * panel_bytes,
                              bias_off = stage_off + 4 * 32 * FS * 2;
    static constexpr size_t lds_bytes = bias_off + MAXP * NP * 4;
    static_assert(K % 128 == 0 && (NP * ROWB) % 4096 == 0, "row-panel kernel: K must be a multiple of 128");
};
using PanelDma = PanelDmaK<256>;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // (m0 is "reserved"; naming it as clobbered is exactly the point)
MTMP_DEV void dma16(unsigned voff, const void* sbase, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_off) : "memory", "m0");
}
#pragma clang diagnostic pop
// 2-byte global load the compiler does not count (the loop's vmcnt budget is kept by hand, see the phase list below); the
// matching wait names the destination register as read-write so that no use can be scheduled above it.
MTMP_DEV void gload_u16(unsigned& d, const void* ptr) { asm volatile("global_load_ushort %0, %1, off" : "=v"(d) : "v"(ptr) : "memory"); }
template <int N> MTMP_DEV void gwait1(unsigned& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }

// SIGNS (forward, with RELU): besides y the kernel writes one bit per output, "y > 0", 16 bits per lane and 32-feature group in
// the lane's own accumulator order (rows through swz23): signs[((2 j + g) M + row) 2 + half] for panel j, group g.  GATE (backward): the FFN's
// dH = dY W2 gated by the saved hidden activation (y = h > 0 ? y * gate_scale : 0, autograd of module.py:77-79) reads those
// bits instead of h itself -- the same kernel geometry, so a lane needs exactly the 16 bits its forward twin wrote: 4 MB of
// gate traffic per launch at config 2 instead of 132 MB.  No LayerNorm in GATE mode (p.gamma == nullptr).
// KNORM (the Q/K/V projection, N = 768 = [Wq; Wk; Wv], head h of K = panel 4 + h): the epilogue also sums the squares of a row's 64
// key features (a lane holds 32 of them, its half-wave partner the rest) and keeps the maximum over the wave's 32 rows -- the
// key-norm table that lets the attention forward drop its running maximum (attention.hip, bounded body) at no extra pass over K.
// both accumulator groups of key panel je went through the epilogue: fold the lane's square sum into the head's maximum over the
// wave's 32 rows (a free function on references: as a lambda captured by the phase lambdas it kept its variables in scratch)
MTMP_DEV void knorm_fold(int je, float& ksq, float& kh0, float& kh1, float& kh2, float& kh3) {
    if (je >= 4 && je < 8) {
        const float mx = wave_max(half_sum(ksq));
        kh0 = je == 4 ? mx : kh0; kh1 = je == 5 ? mx : kh1; kh2 = je == 6 ? mx : kh2; kh3 = je == 7 ? mx : kh3;   // (selects: branches kept them in scratch)
    }
    ksq = 0.f;
}
template <bool RELU, bool DROP, bool GATE, bool SIGNS, bool KNORM = false>
__global__ __launch_bounds__(256, 2) void ln_gemm_dma_kernel(Grouped<GemmArgs<bf16>> grp) {
    using T = bf16;
    constexpr int K = 256;
    const int seg = grp_find(grp, (int)blockIdx.x);          // row blocks of up to three token streams in one grid (common.hip.h)
    const GemmArgs<bf16>& p = grp.seg[seg];
    const int bx = (int)blockIdx.x - grp.first[seg];
    using P = PanelDma;
    constexpr int KC = P::KC;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, half = lane >> 5;
    T* sS = reinterpret_cast<T*>(smem_raw + P::stage_off) + wave * 32 * P::FS;      // wave-private [32][FS]
    float* sG = reinterpret_cast<float*>(smem_raw + P::stage_off);                  // gamma|beta, prologue only (aliases sS)
    float* sB = reinterpret_cast<float*>(smem_raw + P::bias_off);                   // bias of this workgroup's panels
    const int npanels = p.N / P::NP;
    const int M = live_rows(p.M, p.m_live);
    const int per = (npanels + gridDim.y - 1) / gridDim.y;
    const int j0 = blockIdx.y * per, j1 = min(npanels, j0 + per);
    if (j0 >= j1 || bx * BM >= M) return;     // (packed stream: row blocks past the rows in use)
    const int m_wave = bx * BM + wave * 32;
    const int row = min(m_wave + r, M - 1);
    // DMA slot of this lane: piece i of wave w fills LDS bytes [1024 (ND w + i), +1024) of the panel image, i.e. 16-byte slot
    // q = 64 (ND w + i) + lane = position q % (K/8) of panel row q / (K/8), which takes global chunk position ^ (row & 15)
    // (K = 256: rows 16w + 2i + (lane >> 5), position lane & 31)
    unsigned dsrc[P::ND];
#pragma unroll
    for (int i = 0; i < P::ND; ++i) {
        const unsigned q = 64u * (unsigned)(P::ND * wave + i) + (unsigned)lane, prow = q / (K / 8), pos = q % (K / 8);
        dsrc[i] = prow * (unsigned)P::ROWB + 16u * (pos ^ (prow & 15u));
    }
    auto panel_dma = [&](int j, int buf) __attribute__((always_inline)) {
        const char* src = reinterpret_cast<const char*>(p.w + (size_t)j * P::NP * K);
#pragma unroll
        for (int i = 0; i < P::ND; ++i) dma16(dsrc[i], src, lds0 + buf * P::panel_bytes + (unsigned)(P::ND * wave + i) * 1024u);
    };
    panel_dma(j0, 0);
    if (p.gamma) {
        for (int i = tid; i < K; i += 256) {
            sG[i] = p.gamma[i];
            sG[K + i] = p.beta[i];
        }
    }
    for (int i = tid; i < (j1 - j0) * P::NP; i += 256) sB[i] = p.bias ? p.bias[j0 * P::NP + i] : 0.f;
    // ---- LayerNorm prologue, in registers: lane (r, half) holds k = 16c + 8*half + j of row r
    Frag<T> af[KC];
    const T* arow = p.a + (size_t)row * p.lda + 8 * half;
#pragma unroll
    for (int c = 0; c < KC; ++c) af[c] = frag_load<T>(arow + 16 * c);
    if constexpr (GATE) {
        // drop_p > 0 in GATE mode: the A operand is the gradient of a dropout's output (the FFN's drop2 in front of dH = dY W2):
        // a = keep ? a / (1 - p) : 0 with mtmp_dropout_bwd's mask (element row * K + col of a contiguous [M, K] tensor), applied
        // to the fragments in registers; p.xn, if given, receives the masked rows (the weight-gradient product's operand) -- the
        // separate dropout-backward launch and one read of dY are gone.
        if (p.drop_p > 0.f) {
            const unsigned thr_a = dropout_threshold(p.drop_p), seed_a = p.seed ^ (p.seed_dev ? *p.seed_dev : 0u);
            const float sc_a = 1.0f / (1.0f - p.drop_p);
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const unsigned g0 = (unsigned)row * (K / 4) + 4 * c + 2 * half;
                const unsigned k0 = dropout_keep4(seed_a, g0, thr_a), k1 = dropout_keep4(seed_a, g0 + 1, thr_a);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool keep = ((j < 4 ? k0 >> j : k1 >> (j - 4)) & 1u) != 0;
                    af[c].v[j] = keep ? from_f32<T>(to_f32(af[c].v[j]) * sc_a) : (T)0.0f;
                }
                if (p.xn && blockIdx.y == 0) frag_store<T>(p.xn + (size_t)row * K + 16 * c + 8 * half, af[c]);
            }
        }
    }
    if constexpr (!GATE) {
        float s1 = 0.f;
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) s1 += to_f32(af[c].v[j]);
        s1 += __shfl_xor(s1, 32, 64);
        const float mean = s1 * (1.0f / K);
        float s2 = 0.f;
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = to_f32(af[c].v[j]) - mean; s2 = fmaf(d, d, s2); }
        s2 += __shfl_xor(s2, 32, 64);
        // the fusion layers' LayerNorm: torch.std (Bessel-corrected), eps added to it; nn.LayerNorm: biased variance, eps inside
        const float rs = 1.0f / (sqrtf(s2 * (1.0f / (K - 1))) + p.eps);
        __syncthreads();                                         // sG ready
        if (p.gamma) {
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int k = 16 * c + 8 * half;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    af[c].v[j] = from_f32<T>(fmaf(sG[k + j], (to_f32(af[c].v[j]) - mean) * rs, sG[K + k + j]));
                if (p.xn && blockIdx.y == 0) frag_store<T>(p.xn + (size_t)row * K + k, af[c]);
            }
            if (p.stats && half == 0 && blockIdx.y == 0) {
                p.stats[2 * (size_t)row] = mean;
                p.stats[2 * (size_t)row + 1] = rs;
            }
        }
    }
    const unsigned thr = dropout_threshold(p.drop_p);
    const float keep_scale = 1.0f / (1.0f - p.drop_p);
    const unsigned seed_eff = p.seed ^ ((p.drop_p > 0.f && p.seed_dev) ? *p.seed_dev : 0u);
    const int tok = lane / P::LPT, ch = lane % P::LPT;           // store phase: this lane's token (+ 16 per pass), 16 B chunk
    // reader: row 32g + r, k-step c -> chunk (2c + half) ^ (r & 15); the XOR only touches chunk bits 0-3, so c >> 3 and g are
    // immediate offsets on eight per-lane addresses
    // (rows through swz23, common.hip.h: accumulator registers 8s..8s+7 of a lane then hold 8 CONSECUTIVE features, 16s + 8 half + j
    //  -- 16-byte pieces for the staging tile, and exactly the k-step-s operand fragment of a product that consumes this one)
    const char* rd[8];
    const int rs = swz23(r);
#pragma unroll
    for (int c = 0; c < 8; ++c) rd[c] = smem_raw + rs * P::ROWB + 16 * ((2 * c + half) ^ (rs & 15));
    auto ucol = [&](int i4) __attribute__((always_inline)) { return 16 * (i4 >> 1) + 8 * half + 4 * (i4 & 1); };   // first feature of registers 4 i4 .. 4 i4 + 3
    panel_dma(min(j0 + 1, j1 - 1), 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // panels j0, j0 + 1 landed (this wave's share)
    __syncthreads();                                             // ... everyone's; sB written; sG dead (sS may be written)
    f32x16 acc[P::G];
    // One phase = the 16 MFMAs of accumulator group gm of panel jm, issued in eight slices of two; between them (EPI) the
    // epilogue of group ge of panel je, cut into eight pieces of vector work: per 4 features [mask hash] and [ReLU / select /
    // round / park in the staging tile].  sched_barrier(0) between slices keeps hipcc from regrouping them (left alone it
    // issues the MFMAs back to back, and a wave cannot issue past an MFMA that waits for the matrix pipe); inside a slice the
    // order is MFMA, half of the piece, MFMA, the other half.  Weight fragments are read eight MFMAs ahead.
    // The staging tile of an epilogue is drained (two ds_read_b128 + two row-piece stores) in the first slices of the NEXT phase,
    // so that LDS round trip also runs under MFMAs; a wave's LDS instructions execute in order, so the next epilogue's first
    // ds_write (slice 1) cannot overtake those reads.
    constexpr int NG = GATE ? 1 : 0, NS = SIGNS ? 1 : 0;         // gate loads / sign stores per tile
    auto tile_ptr = [&](const T* base, int ld, int jp, int gp, int ps) __attribute__((always_inline)) {
        const int t = tok + ps * (64 / P::LPT);
        return reinterpret_cast<const char*>(base + (size_t)min(m_wave + t, M - 1) * ld + jp * P::NP + 32 * gp) + 16 * ch;
    };
    auto signs_ptr = [&](int j, int g) __attribute__((always_inline)) { return p.signs + ((size_t)(2 * j + g) * p.M + row) * 2 + half; };
    auto drain_load = [&](u32x4_t (&d)[P::PASSES]) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < P::PASSES; ++ps)
            d[ps] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(sS + (tok + ps * (64 / P::LPT)) * P::FS) + 16 * ch);
    };
    auto drain_store = [&](const u32x4_t (&d)[P::PASSES], int jp, int gp) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < P::PASSES; ++ps) *reinterpret_cast<u32x4_t*>(const_cast<char*>(tile_ptr(p.y, p.ldy, jp, gp, ps))) = d[ps];
    };
    // one 4-feature piece of an epilogue: activation / dropout / gate, sign bits, rounding, parked in the staging tile
    float ksq = 0.f, kh0 = 0.f, kh1 = 0.f, kh2 = 0.f, kh3 = 0.f;      // KNORM: running square sum of the key panel in its epilogue; finished heads
    float kmask = 0.f;                                               // 1 while the epilogue's panel is a key head (wave-uniform)
    auto piece = [&](int g, int i4, const unsigned (&fld)[4], unsigned gate_bits, unsigned& sign_bits) __attribute__((always_inline)) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = acc[g][4 * i4 + i];
            if (KNORM) ksq = fmaf(v[i] * kmask, v[i], ksq);
            if (SIGNS) {       // "positive and kept" as ONE lane mask: it selects the value and is shifted into the collection
                unsigned long long lm = __builtin_amdgcn_ballot_w64(v[i] > 0.f);
                if (DROP) lm &= __builtin_amdgcn_ballot_w64(fld[i] >= thr);
                const float sv = DROP ? v[i] * keep_scale : v[i];
                // v = lm ? sv : 0;  sign_bits = 2 * sign_bits + lm  (one add-with-carry whose carry-in is the lane mask; written in
                // C++ hipcc rebuilds the mask from a 0/1 register and spells the update with two selects, a shift and an or).
                // s_nop: a compare result read as a mask by the very next vector instruction is a hazard nobody checks inside asm.
                asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, %2, %3\n\tv_addc_co_u32_e64 %1, vcc, %1, %1, %3"
                    : "=&v"(v[i]), "+v"(sign_bits) : "v"(sv), "s"(lm) : "vcc");
            } else {
                if (RELU) v[i] = relu1(v[i]);
                if (DROP) v[i] = fld[i] >= thr ? v[i] * keep_scale : 0.f;
            }
            if (GATE)          // bit 15 - (4 i4 + i) of the lane's 16: all ones or zero, and-ed onto the scaled value
                v[i] = __builtin_bit_cast(float, __builtin_bit_cast(int, v[i] * p.gate_scale) &
                                                     __builtin_amdgcn_sbfe((int)gate_bits, 15 - (4 * i4 + i), 1));
        }
        store4<T>(sS + r * P::FS + ucol(i4), v[0], v[1], v[2], v[3]);
    };
    // One phase = the 16 MFMAs of accumulator group gm of panel jm, issued in eight slices of two; between them (EPI) the
    // epilogue of group ge of panel je, cut into eight pieces of vector work: per 4 features [mask hash] and [ReLU / select /
    // round / park in the staging tile], and (PEND) the drain of the previous phase's tile.  sched_barrier(0) between slices
    // keeps hipcc from regrouping them (left alone it issues the MFMAs back to back, and a wave cannot issue past an MFMA that
    // waits for the matrix pipe); inside a slice the order is MFMA, half of the piece, MFMA, the other half.  Weight fragments
    // are read eight MFMAs ahead.  gnext / gcur: gate bits of the group being multiplied (loaded here) and of the epilogue's
    // group (loaded one phase ago; GW = vector-memory operations issued since, i.e. the vmcnt that proves them landed).
    auto phase = [&](int jm, int gm, int je, int ge, auto epi_tag, int jp, int gp, auto pend_tag, unsigned& gnext, unsigned& gcur,
                     auto gw_tag) __attribute__((always_inline)) {
        constexpr bool EPI = decltype(epi_tag)::value, PEND = decltype(pend_tag)::value;
        constexpr int GW = decltype(gw_tag)::value;
        const size_t cur = (size_t)(((jm - j0) & 1) * P::panel_bytes + gm * P::group_bytes);
        const int n0 = je * P::NP;
        if (KNORM) kmask = (EPI && je >= 4 && je < 8) ? 1.f : 0.f;
        if (GATE) gload_u16(gnext, signs_ptr(jm, gm));
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sB + (jm - j0) * P::NP + 32 * gm + ucol(i4));
            acc[gm][4 * i4] = v[0]; acc[gm][4 * i4 + 1] = v[1]; acc[gm][4 * i4 + 2] = v[2]; acc[gm][4 * i4 + 3] = v[3];
        }
        Frag<T> b[KC];
#pragma unroll
        for (int c = 0; c < 8; ++c) b[c].v = *reinterpret_cast<const bf16x8*>(rd[c] + cur);
        u32x4_t dr[P::PASSES];
        if (PEND) {
            wave_lds_handover();
            drain_load(dr);
            wave_lds_handover();
        }
        if (GATE && EPI) gwait1<GW>(gcur);
        unsigned fld[4], sign_bits = 0;
#pragma unroll
        for (int s8 = 0; s8 < KC / 2; ++s8) {                     // (K = 256: eight slices, all of them carry epilogue work)
            __builtin_amdgcn_sched_barrier(0);
            if (2 * s8 + 8 < KC) {
                b[8 + 2 * s8].v = *reinterpret_cast<const bf16x8*>(rd[(2 * s8) & 7] + cur + ((8 + 2 * s8) >> 3) * 256);
                b[9 + 2 * s8].v = *reinterpret_cast<const bf16x8*>(rd[(2 * s8 + 1) & 7] + cur + ((9 + 2 * s8) >> 3) * 256);
            }
            mma<T>(acc[gm], b[2 * s8], af[2 * s8]);
            mma<T>(acc[gm], b[2 * s8 + 1], af[2 * s8 + 1]);
            if (PEND && s8 == 0) drain_store(dr, jp, gp);
            if (EPI && s8 < 8) {
                const int i4 = s8 >> 1;
                if ((s8 & 1) == 0) {
                    const int col = n0 + 32 * ge + ucol(i4);
                    if (DROP) dropout_fields4(seed_eff, ((unsigned)row * (unsigned)p.N + (unsigned)col) >> 2, fld);
                } else {
                    piece(ge, i4, fld, gcur, sign_bits);
                }
                constexpr int NV = DROP ? 9 : (KNORM ? 5 : 3);
                if (2 * s8 + 8 < KC) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NV + 2, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (SIGNS && EPI) *signs_ptr(je, ge) = (unsigned short)sign_bits;
        if (KNORM && EPI && ge == 1) knorm_fold(je, ksq, kh0, kh1, kh2, kh3);
    };
    // the last group's epilogue has no MFMAs left to hide under
    auto epi_alone = [&](int j, int g, int jp, int gp, unsigned& gcur) __attribute__((always_inline)) {
        u32x4_t dr[P::PASSES];
        wave_lds_handover();
        drain_load(dr);
        wave_lds_handover();
        drain_store(dr, jp, gp);
        if (GATE) gwait1<0>(gcur);
        const int n0 = j * P::NP;
        unsigned sign_bits = 0;
        if (KNORM) kmask = (j >= 4 && j < 8) ? 1.f : 0.f;
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
            const int col = n0 + 32 * g + ucol(i4);
            unsigned fld[4];
            if (DROP) dropout_fields4(seed_eff, ((unsigned)row * (unsigned)p.N + (unsigned)col) >> 2, fld);
            piece(g, i4, fld, gcur, sign_bits);
        }
        if (KNORM) knorm_fold(j, ksq, kh0, kh1, kh2, kh3);
        if (SIGNS) *signs_ptr(j, g) = (unsigned short)sign_bits;
        wave_lds_handover();
        drain_load(dr);
        drain_store(dr, j, g);
    };
    using Yes = std::true_type;
    using No = std::false_type;
    // Phases of panel j: A(j) = [MFMA g0(j) | epilogue g1(j-1) | drain g0(j-1)], B(j) = [MFMA g1(j) | epilogue g0(j) | drain g1(j-1)].
    // Vector-memory operations in issue order (G = the gate load of the group being multiplied, S = the PASSES row-piece stores
    // of a drain + the sign store of an epilogue):
    //   ... B(j): G S | barrier | DMA(j+2) x8 | A(j+1): G S | B(j+1): G S | wait DMA | barrier ...
    // so DMA(j+2) is older than 2 (G + S) operations when it is waited for; the gate bits an A phase's epilogue uses (loaded at
    // the top of the B phase before) are older than S + 8 + G, a B phase's than S + G.
    constexpr int NST = P::PASSES + NS, WDMA = 2 * (NG + NST), WA = NST + 8 + NG, WB = NST + NG;
    unsigned g0bits = 0, g1bits = 0;                             // gate bits of the g = 0 / g = 1 group in flight
    phase(j0, 0, 0, 0, No{}, 0, 0, No{}, g0bits, g1bits, template_int<0>{});
    if (j0 < j1 - 1) {
        phase(j0, 1, j0, 0, Yes{}, 0, 0, No{}, g1bits, g0bits, template_int<NG>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        panel_dma(min(j0 + 2, j1 - 1), 0);
        phase(j0 + 1, 0, j0, 1, Yes{}, j0, 0, Yes{}, g0bits, g1bits, template_int<8 + NG>{});   // (nothing was drained yet)
        for (int j = j0 + 1; j < j1 - 1; ++j) {
            phase(j, 1, j, 0, Yes{}, j - 1, 1, Yes{}, g1bits, g0bits, template_int<WB>{});
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WDMA) : "memory");
            __syncthreads();                                     // panel j+1 visible to all waves, panel j free
            panel_dma(min(j + 2, j1 - 1), (j - j0) & 1);         // (a harmless repeat at the end keeps vmcnt static)
            phase(j + 1, 0, j, 1, Yes{}, j, 0, Yes{}, g0bits, g1bits, template_int<WA>{});
        }
        phase(j1 - 1, 1, j1 - 1, 0, Yes{}, j1 - 2, 1, Yes{}, g1bits, g0bits, template_int<WB>{});
    } else {
        phase(j0, 1, j0, 0, Yes{}, 0, 0, No{}, g1bits, g0bits, template_int<NG>{});
    }
    epi_alone(j1 - 1, 1, j1 - 1, 0, g1bits);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the repeated last DMA must not outlive the workgroup's LDS)
    if (KNORM && p.knorm && lane == 0 && m_wave < M) {          // heads whose panel this workgroup walked (gridDim.y may split them)
        float* kn = p.knorm + (size_t)(m_wave >> 5) * 4;
        if (j0 <= 4 && 4 < j1) kn[0] = sqrtf(kh0);
        if (j0 <= 5 && 5 < j1) kn[1] = sqrtf(kh1);
        if (j0 <= 6 && 6 < j1) kn[2] = sqrtf(kh2);
        if (j0 <= 7 && 7 < j1) kn[3] = sqrtf(kh3);
    }
}

// ---------------------------------------------------------------------------
// One LDS stage + one register stage (tile k+1 is fetched while tile k is multiplied).  A version with two stages of
// each (as in gemm_tn_tr_kernel) was measured and rejected: it needs 256 VGPRs and 74 KiB of LDS, i.e. two
// workgroups per CU instead of three, and every large launch got 20-30 % slower (dH 93 -> 120 us, FFN2 76 -> 97 us);
// only launches with < 1 workgroup per CU gained.  Occupancy hides this loop's latency better than depth.
template <typename T, bool RELU, int TM, bool DROP>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? (TM == 64 ? 3 : 4) : 1)) void gemm_nt_kernel(Grouped<GemmArgs<T>> grp) {
    using G = NtGeom<TM>;
    int wg = xcd_remap(blockIdx.x, gridDim.x);
    if (grp.first[1] >= (int)gridDim.x && grp.seg[0].m_live) {
        // one packed stream alone in the grid: the XCD chunks are cut over the tiles of the LIVE rows -- cut over the whole grid,
        // the live tiles (the front of the work list) would all land on the first XCDs (half the rows in use: half the chip idle)
        const int live = ((live_rows(grp.seg[0].M, grp.seg[0].m_live) + TM - 1) / TM) * ((grp.seg[0].N + BN - 1) / BN);
        if ((int)blockIdx.x >= live) return;
        wg = xcd_remap(blockIdx.x, live);
    }
    const int seg = grp_find(grp, wg);
    const GemmArgs<T>& p = grp.seg[seg];
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sA = reinterpret_cast<T*>(smem_raw);   // [TM][LDW]
    T* sW = sA + TM * LDW;                    // [BN][LDW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    const int wr = wave % G::WR, foff = (wave / G::WR) * 32 * G::NT;
    const int ntn = (p.N + BN - 1) / BN;
    const int w = wg - grp.first[seg];
    const int m0 = (w / ntn) * TM, n0 = (w % ntn) * BN;
    const int nk = (p.K + BK - 1) / BK;
    const int M = live_rows(p.M, p.m_live);
    if (m0 >= M) return;                      // packed stream: row blocks past the rows in use (workgroup-uniform)
    TileRegs<T> areg, wreg;
    tile_fetch<T, TM>(areg, p.a, p.lda, m0, M, 0, tid, p.K);
    tile_fetch<T>(wreg, p.w, p.K, n0, p.N, 0, tid, p.K);
    f32x16 acc[G::RT][G::NT];
#pragma unroll
    for (int rt = 0; rt < G::RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < G::NT; ++nt) acc[rt][nt] = f32x16{0};
    auto multiply = [&]() {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            Frag<T> a[G::RT];
#pragma unroll
            for (int rt = 0; rt < G::RT; ++rt) a[rt] = frag_load<T>(sA + (32 * (G::RT * wr + rt) + r) * LDW + 16 * c + 8 * half);
#pragma unroll
            for (int nt = 0; nt < G::NT; ++nt) {
                const Frag<T> w = frag_load<T>(sW + (foff + 32 * nt + r) * LDW + 16 * c + 8 * half);
#pragma unroll
                for (int rt = 0; rt < G::RT; ++rt) mma<T>(acc[rt][nt], w, a[rt]);
            }
        }
    };
    if constexpr (TM == 64 && sizeof(T) == 2) {
        // The small tile is chosen for launches of <= ~2 workgroups per CU, where a workgroup's k loop is a serial
        // latency chain.  It has registers to spare (89 VGPRs), so it keeps TWO tiles in flight: step s+2 is fetched
        // while step s is multiplied and step s+1 already sits in registers (k steps past the end fetch clamped
        // addresses with ok = 0: zeros that are never multiplied).
        TileRegs<T> areg2, wreg2;
        tile_fetch<T, TM>(areg2, p.a, p.lda, m0, M, BK, tid, p.K);
        tile_fetch<T>(wreg2, p.w, p.K, n0, p.N, BK, tid, p.K);
        for (int kc = 0; kc < nk; kc += 2) {
            __syncthreads();
            tile_commit<T, TM>(sA, areg, tid);
            tile_commit<T>(sW, wreg, tid);
            __syncthreads();
            tile_fetch<T, TM>(areg, p.a, p.lda, m0, M, (kc + 2) * BK, tid, p.K);
            tile_fetch<T>(wreg, p.w, p.K, n0, p.N, (kc + 2) * BK, tid, p.K);
            multiply();
            if (kc + 1 >= nk) break;
            __syncthreads();
            tile_commit<T, TM>(sA, areg2, tid);
            tile_commit<T>(sW, wreg2, tid);
            __syncthreads();
            tile_fetch<T, TM>(areg2, p.a, p.lda, m0, M, (kc + 3) * BK, tid, p.K);
            tile_fetch<T>(wreg2, p.w, p.K, n0, p.N, (kc + 3) * BK, tid, p.K);
            multiply();
        }
    } else {
        for (int kc = 0; kc < nk; ++kc) {
            __syncthreads();
            tile_commit<T, TM>(sA, areg, tid);
            tile_commit<T>(sW, wreg, tid);
            __syncthreads();
            if (kc + 1 < nk) {
                tile_fetch<T, TM>(areg, p.a, p.lda, m0, M, (kc + 1) * BK, tid, p.K);
                tile_fetch<T>(wreg, p.w, p.K, n0, p.N, (kc + 1) * BK, tid, p.K);
            }
            multiply();
        }
    }
    __syncthreads();                          // every wave is done with sA / sW: reuse them as the staging tile
    epilogue<T, RELU, TM, DROP>(acc, p, M, sA, m0, n0, tid);
}

// ---------------------------------------------------------------------------
// dW = dY^T X.  Tile 128 (n) x 128 (k); the workgroup's M range is walked in steps of 64 tokens
// whose dY / X tiles are staged TRANSPOSED in LDS ([col][token], 4 tokens per ds_write) so that
// fragments are contiguous along the contraction (token) index.  4 waves as 2 x 2, 64 x 64 each.
constexpr int TK = 64, LDX = TK + 8;

template <typename T> struct TnArgs {
    const T* dy; const T* x; float* slab;
    int M, N, K, ldy, ldx, splits, rows_per_split;
    const int* m_live = nullptr;       // packed stream (LDS-DMA kernel): rows in use; the splits then share THOSE rows evenly
};

template <typename T> MTMP_DEV void store_quad(T* p, T a, T b, T c, T d);
template <> MTMP_DEV void store_quad<bf16>(bf16* p, bf16 a, bf16 b, bf16 c, bf16 d) {
    *reinterpret_cast<bf16x4*>(p) = bf16x4{a, b, c, d};
}
template <> MTMP_DEV void store_quad<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}

// 64 tokens x 128 cols -> registers: thread (q = tid&15: tokens 4q..4q+3, cg = tid>>4: cols 8cg..8cg+7);
// unconditional loads from clamped rows, tokens >= m_end zeroed at commit time (see tile_fetch).
template <typename T> struct TnRegs { Frag<T> f[4]; unsigned ok; };
template <typename T>
MTMP_DEV void tn_fetch(TnRegs<T>& t, const T* src, int ld, int m0, int m_end, int c0, int tid) {
    const int q = (tid & 15) * 4, cg = (tid >> 4) * 8;
    t.ok = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + q + i;
        t.f[i] = frag_load<T>(src + (size_t)min(m, max(m_end - 1, 0)) * ld + c0 + cg);
        t.ok |= (m < m_end) ? (1u << i) : 0u;
    }
}
template <typename T> MTMP_DEV void tn_mask(TnRegs<T>& t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) t.f[i] = frag_keep(t.f[i], (t.ok >> i) & 1u);
}
// registers (4 tokens x 8 cols) -> LDS [col][token]: a 4x8 in-register transpose.  For bf16 it is
// spelled with v_perm_b32 on the packed dwords (element-wise bf16 vector shuffles make hipcc
// round-trip through scratch memory).
template <typename T> MTMP_DEV void tn_commit(T* dst, const Frag<T> (&reg)[4], int tid);
template <> MTMP_DEV void tn_commit<float>(float* dst, const Frag<float> (&reg)[4], int tid) {
    const int q = (tid & 15) * 4, cg = (tid >> 4) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) store_quad<float>(dst + (cg + e) * LDX + q, reg[0].v[e], reg[1].v[e], reg[2].v[e], reg[3].v[e]);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <> MTMP_DEV void tn_commit<bf16>(bf16* dst, const Frag<bf16> (&reg)[4], int tid) {
    const int q = (tid & 15) * 4, cg = (tid >> 4) * 8;
    const u32x4 r0 = __builtin_bit_cast(u32x4, reg[0].v), r1 = __builtin_bit_cast(u32x4, reg[1].v);
    const u32x4 r2 = __builtin_bit_cast(u32x4, reg[2].v), r3 = __builtin_bit_cast(u32x4, reg[3].v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // v_perm_b32(src0, src1, sel): byte i of the result = byte sel[i] of {src0 (4..7), src1 (0..3)}
        const u32x2 even = {__builtin_amdgcn_perm(r1[k], r0[k], 0x05040100u), __builtin_amdgcn_perm(r3[k], r2[k], 0x05040100u)};
        const u32x2 odd = {__builtin_amdgcn_perm(r1[k], r0[k], 0x07060302u), __builtin_amdgcn_perm(r3[k], r2[k], 0x07060302u)};
        *reinterpret_cast<u32x2*>(dst + (cg + 2 * k) * LDX + q) = even;
        *reinterpret_cast<u32x2*>(dst + (cg + 2 * k + 1) * LDX + q) = odd;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TnArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sY = reinterpret_cast<T*>(smem_raw);   // [128 n][LDX tokens]
    T* sX = sY + 128 * LDX;                   // [128 k][LDX tokens]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    const int tn = p.N / 128, tk = p.K / 128;
    int w = xcd_remap(blockIdx.x, gridDim.x);         // the tn*tk tiles of one M split share an XCD: re-reads of its dY / X rows hit that L2
    const int split = w / (tn * tk);
    w -= split * tn * tk;
    const int n0 = (w / tk) * 128, k0 = (w % tk) * 128;
    const int M = live_rows(p.M, p.m_live);                // (packed stream: see gemm_tn_dma_kernel)
    const int rps = p.m_live ? ((M + p.splits - 1) / p.splits + TK - 1) / TK * TK : p.rows_per_split;
    const int m_lo = split * rps, m_end = min(M, m_lo + rps);
    const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
    f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // column sums of dY (bias gradient)
    TnRegs<T> yreg, xreg;
    tn_fetch<T>(yreg, p.dy, p.ldy, m_lo, m_end, n0, tid);
    tn_fetch<T>(xreg, p.x, p.ldx, m_lo, m_end, k0, tid);
    for (int m0 = m_lo; m0 < m_end; m0 += TK) {
        __syncthreads();
        tn_mask<T>(yreg);
        tn_mask<T>(xreg);
        tn_commit<T>(sY, yreg.f, tid);
        tn_commit<T>(sX, xreg.f, tid);
        if (k0 == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                csum[e] += to_f32(yreg.f[0].v[e]) + to_f32(yreg.f[1].v[e]) + to_f32(yreg.f[2].v[e]) + to_f32(yreg.f[3].v[e]);
        }
        __syncthreads();
        if (m0 + TK < m_end) {
            tn_fetch<T>(yreg, p.dy, p.ldy, m0 + TK, m_end, n0, tid);
            tn_fetch<T>(xreg, p.x, p.ldx, m0 + TK, m_end, k0, tid);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const Frag<T> a0 = frag_load<T>(sY + (wn + r) * LDX + 16 * c + 8 * half);
            const Frag<T> a1 = frag_load<T>(sY + (wn + 32 + r) * LDX + 16 * c + 8 * half);
            const Frag<T> b0 = frag_load<T>(sX + (wk + r) * LDX + 16 * c + 8 * half);
            const Frag<T> b1 = frag_load<T>(sX + (wk + 32 + r) * LDX + 16 * c + 8 * half);
            mma<T>(acc[0][0], a0, b0); mma<T>(acc[0][1], a0, b1);
            mma<T>(acc[1][0], a1, b0); mma<T>(acc[1][1], a1, b1);
        }
    }
    // partial slab row: [N*K] products then [N] column sums
    float* out = p.slab + (size_t)split * ((size_t)p.N * p.K + p.N);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t)
                out[(size_t)(n0 + wn + 32 * i + acc_row(t, half)) * p.K + k0 + wk + 32 * j + r] = acc[i][j][t];
    if (k0 == 0) {
        // threads with equal (tid>>4) hold the same 8 columns for different token quads: lanes 16g..16g+15
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float s = csum[e];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if ((tid & 15) == 0) out[(size_t)p.N * p.K + n0 + (tid >> 4) * 8 + e] = s;
        }
    }
}

// ---------------------------------------------------------------------------
// bf16 build of dW = dY^T X: same tiling and slab protocol as gemm_tn_kernel, but the 64-token tiles are
// kept in LDS in their NATURAL [token][col] layout (plain 16-byte copies, no register transposes) and the
// MFMA fragments -- 8 consecutive TOKENS of one column -- come from the transposing LDS read
// ds_read_b64_tr_b16.  Two LDS stages (one barrier per 64 tokens) and two register stages: the global loads
// of step s+3 are issued while step s is multiplied, so a step's loads have two full steps to arrive.
// (Ablations of the first version, profiles/: the in-register transposes + masks + two barriers per step cost
//  as much as the memory traffic, and loads were only in flight during the short MFMA phase.)
constexpr int TT = 64;            // tokens per stage
constexpr int LDG = 128 + 32;     // LDS row stride, elements: 320 B = 16 banks mod 64 -> conflict-free ds_read_b64_tr_b16
struct TrRegs { u32x4_t q[4]; unsigned ok; };

// 64 tokens x 128 cols -> 4 x 16 B per thread: thread (rg = tid >> 4, ch = tid & 15) holds rows 16i + rg, cols 8ch..8ch+7
MTMP_DEV void tr_fetch(TrRegs& t, const bf16* src, int ld, int m0, int m_end, int c0, int tid) {
    const int rg = tid >> 4, ch = tid & 15;
    t.ok = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 16 * i + rg;
        t.q[i] = *reinterpret_cast<const u32x4_t*>(src + (size_t)min(m, m_end - 1) * ld + c0 + 8 * ch);
        t.ok |= (m < m_end) ? (1u << i) : 0u;
    }
}
MTMP_DEV void tr_commit(bf16* dst, TrRegs& t, int tid) {
    const int rg = tid >> 4, ch = tid & 15;
    if (!wave_all(t.ok == 15u)) {              // token tail of the split: rows past m_end contribute zero
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned m = (t.ok >> i) & 1u ? 0xFFFFFFFFu : 0u;
            t.q[i] &= u32x4_t{m, m, m, m};
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4_t*>(dst + (16 * i + rg) * LDG + 8 * ch) = t.q[i];
}
// fragment of the transposed role on a [token][LDG] image: element j = img[row0 + 8*half + j][col0 + r]
MTMP_DEV Frag<bf16> frag_tr_g(const bf16* img, int row0, int col0, int lane) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int G = lane >> 4, i = lane & 15;
    const bf16* a = img + (row0 + 8 * (G >> 1) + (i >> 2)) * LDG + col0 + 16 * (G & 1) + 4 * (i & 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * LDG));
    Frag<bf16> f;
    f.v = __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    return f;
}
MTMP_DEV void tr_csum(float (&csum)[8], const TrRegs& y) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf16x8 v = __builtin_bit_cast(bf16x8, y.q[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += (float)v[e];
    }
}
MTMP_DEV void tr_mma(f32x16 (&acc)[2][2], const bf16* sY, const bf16* sX, int wn, int wk, int lane) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const Frag<bf16> a0 = frag_tr_g(sY, 16 * c, wn, lane), a1 = frag_tr_g(sY, 16 * c, wn + 32, lane);
        const Frag<bf16> b0 = frag_tr_g(sX, 16 * c, wk, lane), b1 = frag_tr_g(sX, 16 * c, wk + 32, lane);
        mma<bf16>(acc[0][0], a0, b0); mma<bf16>(acc[0][1], a0, b1);
        mma<bf16>(acc[1][0], a1, b0); mma<bf16>(acc[1][1], a1, b1);
    }
}

// NG = 1: 256 threads, two workgroups per CU.  NG = 2 (large M): 512 threads = two GROUPS of four waves that share the output
// tile and split every 128-token step between them (each group is the NG = 1 workgroup on its own 64 tokens and its own half
// of the LDS); at the end group 1 hands its accumulators over through LDS and group 0 writes ONE partial slab: the same eight
// waves per CU, half the slabs -- the split-M partials (33 MB written here and read again by the reduce launch per launch at
// config 2, against 132-165 MB of operands) were a fifth of this HBM-bound kernel's traffic.
template <int NG>
__global__ __launch_bounds__(256 * NG, NG == 1 ? 2 : 1) void gemm_tn_tr_kernel(TnArgs<bf16> p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int MAT = TT * LDG, GRP = 2 * MAT, STAGE = NG * GRP;
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    bf16* sm = reinterpret_cast<bf16*>(smem_raw) + grp * GRP;   // [2 stages][NG groups][dY | X][TT][LDG]
    const int tn = p.N / 128, tk = p.K / 128;
    int w = xcd_remap(blockIdx.x, gridDim.x);
    const int split = w / (tn * tk);
    w -= split * tn * tk;
    const int n0 = (w / tk) * 128, k0 = (w % tk) * 128;
    const int M = live_rows(p.M, p.m_live);                // (packed stream: see gemm_tn_dma_kernel)
    const int rps = p.m_live ? ((M + p.splits - 1) / p.splits + TK - 1) / TK * TK : p.rows_per_split;
    const int m_lo = split * rps, m_end = min(M, m_lo + rps);
    const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
    constexpr int ST = NG * TT;                            // tokens per step (all groups)
    const int nsteps = m_end > m_lo ? (m_end - m_lo + ST - 1) / ST : 0;
    const int mg = m_lo + grp * TT;                        // this group's first token
    f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool bias_blk = (k0 == 0);
    TrRegs y0, x0, y1, x1;
    // steps past the end fetch clamped rows with ok = 0: their tiles are all zero
    tr_fetch(y0, p.dy, p.ldy, mg, m_end, n0, tid);             tr_fetch(x0, p.x, p.ldx, mg, m_end, k0, tid);
    tr_fetch(y1, p.dy, p.ldy, mg + ST, m_end, n0, tid);        tr_fetch(x1, p.x, p.ldx, mg + ST, m_end, k0, tid);
    if (nsteps > 0) {
        tr_commit(sm, y0, tid); tr_commit(sm + MAT, x0, tid);
        if (bias_blk) tr_csum(csum, y0);
        tr_fetch(y0, p.dy, p.ldy, mg + 2 * ST, m_end, n0, tid); tr_fetch(x0, p.x, p.ldx, mg + 2 * ST, m_end, k0, tid);
    }
    __syncthreads();
    // (ablation builds, tools/dbg/ablate_tn.sh: results are wrong by design, only the timing is read)
    for (int s = 0; s < nsteps; s += 2) {
        // even step: multiply stage 0, stage 1 <- registers y1/x1 (step s+1), refill them with step s+3
        tr_mma(acc, sm, sm + MAT, wn, wk, lane);
        tr_commit(sm + STAGE, y1, tid); tr_commit(sm + STAGE + MAT, x1, tid);
        if (bias_blk) tr_csum(csum, y1);
        tr_fetch(y1, p.dy, p.ldy, mg + (s + 3) * ST, m_end, n0, tid); tr_fetch(x1, p.x, p.ldx, mg + (s + 3) * ST, m_end, k0, tid);
        __syncthreads();
        if (s + 1 >= nsteps) break;
        // odd step: multiply stage 1, stage 0 <- y0/x0 (step s+2), refill with step s+4
        tr_mma(acc, sm + STAGE, sm + STAGE + MAT, wn, wk, lane);
        tr_commit(sm, y0, tid); tr_commit(sm + MAT, x0, tid);
        if (bias_blk) tr_csum(csum, y0);
        tr_fetch(y0, p.dy, p.ldy, mg + (s + 4) * ST, m_end, n0, tid); tr_fetch(x0, p.x, p.ldx, mg + (s + 4) * ST, m_end, k0, tid);
        __syncthreads();
    }
    if constexpr (NG == 2) {
        // group 1 -> LDS [64 accumulator registers][256 threads] (64 KiB, over the dead tiles) -> group 0 adds
        float* xch = reinterpret_cast<float*>(smem_raw);
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 16; ++t) xch[((i * 2 + j) * 16 + t) * 256 + tid] = acc[i][j][t];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc[i][j][t] += xch[((i * 2 + j) * 16 + t) * 256 + tid];
        }
    }
    float* out = p.slab + (size_t)split * ((size_t)p.N * p.K + p.N);
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    out[(size_t)(n0 + wn + 32 * i + acc_row(t, half)) * p.K + k0 + wk + 32 * j + r] = acc[i][j][t];
    }
    if (bias_blk) {
        // thread (rg, ch) holds partial sums of columns 8ch..8ch+7 over its rows: 16 NG row groups -> LDS -> one sum
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);   // [16 NG][128]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(grp * 16 + (tid >> 4)) * 128 + 8 * (tid & 15) + e] = csum[e];
        __syncthreads();
        if (threadIdx.x < 128) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 16 * NG; ++g) sum += red[g * 128 + threadIdx.x];
            out[(size_t)p.N * p.K + n0 + threadIdx.x] = sum;
        }
    }
}

// ---------------------------------------------------------------------------
// Large-M form of the product above: tiles by LDS-DMA (global_load_lds_dwordx4), loader waves and matrix waves.
// What bounded gemm_tn_tr_kernel<2> at config 2 (tools/dbg/ablate_tn.sh, stamp_tn.py, pmc_tn.sh, prof_tn.sh; 48-52 us + 5.5 us
// of reduce): every wave ran the same sequence -- fetch, ds_write_b128 commit (a 64 KiB step costs ~830 LDS cycles), 16 MFMAs --
// between two barriers, and each of those blocks the in-order wave while it queues, so a step cost the SUM of its parts (~4000
// cycles against 1024 of MFMA; matrix pipe 28 % busy, no LDS bank conflicts, clock 1.9-2.1 GHz) and removing any one part
// removed only that part.  A first DMA version with eight identical waves (no commits, five stages in flight) ran at exactly the
// same speed: the waves stood ~500 cycles per stage in the DMA issue queue with the matrix pipe idle.
// Here the roles are split: waves 0-3 (one per SIMD) read fragments and issue MFMAs (a 64-row block of the tile each, over ALL
// tokens of a stage: one partial slab per workgroup, no accumulator hand-over), waves 4-7 (their SIMD partners) only issue DMA
// and wait for it.  A stage is 64 tokens of [64][128] tiles (unpadded 256-byte rows whose 64-byte blocks are XOR-ed with
// row & 3 on the SOURCE side -- what makes the 4-rows-by-64-bytes footprint of a ds_read_b64_tr_b16 lane group conflict-free),
// one barrier per stage.  Token tail: rows past m_end are fetched from the clamped last row and the dY rows are zeroed in LDS
// (one stage per launch).  The bias gradient (column sums of dY) rides on the matrix pipe: dY^T x ones for one 32-column block
// per matrix wave (+12 % MFMAs; from LDS with vector adds it cost 7 us of the launch wherever it was placed).
// XT = 1 (shipped): 128 x 128 tiles, three stages of dY | X, matrix waves 64 x 64, 16.5 MB of partial slabs at config 2: 38 / 46 us
// per launch (kernel trace).  XT = 2 (-DMTMP_TN_WIDE): 128 x 256 tiles, three stages of dY | X lo | X hi, matrix waves 64 x 128:
// 297 instead of 396 MB from L2 into the CUs and 33 / 40 us per launch -- its pure-MFMA loop (no DMA, no reads, no barriers, no
// stores) takes 19 us at the 1.88 GHz the chip holds -- but 33 MB of slabs, which the step's HBM-sharing streams pay for
// (tn_launch_splits below).
constexpr int DT = 64;                                     // tokens per stage
constexpr int DTILE = DT * 256;                            // bytes: one [DT][128] bf16 tile
template <int XT> struct TnDma {
    // three stages (96 KiB) rather than five (all 160 KiB): the loaders idle at the barrier either way, and 64 KiB of the CU stay
    // free for the other streams' workgroups -- 8.80-8.82 against 8.85-8.88 ms/step and 9.03-9.11 against 9.07-9.14 (two boxes)
    static constexpr int DNS = XT == 2 ? 3 : 3;            // stages
    static constexpr int DSTAGE = (1 + XT) * DTILE;        // a stage = dY | X (lo | hi)
    static constexpr int DPW = 4 * (1 + XT);               // DMA pieces per loader wave and stage
    static constexpr int NB = 2 * XT;                      // 32-column X blocks per matrix wave
};
MTMP_DEV Frag<bf16> frag_tr_at(const char* a) {            // lane's 8-byte piece in token rows r and r + 4 of a 256-byte-row image
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * 256));
    Frag<bf16> f;
    f.v = __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    return f;
}
template <int N> MTMP_DEV void tn_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int DPW> MTMP_DEV void tn_wait_stages(int k) {   // all but the last k stages (DPW pieces each) of this wave have landed
    if (k >= 4) tn_wait<4 * DPW>();
    else if (k == 3) tn_wait<3 * DPW>();
    else if (k == 2) tn_wait<2 * DPW>();
    else if (k == 1) tn_wait<DPW>();
    else tn_wait<0>();
}
template <int NB> struct TnChunk { Frag<bf16> a[2], b[NB]; };
template <int XT>
__global__ __launch_bounds__(512, 1) void gemm_tn_dma_kernel(Grouped<TnArgs<bf16>> grp) {
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = grp_find(grp, wg);
    const TnArgs<bf16>& p = grp.seg[seg];
    constexpr int DNS = TnDma<XT>::DNS, DSTAGE = TnDma<XT>::DSTAGE, DPW = TnDma<XT>::DPW, NB = TnDma<XT>::NB;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem_raw);
    const int tid = threadIdx.x & 255, lane = tid & 63, r = lane & 31, half = lane >> 5;
    const bool loader = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = p.N / 128, tk = p.K / (128 * XT);
    int w = wg - grp.first[seg];
    const int split = w / (tn * tk);
    w -= split * tn * tk;
    const int kt = w % tk, n0 = (w / tk) * 128, k0 = kt * 128 * XT;
    const int M = live_rows(p.M, p.m_live);
    // packed stream: the grid (and the slab count) belong to the padded maximum; every split takes its share of the live rows
    const int rps = p.m_live ? ((M + p.splits - 1) / p.splits + TK - 1) / TK * TK : p.rows_per_split;
    const int m_lo = split * rps, m_end = min(M, m_lo + rps);
    const int wn = (wave >> 1) * 64, xt = wave & 1;        // matrix wave: dY columns wn .. wn + 63, X tile xt / X columns 64 xt ..
    const int kcol = (XT == 2 ? 128 : 64) * xt;            // its first output column inside the tile
    const int nst = m_end > m_lo ? (m_end - m_lo + DT - 1) / DT : 0;
    // ---- loader wave `wave`: rows 16 wave .. 16 wave + 15 of the three tiles (four 1 KiB pieces each)
    const int lr = lane >> 4, lc = (lane & 15) ^ (4 * lr);           // row inside a piece; logical 16-byte chunk this lane fetches
    const unsigned ycol = (unsigned)(n0 * 2 + lc * 16), xcol = (unsigned)(k0 * 2 + lc * 16);
    const unsigned ldyb = (unsigned)p.ldy * 2u, ldxb = (unsigned)p.ldx * 2u;
    auto issue = [&](int st_) {
        const unsigned dst = lds0 + (unsigned)((st_ % DNS) * DSTAGE + wave * 4096);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned row = (unsigned)min(m_lo + st_ * DT + 16 * wave + 4 * q + lr, m_end - 1);
            dma16(row * ldyb + ycol, p.dy, dst + q * 1024);
            dma16(row * ldxb + xcol, p.x, dst + DTILE + q * 1024);
            if constexpr (XT == 2) dma16(row * ldxb + xcol + 256, p.x, dst + 2 * DTILE + q * 1024);
        }
    };
    // ---- matrix wave: lane (G = lane >> 4, i = lane & 15) reads token rows 8 (G >> 1) + (i >> 2) (+ 4), bytes 32 (G & 1) + 8 (i & 3)
    const int G = lane >> 4, i16 = lane & 15, sw = 4 * (i16 >> 2);
    const int rowoff = (8 * (G >> 1) + (i16 >> 2)) * 256 + 32 * (G & 1) + 8 * (i16 & 3);
    const int aoff0 = rowoff + 16 * ((wn >> 3) ^ sw), aoff1 = rowoff + 16 * (((wn + 32) >> 3) ^ sw);
    const int xbase = DTILE * (XT == 2 ? 1 + xt : 1) + rowoff, xc0 = XT == 2 ? 0 : 8 * xt;       // (first 16-byte chunk of the wave's X columns)
    int boff[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) boff[j] = xbase + 16 * ((xc0 + 4 * j) ^ sw);
    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x16{0};
    // bias gradient (column sums of dY) on the matrix pipe: dY^T x ones for ONE 32-column block of the tile per matrix wave.
    // Block b = 2 (wn / 64) + ai belongs to K tile b mod min(tk, 4), so the K tiles share the 128 columns.  With one K tile the
    // two waves of a column half take one block each (ai = xt) over all k-chunks; otherwise they split the chunks of their one
    // block (xt = 0: chunks 0, 1; xt = 1: chunks 2, 3).
    const int tkp = min(tk, 4), b0 = 2 * (wn >> 6);
    const bool own0 = b0 % tkp == kt, own1 = (b0 + 1) % tkp == kt;
    const int cs_ai = (own0 && own1) ? xt : own0 ? 0 : own1 ? 1 : -1;
    const bool cs_hi = cs_ai == 1;
    const bool cs_c01 = cs_ai >= 0 && ((own0 && own1) || xt == 0), cs_c23 = cs_ai >= 0 && ((own0 && own1) || xt == 1);
    Frag<bf16> ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones.v[j] = (bf16)1.0f;
    f32x16 acc_cs = {0};
    auto chunk = [&](int st_, int c) {
        const char* sb = smem_raw + (st_ % DNS) * DSTAGE + c * 4096;
        TnChunk<NB> f;
        f.a[0] = frag_tr_at(sb + aoff0);
#pragma unroll
        for (int j = 0; j < NB; ++j) f.b[j] = frag_tr_at(sb + boff[j]);
        f.a[1] = frag_tr_at(sb + aoff1);
        return f;
    };
    auto mma8 = [&](const TnChunk<NB>& f, bool with_cs) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) mma<bf16>(acc[i][j], f.a[i], f.b[j]);
        if (with_cs) {
            Frag<bf16> a;
            a.v = cs_hi ? f.a[1].v : f.a[0].v;
            mma<bf16>(acc_cs, a, ones);
        }
    };
    // stage st_ has landed for every wave (barrier passed): in a tail stage zero the dY rows past m_end (their products and
    // their column sums vanish), then one more barrier (every thread: row = t >> 3, 32 bytes)
    auto fix_tail = [&](int st_) {
        if (st_ < nst && m_lo + (st_ + 1) * DT > m_end) {
            const int row = (int)threadIdx.x >> 3;
            if (m_lo + st_ * DT + row >= m_end) {
                char* d = smem_raw + (st_ % DNS) * DSTAGE + row * 256 + ((int)threadIdx.x & 7) * 32;
                *reinterpret_cast<u32x4_t*>(d) = u32x4_t{0, 0, 0, 0};
                *reinterpret_cast<u32x4_t*>(d + 16) = u32x4_t{0, 0, 0, 0};
            }
            __syncthreads();
        }
    };
    // Barrier B(s) publishes stage s (the loaders waited for their pieces of it) and tells the loaders that the matrix waves
    // are done reading stage s-1, whose buffer takes stage s+DNS-1.  Two loops, one per role, executing the same barriers (no
    // MFMA shares a control-flow merge with loader code).
    if (loader) {
        const int pre = min(nst, DNS);
        for (int s = 0; s < pre; ++s) issue(s);
        for (int s = 0; s < nst; ++s) {
            tn_wait_stages<DPW>(min(nst - 1, max(DNS - 1, s + DNS - 2)) - s);
            __syncthreads();                                 // B(s)
            fix_tail(s);
            if (s >= 1 && s + DNS - 1 < nst) issue(s + DNS - 1);
        }
    } else {
        // one stage: each k-chunk's fragments requested one chunk (8 MFMAs) ahead of its MFMAs (pinned in this order)
        for (int s = 0; s < nst; ++s) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();                                 // B(s)
            fix_tail(s);
            const TnChunk<NB> f0 = chunk(s, 0);
            __builtin_amdgcn_sched_barrier(0);
            const TnChunk<NB> f1 = chunk(s, 1);
            mma8(f0, cs_c01);
            __builtin_amdgcn_sched_barrier(0);
            const TnChunk<NB> f2 = chunk(s, 2);
            mma8(f1, cs_c01);
            __builtin_amdgcn_sched_barrier(0);
            const TnChunk<NB> f3 = chunk(s, 3);
            mma8(f2, cs_c23);
            __builtin_amdgcn_sched_barrier(0);
            mma8(f3, cs_c23);
        }
    }
    float* red = reinterpret_cast<float*>(smem_raw);       // [2][128 columns] over the dead tiles
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 256) red[threadIdx.x] = 0.f;
    __syncthreads();
    if (!loader && cs_ai >= 0 && r == 0) {
#pragma unroll
        for (int t = 0; t < 16; ++t) red[xt * 128 + wn + 32 * cs_ai + acc_row(t, half)] = acc_cs[t];
    }
    __syncthreads();
    float* out = p.slab + (size_t)split * ((size_t)p.N * p.K + p.N);
    if (!loader) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    out[(size_t)(n0 + wn + 32 * i + acc_row(t, half)) * p.K + k0 + kcol + 32 * j + r] = acc[i][j][t];
    } else if (tid < 128 && (tid >> 5) % tkp == kt) {
        out[(size_t)p.N * p.K + n0 + tid] = red[tid] + red[128 + tid];
    }
}

// Several slab reductions in ONE launch: out_e[c] = sum_r slab_e[r][c] for up to RB_MAX entries (blockIdx.y); columns below
// split_e go to a_e, the rest to b_e (the weight / bias gradient of a split-M product share a slab row).  A layer's backward
// produced seven of these reductions (3 weight gradients, 2 x 2 levels of LayerNorm gamma / beta partials): seven launches of
// 5-13 us each on the vital-sign stream, and worse on the image / text streams, whose launches wait for a free CU.
// ld_e: floats between two rows of slab_e (= cols_e for a whole slab; larger when the entry is a column range of a wider slab --
// mtmp_reduce_scatter: one slab's column ranges summed straight into separate destinations, e.g. slices of the flat gradient).
constexpr int RB_MAX = 12;
struct ReduceBatch { const float* slab[RB_MAX]; float* a[RB_MAX]; float* b[RB_MAX]; long long cols[RB_MAX], split[RB_MAX], ld[RB_MAX]; int rows[RB_MAX]; };
__global__ __launch_bounds__(256) void reduce_batch_kernel(ReduceBatch t) {
    __shared__ f32x4 part4[4][64];
    const int e = blockIdx.y;
    const long long cols = t.cols[e];
    const int rows = t.rows[e], rl = threadIdx.x >> 6, cl = threadIdx.x & 63;
    const float* slab = t.slab[e];
    const size_t ld = (size_t)t.ld[e];
    if ((cols & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)slab & 15) == 0) {
        // four columns per thread (16-byte loads, four rows in flight per row lane); the slabs are the bulk of a layer's
        // reduction traffic and 4-byte loads kept this launch at ~1.6 TB/s.  Row lanes per block by the slab's height: the
        // LayerNorm partials (one row per 128 tokens: 503 rows of 512 columns at config 2) would otherwise be two blocks walking
        // 126 dependent rounds each -- the longest chain of the launch.
        const int RL = rows >= 256 ? 64 : rows >= 64 ? 16 : 4, CQ = 256 / RL;
        const int rl4 = threadIdx.x / CQ, cq = threadIdx.x % CQ;
        f32x4* part = &part4[0][0];                        // [RL][CQ]
        for (long long c0 = (long long)blockIdx.x * 4 * CQ; c0 < cols; c0 += (long long)gridDim.x * 4 * CQ) {
            const long long c = c0 + 4 * cq;
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
            if (c < cols) {
                const float* col = slab + c;
                int q = rl4;
                for (; q + 3 * RL < rows; q += 4 * RL) {
                    s0 += ld4f(col + (size_t)q * ld);
                    s1 += ld4f(col + (size_t)(q + RL) * ld);
                    s2 += ld4f(col + (size_t)(q + 2 * RL) * ld);
                    s3 += ld4f(col + (size_t)(q + 3 * RL) * ld);
                }
                for (; q < rows; q += RL) s0 += ld4f(col + (size_t)q * ld);
            }
            __syncthreads();
            part[rl4 * CQ + cq] = (s0 + s1) + (s2 + s3);
            __syncthreads();
            if (rl4 == 0 && c < cols) {
                f32x4 s = part[cq];
                for (int l = 1; l < RL; ++l) s += part[l * CQ + cq];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (c + i < t.split[e]) t.a[e][c + i] = s[i];
                    else if (t.b[e]) t.b[e][c + i - t.split[e]] = s[i];
                }
            }
        }
        return;
    }
    float (*part)[64] = reinterpret_cast<float (*)[64]>(&part4[0][0]);
    for (long long c0 = (long long)blockIdx.x * 64; c0 < cols; c0 += (long long)gridDim.x * 64) {
        const long long c = c0 + cl;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (c < cols) {
            int q = rl;
            for (; q + 12 < rows; q += 16) {                      // four independent loads in flight per row lane
                s0 += slab[(size_t)q * ld + c];
                s1 += slab[(size_t)(q + 4) * ld + c];
                s2 += slab[(size_t)(q + 8) * ld + c];
                s3 += slab[(size_t)(q + 12) * ld + c];
            }
            for (; q < rows; q += 4) s0 += slab[(size_t)q * ld + c];
        }
        __syncthreads();
        part[rl][cl] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (rl == 0 && c < cols) {
            const float s = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
            if (c < t.split[e]) t.a[e][c] = s;
            else if (t.b[e]) t.b[e][c - t.split[e]] = s;
        }
    }
}

int tn_splits(int M, int N, int K, int target_wgs) {
    const int tiles = (N / 128) * (K / 128);
    int s = target_wgs / tiles;
    const int max_s = (M + 4 * TK - 1) / (4 * TK);         // at least 4 token steps per split
    if (s > max_s) s = max_s;
    return s < 1 ? 1 : s;
}

// segments -> one grid: first[i] = first block of segment i; unused entries repeat segment 0 behind the end of the grid
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

// mode: 0 forward (relu / dropout from the arguments; a.signs != nullptr with relu: also write the sign bits), 1 gate by a.signs
// n segments (token streams) in one launch: same N / activation / dropout switch / sign bits / key-norm table in all of them
int launch_ln_gemm_dma(int n, const GemmArgs<bf16>* segs, int relu, int gate, hipStream_t st) {
    using P = PanelDma;
    const GemmArgs<bf16>& a = segs[0];
    const bool drop = a.drop_p > 0.f, signs = !gate && relu && a.signs;
    for (int i = 1; i < n; ++i)
        if (segs[i].N != a.N || (segs[i].drop_p > 0.f) != drop || !segs[i].signs != !a.signs || !segs[i].knorm != !a.knorm ||
            !segs[i].gamma != !a.gamma) {
            mtmp_set_error("mtmp_ln_gemm (grouped): the streams of one launch must agree in N, dropout, sign bits and key norms");
            return MTMP_ERR_ARG;
        }
    const void* fs[7] = {(const void*)ln_gemm_dma_kernel<false, false, false, false>, (const void*)ln_gemm_dma_kernel<false, true, false, false>,
                         (const void*)ln_gemm_dma_kernel<true, false, false, false>, (const void*)ln_gemm_dma_kernel<true, true, false, false>,
                         (const void*)ln_gemm_dma_kernel<true, false, false, true>, (const void*)ln_gemm_dma_kernel<true, true, false, true>,
                         (const void*)ln_gemm_dma_kernel<false, false, true, false>};
    const bool knorm = a.knorm != nullptr;           // (the Q/K/V projection: no activation, no dropout, N = 768)
    const int which = gate ? 6 : signs ? 4 + (drop ? 1 : 0) : (relu ? 2 : 0) + (drop ? 1 : 0);
    const void* fk = (const void*)ln_gemm_dma_kernel<false, false, false, false, true>;
    if (hipFuncSetAttribute(knorm ? fk : fs[which], hipFuncAttributeMaxDynamicSharedMemorySize, (int)P::lds_bytes) != hipSuccess) {
        mtmp_set_error("mtmp_ln_gemm: cannot raise dynamic LDS to %zu", P::lds_bytes);
        return MTMP_ERR_LAUNCH;
    }
    // as launch_ln_gemm below; a workgroup's bias block in LDS holds at most MAXP panels
    int mtiles;
    const Grouped<GemmArgs<bf16>> g = make_group(n, segs, [](const GemmArgs<bf16>& x) { return (x.M + BM - 1) / BM; }, mtiles);
    const int npanels = a.N / P::NP;
    int nsplit = 512 / mtiles;
    nsplit = nsplit < 1 ? 1 : (nsplit > npanels ? npanels : nsplit);
    const int need = (npanels + P::MAXP - 1) / P::MAXP;
    if (nsplit < need) nsplit = need;
    dim3 grid(mtiles, nsplit);
    if (knorm) {
        hipLaunchKernelGGL((ln_gemm_dma_kernel<false, false, false, false, true>), grid, dim3(256), P::lds_bytes, st, g);
        MTMP_CHECK_LAUNCH("mtmp_ln_gemm_qkv");
        return MTMP_OK;
    }
    switch (which) {
    case 0: hipLaunchKernelGGL((ln_gemm_dma_kernel<false, false, false, false>), grid, dim3(256), P::lds_bytes, st, g); break;
    case 1: hipLaunchKernelGGL((ln_gemm_dma_kernel<false, true, false, false>), grid, dim3(256), P::lds_bytes, st, g); break;
    case 2: hipLaunchKernelGGL((ln_gemm_dma_kernel<true, false, false, false>), grid, dim3(256), P::lds_bytes, st, g); break;
    case 3: hipLaunchKernelGGL((ln_gemm_dma_kernel<true, true, false, false>), grid, dim3(256), P::lds_bytes, st, g); break;
    case 4: hipLaunchKernelGGL((ln_gemm_dma_kernel<true, false, false, true>), grid, dim3(256), P::lds_bytes, st, g); break;
    case 5: hipLaunchKernelGGL((ln_gemm_dma_kernel<true, true, false, true>), grid, dim3(256), P::lds_bytes, st, g); break;
    default: hipLaunchKernelGGL((ln_gemm_dma_kernel<false, false, true, false>), grid, dim3(256), P::lds_bytes, st, g); break;
    }
    MTMP_CHECK_LAUNCH("mtmp_ln_gemm");
    return MTMP_OK;
}
template <typename T>
int launch_ln_gemm(GemmArgs<T> a, int relu, hipStream_t st) {
    const size_t sm = Panel<T>::lds_bytes;
    if (a.N % Panel<T>::NP != 0) {
        mtmp_set_error("mtmp_ln_gemm: N=%d must be a multiple of %d for this dtype", a.N, Panel<T>::NP);
        return MTMP_ERR_ARG;
    }
    if constexpr (sizeof(T) == 2) return launch_ln_gemm_dma(1, &a, relu, 0, st);
    if (sm > 48 * 1024) {
        const void* fs[4] = {(const void*)ln_gemm_kernel<T, false, false>, (const void*)ln_gemm_kernel<T, false, true>,
                             (const void*)ln_gemm_kernel<T, true, false>, (const void*)ln_gemm_kernel<T, true, true>};
        const void* f = fs[(relu ? 2 : 0) + (a.drop_p > 0.f ? 1 : 0)];
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) {
            mtmp_set_error("mtmp_ln_gemm: cannot raise dynamic LDS to %zu", sm);
            return MTMP_ERR_LAUNCH;
        }
    }
    // one workgroup per 128 rows walks all panels when the row blocks alone fill the chip (2 per CU);
    // otherwise the panels of a row block are split over gridDim.y workgroups
    const int mtiles = (a.M + BM - 1) / BM, npanels = a.N / Panel<T>::NP;
    int nsplit = 512 / mtiles;
    nsplit = nsplit < 1 ? 1 : (nsplit > npanels ? npanels : nsplit);
    dim3 grid(mtiles, nsplit);
    const bool drop = a.drop_p > 0.f;         // (RELU, DROP) are template parameters: no per-element selects in the epilogue
    if (relu && drop)       hipLaunchKernelGGL((ln_gemm_kernel<T, true, true>), grid, dim3(256), sm, st, a);
    else if (relu)          hipLaunchKernelGGL((ln_gemm_kernel<T, true, false>), grid, dim3(256), sm, st, a);
    else if (drop)          hipLaunchKernelGGL((ln_gemm_kernel<T, false, true>), grid, dim3(256), sm, st, a);
    else                    hipLaunchKernelGGL((ln_gemm_kernel<T, false, false>), grid, dim3(256), sm, st, a);
    MTMP_CHECK_LAUNCH("mtmp_ln_gemm");
    return MTMP_OK;
}
template <typename T, int TM>
int launch_gemm_nt_tm(int n, const GemmArgs<T>* segs, int relu, hipStream_t st) {
    const GemmArgs<T>& a = segs[0];
    size_t sm = (size_t)(TM + BN) * LDW * sizeof(T);
    const size_t stage = (size_t)TM * LDO * sizeof(T);
    if (sm < stage) sm = stage;
    if (sm > 48 * 1024) {
        const void* fs[4] = {(const void*)gemm_nt_kernel<T, false, TM, false>, (const void*)gemm_nt_kernel<T, false, TM, true>,
                             (const void*)gemm_nt_kernel<T, true, TM, false>, (const void*)gemm_nt_kernel<T, true, TM, true>};
        const void* f = fs[(relu ? 2 : 0) + (a.drop_p > 0.f ? 1 : 0)];
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) {
            mtmp_set_error("mtmp_gemm_nt: cannot raise dynamic LDS to %zu", sm);
            return MTMP_ERR_LAUNCH;
        }
    }
    int nwg;
    const Grouped<GemmArgs<T>> g = make_group(n, segs, [](const GemmArgs<T>& x) { return ((x.M + TM - 1) / TM) * ((x.N + BN - 1) / BN); }, nwg);
    dim3 grid(nwg);
    const bool drop = a.drop_p > 0.f;
    if (relu && drop)       hipLaunchKernelGGL((gemm_nt_kernel<T, true, TM, true>), grid, dim3(256), sm, st, g);
    else if (relu)          hipLaunchKernelGGL((gemm_nt_kernel<T, true, TM, false>), grid, dim3(256), sm, st, g);
    else if (drop)          hipLaunchKernelGGL((gemm_nt_kernel<T, false, TM, true>), grid, dim3(256), sm, st, g);
    else                    hipLaunchKernelGGL((gemm_nt_kernel<T, false, TM, false>), grid, dim3(256), sm, st, g);
    MTMP_CHECK_LAUNCH("mtmp_gemm_nt");
    return MTMP_OK;
}
// n segments in one launch: same N, K, activation, dropout switch; the tile height by the size of the whole grid
template <typename T>
int launch_gemm_nt(int n, const GemmArgs<T>* segs, int relu, hipStream_t st) {
    long long wgs128 = 0;
    for (int i = 0; i < n; ++i) {
        wgs128 += (long long)((segs[i].M + 127) / 128) * ((segs[i].N + BN - 1) / BN);
        if (segs[i].N != segs[0].N || segs[i].K != segs[0].K || segs[i].act != segs[0].act ||
            (segs[i].drop_p > 0.f) != (segs[0].drop_p > 0.f)) {
            mtmp_set_error("mtmp_gemm_nt (grouped): the streams of one launch must agree in N, K, activation and dropout");
            return MTMP_ERR_ARG;
        }
    }
    return wgs128 < 512 ? launch_gemm_nt_tm<T, 64>(n, segs, relu, st) : launch_gemm_nt_tm<T, 128>(n, segs, relu, st);
}
// ---------------------------------------------------------------------------
// dX of a LayerNorm-fed projection fused with that LayerNorm's backward (autograd of module.py:138-144 in front of
// attention.py:68-70 / module.py:74-77):
//     dxn[M,256] = dY[M,K] W[K,256]        (W^T [256,K] is handed in, so both operands are K-contiguous: "NT")
//     dz = LNbackward(dxn; z, stats, gamma) + d_res ;  dgamma = sum_rows dxn * xhat ;  dbeta = sum_rows dxn
// One workgroup owns 128 token rows and ALL 256 output features (2 x 2 waves of 64 tokens x 128 features), so the
// finished dxn tile never goes to HBM: it is parked in LDS, rounded to T exactly like the stand-alone GEMM's output
// was, and the LayerNorm backward (same arithmetic and row-per-wave mapping as ln_bwd_kernel) reads it from there.
// Replaces a library GEMM (hipBLASLt, 38 us at M = 64,320) + mtmp_ln_bwd (32 us) and their M x 256 round trip.
constexpr int LDXT = 256 + 8;       // staging row (elements)

template <typename T> struct LnBwdGemmArgs {
    const T* dy; const T* wt; const T* z; const float* stats; const float* gamma; const T* d_res; T* dz; float* slab;
    int M, K, ldy, ldz, ldr;
    float eps;
    const int* m_live = nullptr;       // packed stream: rows in use (common.hip.h live_rows)
};

template <typename T>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void gemm_lnbwd_kernel(Grouped<LnBwdGemmArgs<T>> grp) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int seg = grp_find(grp, (int)blockIdx.x);
    const LnBwdGemmArgs<T>& p = grp.seg[seg];
    const int bx = (int)blockIdx.x - grp.first[seg];
    T* sA = reinterpret_cast<T*>(smem_raw);   // [128][LDW]  dY tile
    T* sW = sA + 128 * LDW;                   // [256][LDW]  W^T tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    const int wr = wave & 1, wc = wave >> 1;
    const int m0 = bx * 128;
    const int nk = (p.K + BK - 1) / BK;
    const int M = live_rows(p.M, p.m_live);
    if (m0 >= M) {                            // packed stream, row block past the rows in use: its partial row counts as zeros
        for (int i = tid; i < 512; i += 256) p.slab[(size_t)bx * 512 + i] = 0.f;
        return;
    }
    TileRegs<T> areg, wreg0, wreg1;
    tile_fetch<T>(areg, p.dy, p.ldy, m0, M, 0, tid, p.K);
    tile_fetch<T>(wreg0, p.wt, p.K, 0, 256, 0, tid, p.K);
    tile_fetch<T>(wreg1, p.wt, p.K, 128, 256, 0, tid, p.K);
    f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[rt][nt] = f32x16{0};
    for (int kc = 0; kc < nk; ++kc) {
        __syncthreads();
        tile_commit<T>(sA, areg, tid);
        tile_commit<T>(sW, wreg0, tid);
        tile_commit<T>(sW + 128 * LDW, wreg1, tid);
        __syncthreads();
        if (kc + 1 < nk) {
            tile_fetch<T>(areg, p.dy, p.ldy, m0, M, (kc + 1) * BK, tid, p.K);
            tile_fetch<T>(wreg0, p.wt, p.K, 0, 256, (kc + 1) * BK, tid, p.K);
            tile_fetch<T>(wreg1, p.wt, p.K, 128, 256, (kc + 1) * BK, tid, p.K);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            Frag<T> a[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) a[rt] = frag_load<T>(sA + (64 * wr + 32 * rt + r) * LDW + 16 * c + 8 * half);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const Frag<T> w = frag_load<T>(sW + (128 * wc + 32 * nt + r) * LDW + 16 * c + 8 * half);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) mma<T>(acc[rt][nt], w, a[rt]);
            }
        }
    }
    __syncthreads();                          // every wave is done with sA / sW: the dxn tile takes their place
    T* sX = reinterpret_cast<T*>(smem_raw);   // [128][LDXT]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                store4<T>(sX + (64 * wr + 32 * rt + r) * LDXT + 128 * wc + 32 * nt + 8 * g + 4 * half, acc[rt][nt][4 * g],
                          acc[rt][nt][4 * g + 1], acc[rt][nt][4 * g + 2], acc[rt][nt][4 * g + 3]);
    __syncthreads();
    // LayerNorm backward over the tile's rows: a wave owns 32 consecutive rows, 4 in flight per trip (ln_bwd_kernel)
    const f32x4 gm = ld4f(p.gamma + 4 * lane);
    float part[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    constexpr int RPW = 4;
    const T* res = p.d_res ? p.d_res : p.z;                // no residual: a finite stand-in scaled by 0
    const int ldres = p.d_res ? p.ldr : p.ldz;
    const float rscale = p.d_res ? 1.0f : 0.0f;
    for (int trip = 0; trip < 32 / RPW; ++trip) {
        const int rl0 = wave * 32 + trip * RPW;
        if (m0 + rl0 >= M) break;                        // wave-uniform: whole trips past M do nothing
        f32x4 zv[RPW], dv[RPW], rv[RPW];
        float mu[RPW], rs[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const size_t row = (size_t)min(m0 + rl0 + k, M - 1);
            zv[k] = load4<T>(p.z + row * p.ldz + 4 * lane);
            rv[k] = load4<T>(res + row * ldres + 4 * lane);
            mu[k] = p.stats[2 * row];
            rs[k] = p.stats[2 * row + 1];
            dv[k] = load4<T>(sX + (rl0 + k) * LDXT + 4 * lane);
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const float live = m0 + rl0 + k < M ? 1.0f : 0.0f;   // rows past M (clamped loads) rewrite row M-1 identically
            const float sigma = 1.0f / rs[k] - p.eps;
            float xh[4], g[4], sg = 0.f, sgx = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xh[i] = (zv[k][i] - mu[k]) * rs[k];
                g[i] = dv[k][i] * gm[i];
                sg += g[i];
                sgx += g[i] * xh[i];
                part[0][i] += live * dv[k][i] * xh[i];
                part[1][i] += live * dv[k][i];
            }
            wave_sum2(sg, sgx);
            const float mg = sg * (1.0f / 256);
            const float kx = sgx / (255.0f * sigma);
            // a clamped row recomputes row M-1 from ITS OWN dxn only when its staged row is that row's copy: the A tile
            // replicates row M-1 for rows past M (tile_fetch), so the staged rows past M equal row M-1's
            store4<T>(p.dz + (size_t)min(m0 + rl0 + k, M - 1) * 256 + 4 * lane,
                      (g[0] - mg) * rs[k] - xh[0] * kx + rscale * rv[k][0], (g[1] - mg) * rs[k] - xh[1] * kx + rscale * rv[k][1],
                      (g[2] - mg) * rs[k] - xh[2] * kx + rscale * rv[k][2], (g[3] - mg) * rs[k] - xh[3] * kx + rscale * rv[k][3]);
        }
    }
    __syncthreads();                          // all rows of sX are consumed: its head becomes the partials' scratch
    flush_partials<2>(part, p.slab + (size_t)bx * 512, reinterpret_cast<float*>(smem_raw), lane, wave);
}

template <typename T>
int launch_gemm_lnbwd(int n, LnBwdGemmArgs<T>* segs, float* const* dgamma_dbeta, float* const* ws, hipStream_t st) {
    size_t sm = (size_t)(128 + 256) * LDW * sizeof(T);
    const size_t stage = (size_t)128 * LDXT * sizeof(T);
    if (sm < stage) sm = stage;
    if (sm < 4 * 2 * 256 * sizeof(float)) sm = 4 * 2 * 256 * sizeof(float);
    if (sm > 48 * 1024 && hipFuncSetAttribute((const void*)gemm_lnbwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)sm) != hipSuccess) {
        mtmp_set_error("mtmp_gemm_lnbwd: cannot raise dynamic LDS to %zu", sm);
        return MTMP_ERR_LAUNCH;
    }
    for (int i = 0; i < n; ++i) {
        segs[i].slab = ws[i];
        if (segs[i].K != segs[0].K) {
            mtmp_set_error("mtmp_gemm_lnbwd (grouped): the streams of one launch must agree in K");
            return MTMP_ERR_ARG;
        }
    }
    int nb;
    const Grouped<LnBwdGemmArgs<T>> g = make_group(n, segs, [](const LnBwdGemmArgs<T>& x) { return (x.M + 127) / 128; }, nb);
    hipLaunchKernelGGL(gemm_lnbwd_kernel<T>, dim3(nb), dim3(256), sm, st, g);
    MTMP_CHECK_LAUNCH("mtmp_gemm_lnbwd");
    for (int i = 0; i < n; ++i) {
        if (!dgamma_dbeta || !dgamma_dbeta[i]) continue;      // partials only: the caller reduces them (mtmp_reduce_batch)
        const int nbi = (segs[i].M + 127) / 128;
        launch_slab_reduce(ws[i], nbi, 512, ws[i] + (size_t)nbi * 512, dgamma_dbeta[i], st);
        MTMP_CHECK_LAUNCH("mtmp_gemm_lnbwd(reduce)");
    }
    return MTMP_OK;
}

// partial slabs the launch writes (rows of the [splits][N K + N] workspace) -- a function of the shape only, the deferred
// reductions ask for it through mtmp_gemm_tn_slab_rows.  mode: 0 one token group per workgroup, 1 two token groups (half the
// slabs), 3 the LDS-DMA kernel with 128 x 128 tiles at the split count of mode 1 (bf16, large M; -DMTMP_TN_OLD: never),
// 2 the LDS-DMA kernel with 128 x 256 tiles (-DMTMP_TN_WIDE)
int tn_launch_splits(bool tr, int M, int N, int K, int* mode_out) {
    // two token groups per workgroup / wide tiles once the split count is not what limits the grid
    // Workgroups of the one-per-CU kernels: 192, not one per CU -- these workgroups take a whole CU's LDS, so every CU that holds
    // a workgroup of the image / text streams' concurrent launches delays one of them to a second round; with a quarter of the
    // CUs left free the launch is ~15 % longer alone and the step shorter (9.03-9.06 / 9.25-9.35 ms against 9.11-9.14 / 9.30-9.42
    // with 256, two boxes; 224: 9.08-9.11; 160 and 128: no better than 256).
    constexpr int TN_WGS = 192;
    const bool two = tr && tn_splits(M, N, K, TN_WGS) * 8 * TK <= M;
    int mode = two ? 1 : 0, splits = tn_splits(M, N, K, two ? TN_WGS : (tr ? 512 : 640));
    if (two && K >= 256) mode = 3;                         // the DMA kernel with 128 x 128 tiles, split count of `two`
    if (mode_out) *mode_out = mode;
    return splits;
}
template <typename T>
int launch_gemm_tn(const void* dy, const void* x, float* dw, float* db, float* ws, int M, int N, int K, int ldy, int ldx,
                   const int* m_live, hipStream_t st) {
    constexpr bool TR = sizeof(T) == 2;                    // bf16: transposing-read kernels, eight waves per CU
    int mode;
    const int splits = tn_launch_splits(TR, M, N, K, &mode);
    int rps = (M + splits - 1) / splits;
    rps = (rps + TK - 1) / TK * TK;
    TnArgs<T> a{(const T*)dy, (const T*)x, ws, M, N, K, ldy, ldx, splits, rps};
    a.m_live = m_live;
    // the DMA kernel needs 16-byte aligned rows and 32-bit byte offsets; operands that are not get the register-staged kernel
    // at the same split count
    const bool dma = mode >= 2 && ldy % 8 == 0 && ldx % 8 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)x % 16 == 0 &&
                     (unsigned long long)M * (unsigned)(ldy > ldx ? ldy : ldx) * 2ull < (1ull << 32);
    const bool two = mode == 1 || (mode == 3 && !dma);
    const bool wide = mode == 2;
    const size_t sm = dma ? (wide ? (size_t)TnDma<2>::DNS * TnDma<2>::DSTAGE : (size_t)TnDma<1>::DNS * TnDma<1>::DSTAGE) : TR ? (size_t)(two ? 8 : 4) * TT * LDG * sizeof(bf16) : (size_t)256 * LDX * sizeof(T);
    const void* fn = dma ? (wide ? (const void*)gemm_tn_dma_kernel<2> : (const void*)gemm_tn_dma_kernel<1>)
                         : TR ? (two ? (const void*)gemm_tn_tr_kernel<2> : (const void*)gemm_tn_tr_kernel<1>) : (const void*)gemm_tn_kernel<T>;
    if (sm > 48 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) {
        mtmp_set_error("mtmp_gemm_tn: cannot raise dynamic LDS to %zu", sm);
        return MTMP_ERR_LAUNCH;
    }
    const dim3 grid(splits * (N / 128) * (K / 128)), grid_w(splits * (N / 128) * (K / 256));
    if constexpr (TR) {
        int nwg;
        const Grouped<TnArgs<bf16>> g = make_group(1, &a, [&](const TnArgs<bf16>& x) { return (int)(wide ? grid_w.x : grid.x); }, nwg);
        if (dma && wide) hipLaunchKernelGGL(gemm_tn_dma_kernel<2>, grid_w, dim3(512), sm, st, g);
        else if (dma) hipLaunchKernelGGL(gemm_tn_dma_kernel<1>, grid, dim3(512), sm, st, g);
        else if (two) hipLaunchKernelGGL(gemm_tn_tr_kernel<2>, grid, dim3(512), sm, st, a);
        else     hipLaunchKernelGGL(gemm_tn_tr_kernel<1>, grid, dim3(256), sm, st, a);
    } else {
        hipLaunchKernelGGL(gemm_tn_kernel<T>, grid, dim3(256), sm, st, a);
    }
    MTMP_CHECK_LAUNCH("mtmp_gemm_tn");
    if (!dw) return MTMP_OK;                                  // partials only: the caller reduces them (mtmp_reduce_batch)
    const size_t nk = (size_t)N * K, cols = nk + N;
    ReduceBatch t;                                            // the same kernel (and summation order) as the deferred form
    for (int i = 0; i < RB_MAX; ++i) {
        t.slab[i] = ws; t.a[i] = dw; t.b[i] = db; t.cols[i] = (long long)cols; t.split[i] = (long long)nk; t.rows[i] = splits;
        t.ld[i] = (long long)cols;
    }
    const int rlanes = splits >= 256 ? 64 : splits >= 64 ? 16 : 4;
    const size_t blocks = (cols + 4 * (256 / rlanes) - 1) / (4 * (256 / rlanes));
    hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks), 1), dim3(256), 0, st, t);
    MTMP_CHECK_LAUNCH("mtmp_gemm_tn(reduce)");
    return MTMP_OK;
}

// Grouped weight gradients (bf16, LDS-DMA kernel): the same product of up to three token streams in one launch, partial slabs
// only (the caller reduces them with mtmp_reduce_batch).  The longest stream is split as its single launch would be; the others
// so that a workgroup walks about as many tokens (a workgroup = one 128 x 128 tile of one split, one per CU).
int tn_group_plan(int n, const int* M, int N, int K, int* splits) {
    int big = 0;
    for (int i = 1; i < n; ++i) if (M[i] > M[big]) big = i;
    int mode;
    const int s0 = tn_launch_splits(true, M[big], N, K, &mode);
    if (mode != 3 || N % 128 || K % 128) return 1;            // the longest stream would not take the DMA kernel: no grouped form
    const int target = (M[big] + s0 - 1) / s0;
    for (int i = 0; i < n; ++i) {
        int si = i == big ? s0 : (M[i] + target / 2) / target;
        const int max_s = (M[i] + 4 * TK - 1) / (4 * TK);
        if (si > max_s) si = max_s;
        splits[i] = si < 1 ? 1 : si;
    }
    return 0;
}
int launch_gemm_tn_grouped(int n, TnArgs<bf16>* segs, hipStream_t st) {
    const size_t sm = (size_t)TnDma<1>::DNS * TnDma<1>::DSTAGE;
    if (hipFuncSetAttribute((const void*)gemm_tn_dma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) {
        mtmp_set_error("mtmp_gemm_tn: cannot raise dynamic LDS to %zu", sm);
        return MTMP_ERR_LAUNCH;
    }
    int nwg;
    const Grouped<TnArgs<bf16>> g = make_group(n, segs, [](const TnArgs<bf16>& x) { return x.splits * (x.N / 128) * (x.K / 128); }, nwg);
    hipLaunchKernelGGL(gemm_tn_dma_kernel<1>, dim3(nwg), dim3(512), sm, st, g);
    MTMP_CHECK_LAUNCH("mtmp_gemm_tn_grouped");
    return MTMP_OK;
}

}  // namespace

// Y[M,N] = act(LN(X[M,256]; gamma, beta, eps) W[N,256]^T + bias); xn[M,256] and stats[M,2]
// (mean, 1/(std+eps)) are optional outputs.  Replaces module.py:138-144 + attention.py:68-70
// (relu=0, N=768) and module.py:138-144 + module.py:74-77 (relu=1, N=1024).
static int ln_gemm_entry(int dtype, const void* x, const float* gamma, const float* beta, const void* w, const float* bias,
                         void* y, void* xn, float* stats, int M, int N, int ldx, int ldy, float eps, int relu, float drop_p,
                         unsigned seed, const unsigned* seed_dev, void* signs, void* stream) {
    MTMP_CHECK_ARG(x && gamma && beta && w && y, "mtmp_ln_gemm: null pointer");
    MTMP_CHECK_ARG(M > 0 && N > 0 && N % 32 == 0 && ldx >= 256 && ldx % 8 == 0 && ldy >= N && ldy % 8 == 0,
                   "mtmp_ln_gemm: bad shape M=%d N=%d ldx=%d ldy=%d (K is fixed at 256)", M, N, ldx, ldy);
    MTMP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (double)M * N < 4294967296.0, "mtmp_ln_gemm: bad dropout %f", drop_p);
    MTMP_CHECK_ARG(!signs || (dtype == 1 && relu && N % PanelDma::NP == 0), "mtmp_ln_gemm_signs: bf16, ReLU and N %% 64 == 0 only");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        GemmArgs<float> a{(const float*)x, (const float*)w, bias, nullptr, (float*)y, gamma, beta, (float*)xn, stats,
                          M, N, 256, ldx, ldy, 0, eps, drop_p, seed, seed_dev, nullptr, 1.f, 0, nullptr, 1};
        return launch_ln_gemm<float>(a, relu, st);
    }
    if (dtype == 1) {
        GemmArgs<bf16> a{(const bf16*)x, (const bf16*)w, bias, nullptr, (bf16*)y, gamma, beta, (bf16*)xn, stats,
                         M, N, 256, ldx, ldy, 0, eps, drop_p, seed, seed_dev, nullptr, 1.f, 0, nullptr, 1, (unsigned short*)signs};
        return launch_ln_gemm<bf16>(a, relu, st);
    }
    mtmp_set_error("mtmp_ln_gemm: unknown dtype %d", dtype);
    return MTMP_ERR_ARG;
}
extern "C" int mtmp_ln_gemm(int dtype, const void* x, const float* gamma, const float* beta, const void* w,
                            const float* bias, void* y, void* xn, float* stats, int M, int N, int ldx, int ldy,
                            float eps, int relu, float drop_p, unsigned seed, const unsigned* seed_dev, void* stream) {
    return ln_gemm_entry(dtype, x, gamma, beta, w, bias, y, xn, stats, M, N, ldx, ldy, eps, relu, drop_p, seed, seed_dev, nullptr, stream);
}
// The Q/K/V projection of an encoder layer (module.py:138-144 + attention.py:68-70; w = [Wq; Wk; Wv], N = 768) that also
// writes the key-norm table of the attention forward's bounded body: key_norms[ceil(M / 32)][4] = max ||k_h|| per block of 32
// token rows (mtmp_key_norms_floats(M, 4) floats).  bf16: from the projection's epilogue, no extra pass; fp32: the projection
// followed by mtmp_key_norms (the parity build keeps the round-1 row-panel kernel).
extern "C" int mtmp_key_norms(int dtype, const void* k, float* out, long long rows, int H, int ld, void* stream);
extern "C" int mtmp_ln_gemm_qkv(int dtype, const void* x, const float* gamma, const float* beta, const void* w, const float* bias,
                                void* y, void* xn, float* stats, float* key_norms, int M, int ldx, float eps, void* stream) {
    MTMP_CHECK_ARG(x && gamma && beta && w && y && key_norms, "mtmp_ln_gemm_qkv: null pointer");
    MTMP_CHECK_ARG(M > 0 && ldx >= 256 && ldx % 8 == 0, "mtmp_ln_gemm_qkv: bad shape M=%d ldx=%d", M, ldx);
    if (dtype == 1) {
        GemmArgs<bf16> a{(const bf16*)x, (const bf16*)w, bias, nullptr, (bf16*)y, gamma, beta, (bf16*)xn, stats,
                         M, 768, 256, ldx, 768, 0, eps, 0.f, 0u, nullptr, nullptr, 1.f, 0, nullptr, 1, nullptr, key_norms};
        return launch_ln_gemm_dma(1, &a, 0, 0, (hipStream_t)stream);
    }
    if (int e = ln_gemm_entry(dtype, x, gamma, beta, w, bias, y, xn, stats, M, 768, ldx, 768, eps, 0, 0.f, 0u, nullptr, nullptr, stream)) return e;
    const size_t es = dtype == 0 ? 4 : 2;
    return mtmp_key_norms(dtype, (const char*)y + 256 * es, key_norms, M, 4, 768, stream);
}
// Sign bits of a ReLU projection (bf16 build of the row-panel kernels only): one bit per output, "y > 0", in the kernels'
// private order; mtmp_sign_bits_bytes(M, N) bytes.  mtmp_ln_gemm_signs = mtmp_ln_gemm (relu = 1) that also writes them;
// mtmp_gemm_nt_signs: Y[M,N] = signs ? (A[M,256] W[N,256]^T) * gate_scale : 0 -- the FFN backward's dH = dY W2 through the
// ReLU + dropout of module.py:77-79 without re-reading the M x N hidden activation (its sign is all the backward needs).
extern "C" long long mtmp_sign_bits_bytes(int M, int N) { return (long long)(N / 32) * M * 4; }
extern "C" int mtmp_ln_gemm_signs(int dtype, const void* x, const float* gamma, const float* beta, const void* w,
                                  const float* bias, void* y, void* xn, float* stats, int M, int N, int ldx, int ldy,
                                  float eps, float drop_p, unsigned seed, const unsigned* seed_dev, void* signs, void* stream) {
    MTMP_CHECK_ARG(signs, "mtmp_ln_gemm_signs: null pointer");
    return ln_gemm_entry(dtype, x, gamma, beta, w, bias, y, xn, stats, M, N, ldx, ldy, eps, 1, drop_p, seed, seed_dev, signs, stream);
}
extern "C" int mtmp_gemm_nt_signs(int dtype, const void* a, const void* w, void* y, int M, int N, int lda, int ldy,
                                  const void* signs, float gate_scale, void* stream) {
    MTMP_CHECK_ARG(a && w && y && signs, "mtmp_gemm_nt_signs: null pointer");
    MTMP_CHECK_ARG(dtype == 1, "mtmp_gemm_nt_signs: bf16 only (dtype %d)", dtype);
    MTMP_CHECK_ARG(M > 0 && N > 0 && N % PanelDma::NP == 0 && lda >= 256 && lda % 8 == 0 && ldy >= N && ldy % 8 == 0,
                   "mtmp_gemm_nt_signs: bad shape M=%d N=%d lda=%d ldy=%d (K is fixed at 256)", M, N, lda, ldy);
    GemmArgs<bf16> g{(const bf16*)a, (const bf16*)w, nullptr, nullptr, (bf16*)y, nullptr, nullptr, nullptr, nullptr, M, N, 256, lda, ldy,
                     0, 0.f, 0.f, 0u, nullptr, nullptr, gate_scale, 0, nullptr, 1, (unsigned short*)signs};
    return launch_ln_gemm_dma(1, &g, 0, 1, (hipStream_t)stream);
}

// mtmp_gemm_nt_signs with the backward of a dropout on its A operand folded in (autograd of module.py:78-80 in front of the dH
// product): a' = keep ? a / (1 - p) : 0 with mtmp_dropout_bwd's mask for (seed, seed_dev, p) on the contiguous [M,256] tensor;
// y = signs ? (a' W^T) * gate_scale : 0; a_out [M,256] (may be NULL) receives a' -- the weight-gradient product's operand.
extern "C" int mtmp_gemm_nt_signs_drop(int dtype, const void* a, const void* w, void* y, int M, int N, int lda, int ldy,
                                       const void* signs, float gate_scale, float drop_p, unsigned seed, const unsigned* seed_dev,
                                       void* a_out, void* stream) {
    MTMP_CHECK_ARG(a && w && y && signs, "mtmp_gemm_nt_signs_drop: null pointer");
    MTMP_CHECK_ARG(dtype == 1, "mtmp_gemm_nt_signs_drop: bf16 only (dtype %d)", dtype);
    MTMP_CHECK_ARG(M > 0 && M < (1 << 24) && N > 0 && N % PanelDma::NP == 0 && lda >= 256 && lda % 8 == 0 && ldy >= N && ldy % 8 == 0 &&
                       drop_p >= 0.f && drop_p < 1.f,
                   "mtmp_gemm_nt_signs_drop: bad argument M=%d N=%d lda=%d ldy=%d p=%f (K is fixed at 256)", M, N, lda, ldy, drop_p);
    GemmArgs<bf16> g{(const bf16*)a, (const bf16*)w, nullptr, nullptr, (bf16*)y, nullptr, nullptr, (bf16*)a_out, nullptr, M, N, 256, lda,
                     ldy, 0, 0.f, drop_p, seed, seed_dev, nullptr, gate_scale, 0, nullptr, 1, (unsigned short*)signs};
    return launch_ln_gemm_dma(1, &g, 0, 1, (hipStream_t)stream);
}

// Y[M,N] = drop(act(A[M,K] W[N,K]^T + bias)) (+ R[M,N]).  Replaces module.py:78-80 + encoder.py:32
// (Conv1d(1024,256,1) + drop2 + residual) and is the generic NT projection of the path.
extern "C" int mtmp_gemm_nt_live(int dtype, const void* a, const void* w, const float* bias, const void* res, void* y,
                                 int M, int N, int K, int lda, int ldy, int ldr, int act, float drop_p, unsigned seed,
                                 const unsigned* seed_dev, const void* gate, float gate_scale, const float* row_scale,
                                 int rows_per_scale, const int32_t* rows_live, void* stream);
extern "C" int mtmp_gemm_nt(int dtype, const void* a, const void* w, const float* bias, const void* res, void* y,
                            int M, int N, int K, int lda, int ldy, int ldr, int act, float drop_p, unsigned seed,
                            const unsigned* seed_dev, const void* gate, float gate_scale, const float* row_scale,
                            int rows_per_scale, void* stream) {
    return mtmp_gemm_nt_live(dtype, a, w, bias, res, y, M, N, K, lda, ldy, ldr, act, drop_p, seed, seed_dev, gate, gate_scale,
                             row_scale, rows_per_scale, nullptr, stream);
}
// rows_live (may be NULL): a device word with the rows in use (<= M), see the grouped forms
extern "C" int mtmp_gemm_nt_live(int dtype, const void* a, const void* w, const float* bias, const void* res, void* y,
                                 int M, int N, int K, int lda, int ldy, int ldr, int act, float drop_p, unsigned seed,
                                 const unsigned* seed_dev, const void* gate, float gate_scale, const float* row_scale,
                                 int rows_per_scale, const int32_t* rows_live, void* stream) {
    MTMP_CHECK_ARG(a && w && y, "mtmp_gemm_nt: null pointer");
    MTMP_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 32 == 0 && lda >= K && lda % 8 == 0 && ldy >= N &&
                       ldy % 8 == 0 && (!res || (ldr >= N && ldr % 8 == 0)),
                   "mtmp_gemm_nt: bad shape M=%d N=%d K=%d lda=%d ldy=%d ldr=%d", M, N, K, lda, ldy, ldr);
    MTMP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (double)M * N < 4294967296.0, "mtmp_gemm_nt: bad dropout %f", drop_p);
    MTMP_CHECK_ARG(act >= 0 && act <= 2 && (!row_scale || rows_per_scale > 0), "mtmp_gemm_nt: bad act %d / row_scale", act);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        GemmArgs<float> g{(const float*)a, (const float*)w, bias, (const float*)res, (float*)y, nullptr, nullptr,
                          nullptr, nullptr, M, N, K, lda, ldy, ldr, 0.f, drop_p, seed, seed_dev, (const float*)gate, gate_scale, act, row_scale,
                          rows_per_scale};
        g.m_live = rows_live;
        return launch_gemm_nt<float>(1, &g, 0, st);
    }
    if (dtype == 1) {
        GemmArgs<bf16> g{(const bf16*)a, (const bf16*)w, bias, (const bf16*)res, (bf16*)y, nullptr, nullptr, nullptr,
                         nullptr, M, N, K, lda, ldy, ldr, 0.f, drop_p, seed, seed_dev, (const bf16*)gate, gate_scale, act, row_scale,
                         rows_per_scale};
        g.m_live = rows_live;
        return launch_gemm_nt<bf16>(1, &g, 0, st);
    }
    mtmp_set_error("mtmp_gemm_nt: unknown dtype %d", dtype);
    return MTMP_ERR_ARG;
}

extern "C" int mtmp_gemm_lnbwd_ws_floats(int M) { return ((M + 127) / 128 + RED_GROUPS) * 512; }

// dz[M,256] = LNbackward(dY[M,K] Wt[256,K]^T; z, stats, gamma) (+ d_res);  dgamma_dbeta[512] overwritten.
// Wt = W^T of the projection y = LN(z) W^T (W [K,256] row-major -> Wt [256,K]); ws: mtmp_gemm_lnbwd_ws_floats(M).
// The dX product of attention.py:68-70 / module.py:74-77 and the backward of module.py:138-144 in one launch
// (+ the two-level reduce of the gamma / beta partials).
extern "C" int mtmp_gemm_lnbwd(int dtype, const void* dy, const void* wt, const void* z, int ldz, const float* stats,
                               const float* gamma, const void* d_res, int ldr, void* dz, float* dgamma_dbeta, float* ws,
                               int M, int K, int ldy, float eps, void* stream) {
    MTMP_CHECK_ARG(dy && wt && z && stats && gamma && dz && ws, "mtmp_gemm_lnbwd: null pointer");
    MTMP_CHECK_ARG(M > 0 && K > 0 && K % 8 == 0 && ldy >= K && ldy % 8 == 0 && ldz >= 256 && ldz % 4 == 0 &&
                       (!d_res || (ldr >= 256 && ldr % 4 == 0)), "mtmp_gemm_lnbwd: bad shape M=%d K=%d ldy=%d ldz=%d", M, K, ldy, ldz);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        LnBwdGemmArgs<float> a{(const float*)dy, (const float*)wt, (const float*)z, stats, gamma, (const float*)d_res,
                               (float*)dz, nullptr, M, K, ldy, ldz, ldr, eps};
        return launch_gemm_lnbwd<float>(1, &a, &dgamma_dbeta, &ws, st);
    }
    if (dtype == 1) {
        LnBwdGemmArgs<bf16> a{(const bf16*)dy, (const bf16*)wt, (const bf16*)z, stats, gamma, (const bf16*)d_res,
                              (bf16*)dz, nullptr, M, K, ldy, ldz, ldr, eps};
        return launch_gemm_lnbwd<bf16>(1, &a, &dgamma_dbeta, &ws, st);
    }
    mtmp_set_error("mtmp_gemm_lnbwd: unknown dtype %d", dtype);
    return MTMP_ERR_ARG;
}

// Deferred reductions.  mtmp_gemm_tn with dw == NULL and mtmp_gemm_lnbwd with dgamma_dbeta == NULL leave their partial slabs in
// ws -- [mtmp_gemm_tn_slab_rows][N K + N] and [mtmp_gemm_lnbwd_slab_rows][512] floats -- and mtmp_reduce_batch sums up to 8 such
// slabs in one launch: out_a[i][c] = sum_r slab[i][r][c] for c < split[i], out_b[i][c - split[i]] for the rest (out_b[i] may be
// NULL).  All arrays are HOST arrays of n entries.
extern "C" int mtmp_gemm_tn_slab_rows(int dtype, int M, int N, int K) { return tn_launch_splits(dtype == 1, M, N, K, nullptr); }
extern "C" int mtmp_gemm_lnbwd_slab_rows(int M) { return (M + 127) / 128; }
namespace {
int launch_reduce_batch(const float* const* slab, const int* rows, const long long* cols, const long long* ld, float* const* out_a,
                        const long long* split, float* const* out_b, int n, hipStream_t st, const char* who) {
    ReduceBatch t;
    long long most = 0;
    for (int i = 0; i < RB_MAX; ++i) {
        const int k = i < n ? i : 0;
        const long long ldk = ld ? ld[k] : cols[k];
        MTMP_CHECK_ARG(slab[k] && out_a[k] && rows[k] > 0 && cols[k] > 0 && split[k] >= 0 && split[k] <= cols[k] && ldk >= cols[k],
                       "%s: bad entry %d", who, k);
        t.slab[i] = slab[k]; t.a[i] = out_a[k]; t.b[i] = out_b ? out_b[k] : nullptr; t.cols[i] = cols[k]; t.split[i] = split[k];
        t.rows[i] = rows[k]; t.ld[i] = ldk;
        most = most > cols[k] ? most : cols[k];
    }
    long long blocks = (most + 255) / 256;
    for (int i = 0; i < n; ++i) {                          // (tall slabs take more row lanes and fewer columns per block)
        const int rl = rows[i] >= 256 ? 64 : rows[i] >= 64 ? 16 : 4;
        const long long need = (cols[i] + 4 * (256 / rl) - 1) / (4 * (256 / rl));
        blocks = blocks > need ? blocks : need;
    }
    hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks), n), dim3(256), 0, st, t);
    MTMP_CHECK_LAUNCH(who);
    return MTMP_OK;
}
}  // namespace
extern "C" int mtmp_reduce_batch(const float* const* slab, const int* rows, const long long* cols, float* const* out_a,
                                 const long long* split, float* const* out_b, int n, void* stream) {
    MTMP_CHECK_ARG(slab && rows && cols && out_a && split && out_b && n > 0 && n <= RB_MAX, "mtmp_reduce_batch: bad argument (n=%d)", n);
    return launch_reduce_batch(slab, rows, cols, nullptr, out_a, split, out_b, n, (hipStream_t)stream, "mtmp_reduce_batch");
}
// Column ranges of partial slabs summed straight into separate destinations, up to 12 per launch: out[i][c] = sum_r src[i][r * ld[i]
// + c] for c < cols[i], where src[i] points at the first column of the range inside its slab and ld[i] is the slab's row length.
// The input nodes' backward (mtmp_stream_input_bwd_partials, mtmp_tie_time_embed_bwd_partials) ends in ONE of these launches that
// writes every parameter's slice of the flat gradient -- instead of two reduction levels, a multi-tensor copy and the gradient
// accumulations of shared parameters (11 launches of ~5 us on the step's tail).  All arrays are HOST arrays of n entries.
extern "C" int mtmp_reduce_scatter(const float* const* src, const int* rows, const long long* ld, const long long* cols,
                                   float* const* out, int n, void* stream) {
    MTMP_CHECK_ARG(src && rows && ld && cols && out && n > 0 && n <= RB_MAX, "mtmp_reduce_scatter: bad argument (n=%d)", n);
    return launch_reduce_batch(src, rows, cols, ld, out, cols, nullptr, n, (hipStream_t)stream, "mtmp_reduce_scatter");
}

extern "C" long long mtmp_gemm_tn_ws_floats(int M, int N, int K) {
    return (long long)tn_splits(M, N, K, 640) * ((long long)N * K + N);   // upper bound over both dtypes' split counts
}

// dW[N,K] (fp32) = dY[M,N]^T X[M,K];  db[N] (fp32, optional) = column sums of dY.  N, K multiples of
// 128.  ws: mtmp_gemm_tn_ws_floats(M,N,K) floats.  The weight / bias gradients of the Linear and
// k=1 Conv1d layers of attention.py:60-62 and module.py:74-78.
extern "C" int mtmp_gemm_tn_live(int dtype, const void* dy, const void* x, float* dw, float* db, float* ws, int M, int N,
                                 int K, int ldy, int ldx, const int32_t* rows_live, void* stream);
extern "C" int mtmp_gemm_tn(int dtype, const void* dy, const void* x, float* dw, float* db, float* ws, int M, int N,
                            int K, int ldy, int ldx, void* stream) {
    return mtmp_gemm_tn_live(dtype, dy, x, dw, db, ws, M, N, K, ldy, ldx, nullptr, stream);
}
// rows_live (may be NULL): a device word with the rows in use (<= M), see mtmp_gemm_tn_grouped -- the split count, the
// workspace and the grid stay those of M, every split takes its share of the live rows.
extern "C" int mtmp_gemm_tn_live(int dtype, const void* dy, const void* x, float* dw, float* db, float* ws, int M, int N,
                                 int K, int ldy, int ldx, const int32_t* rows_live, void* stream) {
    MTMP_CHECK_ARG(dy && x && ws && (dw || !db), "mtmp_gemm_tn: null pointer");
    MTMP_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 128 == 0 && K % 128 == 0 && ldy >= N && ldx >= K && ldy % 8 == 0 &&
                       ldx % 8 == 0, "mtmp_gemm_tn: bad shape M=%d N=%d K=%d ldy=%d ldx=%d", M, N, K, ldy, ldx);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) return launch_gemm_tn<float>(dy, x, dw, db, ws, M, N, K, ldy, ldx, rows_live, st);
    if (dtype == 1) return launch_gemm_tn<bf16>(dy, x, dw, db, ws, M, N, K, ldy, ldx, rows_live, st);
    mtmp_set_error("mtmp_gemm_tn: unknown dtype %d", dtype);
    return MTMP_ERR_ARG;
}

namespace {
// g_out[i] = keep(seed, i) ? g_in[i] / (1-p) : 0 -- backward of the epilogue dropout (same mask).
// ------------------------------------------------------------------------------------------------------------------------
// Grouped forms (bf16): the same operation of up to three token streams of one fusion layer in ONE launch (common.hip.h, Grouped).
// All pointer / int arrays are HOST arrays of n entries; scalars are common to the streams.  rows_live (may be NULL, entries may be
// NULL): per stream a DEVICE word with the rows in use this step (<= M[i]) -- the packed vital-sign stream, whose buffers and
// grids are sized for the padded maximum M[i] (mtmp_row_starts); rows past it are neither read nor written.  Same kernels, same results as n
// calls of the single forms.  The parity (fp32) build keeps the single forms.
#define MTMP_CHECK_GROUP(name, n, dtype)                                                                             \
    MTMP_CHECK_ARG((n) >= 1 && (n) <= GRP_MAX && (dtype) == 1, name ": 1..%d streams, bf16 only (n=%d dtype=%d)", GRP_MAX, n, dtype)

extern "C" int mtmp_ln_gemm_qkv_grouped(int dtype, int n, const void* const* x, const float* const* gamma, const float* const* beta,
                                        const void* const* w, const float* const* bias, void* const* y, void* const* xn,
                                        float* const* stats, float* const* key_norms, const int* M, const int* ldx, float eps,
                                        const int32_t* const* rows_live, void* stream) {
    MTMP_CHECK_GROUP("mtmp_ln_gemm_qkv_grouped", n, dtype);
    GemmArgs<bf16> a[GRP_MAX];
    for (int i = 0; i < n; ++i) {
        MTMP_CHECK_ARG(x[i] && gamma[i] && beta[i] && w[i] && y[i] && key_norms[i] && M[i] > 0 && ldx[i] >= 256 && ldx[i] % 8 == 0,
                       "mtmp_ln_gemm_qkv_grouped: bad argument (stream %d)", i);
        a[i] = GemmArgs<bf16>{(const bf16*)x[i], (const bf16*)w[i], bias ? bias[i] : nullptr, nullptr, (bf16*)y[i], gamma[i], beta[i],
                              xn ? (bf16*)xn[i] : nullptr, stats ? stats[i] : nullptr, M[i], 768, 256, ldx[i], 768, 0, eps, 0.f, 0u,
                              nullptr, nullptr, 1.f, 0, nullptr, 1, nullptr, key_norms[i]};
        a[i].m_live = rows_live ? rows_live[i] : nullptr;
    }
    return launch_ln_gemm_dma(n, a, 0, 0, (hipStream_t)stream);
}

// mtmp_ln_gemm_signs (FFN1: LayerNorm + Conv1d(k=1) + ReLU + drop1 + sign bits), N common, one seed per stream
extern "C" int mtmp_ln_gemm_signs_grouped(int dtype, int n, const void* const* x, const float* const* gamma, const float* const* beta,
                                          const void* const* w, const float* const* bias, void* const* y, void* const* xn,
                                          float* const* stats, void* const* signs, const int* M, int N, const int* ldx, float eps,
                                          float drop_p, const unsigned* seeds, const unsigned* seed_dev,
                                          const int32_t* const* rows_live, void* stream) {
    MTMP_CHECK_GROUP("mtmp_ln_gemm_signs_grouped", n, dtype);
    MTMP_CHECK_ARG(N > 0 && N % PanelDma::NP == 0 && drop_p >= 0.f && drop_p < 1.f, "mtmp_ln_gemm_signs_grouped: bad N=%d / dropout %f", N, drop_p);
    GemmArgs<bf16> a[GRP_MAX];
    for (int i = 0; i < n; ++i) {
        MTMP_CHECK_ARG(x[i] && gamma[i] && beta[i] && w[i] && y[i] && signs[i] && M[i] > 0 && ldx[i] >= 256 && ldx[i] % 8 == 0 &&
                           (double)M[i] * N < 4294967296.0, "mtmp_ln_gemm_signs_grouped: bad argument (stream %d)", i);
        a[i] = GemmArgs<bf16>{(const bf16*)x[i], (const bf16*)w[i], bias ? bias[i] : nullptr, nullptr, (bf16*)y[i], gamma[i], beta[i],
                              xn ? (bf16*)xn[i] : nullptr, stats ? stats[i] : nullptr, M[i], N, 256, ldx[i], N, 0, eps, drop_p,
                              seeds ? seeds[i] : 0u, seed_dev, nullptr, 1.f, 0, nullptr, 1, (unsigned short*)signs[i]};
        a[i].m_live = rows_live ? rows_live[i] : nullptr;
    }
    return launch_ln_gemm_dma(n, a, 1, 0, (hipStream_t)stream);
}

// mtmp_gemm_nt_signs_drop (dH = dY W2 through ReLU + drop1, drop2's backward on the operand), N common
extern "C" int mtmp_gemm_nt_signs_drop_grouped(int dtype, int n, const void* const* a_in, const void* const* w, void* const* y,
                                               const int* M, int N, const int* lda, const void* const* signs, float gate_scale,
                                               float drop_p, const unsigned* seeds, const unsigned* seed_dev, void* const* a_out,
                                               const int32_t* const* rows_live, void* stream) {
    MTMP_CHECK_GROUP("mtmp_gemm_nt_signs_drop_grouped", n, dtype);
    MTMP_CHECK_ARG(N > 0 && N % PanelDma::NP == 0 && drop_p >= 0.f && drop_p < 1.f, "mtmp_gemm_nt_signs_drop_grouped: bad N=%d / dropout %f", N, drop_p);
    GemmArgs<bf16> g[GRP_MAX];
    for (int i = 0; i < n; ++i) {
        MTMP_CHECK_ARG(a_in[i] && w[i] && y[i] && signs[i] && M[i] > 0 && M[i] < (1 << 24) && lda[i] >= 256 && lda[i] % 8 == 0,
                       "mtmp_gemm_nt_signs_drop_grouped: bad argument (stream %d)", i);
        g[i] = GemmArgs<bf16>{(const bf16*)a_in[i], (const bf16*)w[i], nullptr, nullptr, (bf16*)y[i], nullptr, nullptr,
                              a_out ? (bf16*)a_out[i] : nullptr, nullptr, M[i], N, 256, lda[i], N, 0, 0.f, drop_p, seeds ? seeds[i] : 0u,
                              seed_dev, nullptr, gate_scale, 0, nullptr, 1, (unsigned short*)signs[i]};
        g[i].m_live = rows_live ? rows_live[i] : nullptr;
    }
    return launch_ln_gemm_dma(n, g, 0, 1, (hipStream_t)stream);
}

// mtmp_gemm_nt (FFN2: y = drop(a w^T + bias) + res), N / K / act common, no gate, no row scale
extern "C" int mtmp_gemm_nt_grouped(int dtype, int n, const void* const* a_in, const void* const* w, const float* const* bias,
                                    const void* const* res, void* const* y, const int* M, int N, int K, const int* lda, const int* ldy,
                                    const int* ldr, int act, float drop_p, const unsigned* seeds, const unsigned* seed_dev,
                                    const int32_t* const* rows_live, void* stream) {
    MTMP_CHECK_GROUP("mtmp_gemm_nt_grouped", n, dtype);
    MTMP_CHECK_ARG(N > 0 && K > 0 && K % 8 == 0 && N % 32 == 0 && act >= 0 && act <= 2 && drop_p >= 0.f && drop_p < 1.f,
                   "mtmp_gemm_nt_grouped: bad shape N=%d K=%d act=%d p=%f", N, K, act, drop_p);
    GemmArgs<bf16> g[GRP_MAX];
    for (int i = 0; i < n; ++i) {
        const bool has_res = res && res[i];
        MTMP_CHECK_ARG(a_in[i] && w[i] && y[i] && M[i] > 0 && lda[i] >= K && lda[i] % 8 == 0 && ldy[i] >= N && ldy[i] % 8 == 0 &&
                           (!has_res || (ldr[i] >= N && ldr[i] % 8 == 0)) && (double)M[i] * N < 4294967296.0,
                       "mtmp_gemm_nt_grouped: bad argument (stream %d)", i);
        g[i] = GemmArgs<bf16>{(const bf16*)a_in[i], (const bf16*)w[i], bias ? bias[i] : nullptr, has_res ? (const bf16*)res[i] : nullptr,
                              (bf16*)y[i], nullptr, nullptr, nullptr, nullptr, M[i], N, K, lda[i], ldy[i], has_res ? ldr[i] : 0, 0.f,
                              drop_p, seeds ? seeds[i] : 0u, seed_dev, nullptr, 1.f, act, nullptr, 1};
        g[i].m_live = rows_live ? rows_live[i] : nullptr;
    }
    return launch_gemm_nt<bf16>(n, g, 0, (hipStream_t)stream);
}

// mtmp_gemm_lnbwd, partial slabs only (ws[i]: mtmp_gemm_lnbwd_ws_floats(M[i]) floats; reduce with mtmp_reduce_batch), K common
extern "C" int mtmp_gemm_lnbwd_grouped(int dtype, int n, const void* const* dy, const void* const* wt, const void* const* z,
                                       const int* ldz, const float* const* stats, const float* const* gamma, const void* const* d_res,
                                       const int* ldr, void* const* dz, float* const* ws, const int* M, int K, const int* ldy, float eps,
                                       const int32_t* const* rows_live, void* stream) {
    MTMP_CHECK_GROUP("mtmp_gemm_lnbwd_grouped", n, dtype);
    MTMP_CHECK_ARG(K > 0 && K % 8 == 0, "mtmp_gemm_lnbwd_grouped: bad K=%d", K);
    LnBwdGemmArgs<bf16> a[GRP_MAX];
    for (int i = 0; i < n; ++i) {
        const bool has_res = d_res && d_res[i];
        MTMP_CHECK_ARG(dy[i] && wt[i] && z[i] && stats[i] && gamma[i] && dz[i] && ws[i] && M[i] > 0 && ldy[i] >= K && ldy[i] % 8 == 0 &&
                           ldz[i] >= 256 && ldz[i] % 4 == 0 && (!has_res || (ldr[i] >= 256 && ldr[i] % 4 == 0)),
                       "mtmp_gemm_lnbwd_grouped: bad argument (stream %d)", i);
        a[i] = LnBwdGemmArgs<bf16>{(const bf16*)dy[i], (const bf16*)wt[i], (const bf16*)z[i], stats[i], gamma[i],
                                   has_res ? (const bf16*)d_res[i] : nullptr, (bf16*)dz[i], nullptr, M[i], K, ldy[i], ldz[i],
                                   has_res ? ldr[i] : 0, eps};
        a[i].m_live = rows_live ? rows_live[i] : nullptr;
    }
    return launch_gemm_lnbwd<bf16>(n, a, nullptr, ws, (hipStream_t)stream);
}

// Weight gradients of up to three streams in one launch (bf16 LDS-DMA kernel), partial slabs only: ws[i] holds
// splits[i] x (N K + N) floats, splits from mtmp_gemm_tn_group_plan (non-zero return: no grouped form for these shapes --
// use mtmp_gemm_tn per stream); reduce with mtmp_reduce_batch (rows = splits[i]).
extern "C" int mtmp_gemm_tn_group_plan(int n, const int* M, int N, int K, int* splits_out) {
    if (n < 1 || n > GRP_MAX || !M || !splits_out || N <= 0 || K <= 0) return 1;
    for (int i = 0; i < n; ++i) if (M[i] <= 0) return 1;
    return tn_group_plan(n, M, N, K, splits_out);
}
extern "C" int mtmp_gemm_tn_grouped(int dtype, int n, const void* const* dy, const void* const* x, float* const* ws, const int* M,
                                    int N, int K, const int* ldy, const int* ldx, const int* splits, const int32_t* const* rows_live,
                                    void* stream) {
    MTMP_CHECK_GROUP("mtmp_gemm_tn_grouped", n, dtype);
    MTMP_CHECK_ARG(N > 0 && K > 0 && N % 128 == 0 && K % 128 == 0, "mtmp_gemm_tn_grouped: N=%d K=%d must be multiples of 128", N, K);
    TnArgs<bf16> a[GRP_MAX];
    for (int i = 0; i < n; ++i) {
        MTMP_CHECK_ARG(dy[i] && x[i] && ws[i] && M[i] > 0 && splits[i] >= 1 && ldy[i] >= N && ldx[i] >= K && ldy[i] % 8 == 0 &&
                           ldx[i] % 8 == 0 && (uintptr_t)dy[i] % 16 == 0 && (uintptr_t)x[i] % 16 == 0 &&
                           (unsigned long long)M[i] * (unsigned)(ldy[i] > ldx[i] ? ldy[i] : ldx[i]) * 2ull < (1ull << 32),
                       "mtmp_gemm_tn_grouped: bad argument (stream %d)", i);
        int rps = (M[i] + splits[i] - 1) / splits[i];
        rps = (rps + TK - 1) / TK * TK;
        a[i] = TnArgs<bf16>{(const bf16*)dy[i], (const bf16*)x[i], ws[i], M[i], N, K, ldy[i], ldx[i], splits[i], rps};
        a[i].m_live = rows_live ? rows_live[i] : nullptr;
    }
    return launch_gemm_tn_grouped(n, a, (hipStream_t)stream);
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const T* gi, T* go, size_t n4, unsigned seed0, const unsigned* seed_dev,
                                                          float p) {
    const unsigned seed = seed0 ^ (seed_dev ? *seed_dev : 0u);
    const unsigned thr = dropout_threshold(p);
    const float sc = 1.0f / (1.0f - p);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 v = load4<T>(gi + 4 * i);
        const unsigned keep = dropout_keep4(seed, (unsigned)i, thr);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (keep >> k) & 1u ? v[k] * sc : 0.f;
        store4<T>(go + 4 * i, v[0], v[1], v[2], v[3]);
    }
}
}  // namespace

// Backward of the dropout applied in the mtmp_gemm_nt / mtmp_ln_gemm epilogue with the same
// (seed, p) on a contiguous [M,N] tensor of n = M*N elements (n % 4 == 0); in place allowed.
extern "C" int mtmp_dropout_bwd(int dtype, const void* g_in, void* g_out, long long n, unsigned seed,
                                const unsigned* seed_dev, float p, void* stream) {
    MTMP_CHECK_ARG(g_in && g_out && n > 0 && n % 4 == 0 && n < 4294967296LL && p >= 0.f && p < 1.f,
                   "mtmp_dropout_bwd: bad argument n=%lld p=%f", n, p);
    const size_t n4 = (size_t)n / 4;
    const int nb = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) hipLaunchKernelGGL(dropout_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)g_in, (float*)g_out, n4, seed, seed_dev, p);
    else if (dtype == 1) hipLaunchKernelGGL(dropout_bwd_kernel<bf16>, dim3(nb), dim3(256), 0, st, (const bf16*)g_in, (bf16*)g_out, n4, seed, seed_dev, p);
    else { mtmp_set_error("mtmp_dropout_bwd: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_dropout_bwd");
    return MTMP_OK;
}

