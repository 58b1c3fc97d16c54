// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels.
//
// Every contraction on the hot path goes through ONE fragment convention so
// that the fp32 "parity" build of a kernel and its bf16 "perf" build share all
// indexing code and differ only in the MFMA instruction:
//
//   Frag<T>   : 8 elements of T per lane = the k-slice  k = 16*c + 8*(lane>>5) + j,
//               j = 0..7, of row/col (lane & 31) of the operand.
//   mma<T>    : acc(32x32 f32) += A(32 x 16) * B(16 x 32)
//               bf16 -> one v_mfma_f32_32x32x16_bf16
//               f32  -> eight v_mfma_f32_32x32x2_f32 (element u of each half-wave
//                       pairs k = 16c+u with k = 16c+8+u) : bit-exact f32 fma chain.
//   acc layout: lane (col = lane&31, half = lane>>5), register t holds
//               row (t&3) + 8*(t>>2) + 4*half   (dtype independent on gfx950).
//   acc -> Frag: registers 8s..8s+7 of an accumulator are, unchanged, the k-step-s
//               fragment of a following product that contracts over the
//               accumulator's ROW index (row 16s + 8(j>>2) + 4*half + (j&3)).
//               Kernels read the FIRST product's A rows through swz23() so that
//               this becomes the contiguous index 16s + 8*half + j.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MTMP_DEV __device__ __forceinline__

template <typename T> struct Frag;
template <> struct Frag<bf16> { bf16x8 v; };
template <> struct Frag<float> { f32x8 v; };

template <typename T> MTMP_DEV Frag<T> frag_zero() {
    Frag<T> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (T)0.0f;
    return f;
}

// 8 contiguous elements (16-byte aligned for bf16, 32-byte for f32) from global or LDS.
template <typename T> MTMP_DEV Frag<T> frag_load(const T* p);
template <> MTMP_DEV Frag<bf16> frag_load<bf16>(const bf16* p) {
    Frag<bf16> f;
    f.v = *reinterpret_cast<const bf16x8*>(p);
    return f;
}
template <> MTMP_DEV Frag<float> frag_load<float>(const float* p) {
    Frag<float> f;
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    f.v = f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return f;
}
template <typename T> MTMP_DEV void frag_store(T* p, const Frag<T>& f);
template <> MTMP_DEV void frag_store<bf16>(bf16* p, const Frag<bf16>& f) {
    *reinterpret_cast<bf16x8*>(p) = f.v;
}
template <> MTMP_DEV void frag_store<float>(float* p, const Frag<float>& f) {
    *reinterpret_cast<f32x4*>(p) = f32x4{f.v[0], f.v[1], f.v[2], f.v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{f.v[4], f.v[5], f.v[6], f.v[7]};
}

// ok ? f : 0, spelled as a bitwise AND so that it can never become a branch around the load that
// produced f.  Guarded loads written as `cond ? load(p) : zero` make hipcc branch around EACH load
// and wait for it (vmcnt(0)) before the next one -- a tile fetch then costs 8 serial memory round
// trips.  Kernels therefore load from a CLAMPED (always valid) address and mask the result.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x8_t __attribute__((ext_vector_type(8)));
MTMP_DEV Frag<bf16> frag_keep(const Frag<bf16>& f, bool ok) {
    const unsigned m = ok ? 0xFFFFFFFFu : 0u;
    Frag<bf16> r;
    r.v = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4_t, f.v) & u32x4_t{m, m, m, m});
    return r;
}
MTMP_DEV Frag<float> frag_keep(const Frag<float>& f, bool ok) {
    const unsigned m = ok ? 0xFFFFFFFFu : 0u;
    Frag<float> r;
    r.v = __builtin_bit_cast(f32x8, __builtin_bit_cast(u32x8_t, f.v) & u32x8_t{m, m, m, m, m, m, m, m});
    return r;
}

// All lanes of the wave agree that nothing needs masking -> skip the ANDs (interior tiles).
MTMP_DEV bool wave_all(bool ok) { return __builtin_amdgcn_ballot_w64(!ok) == 0; }

// acc = A*B (no accumulator input: the MFMA takes the inline constant 0 as C, no zero-fill moves)
template <typename T> MTMP_DEV f32x16 mma0(const Frag<T>& a, const Frag<T>& b);
template <> MTMP_DEV f32x16 mma0<bf16>(const Frag<bf16>& a, const Frag<bf16>& b) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, f32x16{0}, 0, 0, 0);
}
template <> MTMP_DEV f32x16 mma0<float>(const Frag<float>& a, const Frag<float>& b) {
    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[0], b.v[0], f32x16{0}, 0, 0, 0);
#pragma unroll
    for (int u = 1; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[u], b.v[u], acc, 0, 0, 0);
    return acc;
}
// acc = A*B + C with C left untouched (the destination is a different register block)
template <typename T> MTMP_DEV f32x16 mma_c(const Frag<T>& a, const Frag<T>& b, const f32x16& c);
template <> MTMP_DEV f32x16 mma_c<bf16>(const Frag<bf16>& a, const Frag<bf16>& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
}
template <> MTMP_DEV f32x16 mma_c<float>(const Frag<float>& a, const Frag<float>& b, const f32x16& c) {
    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[0], b.v[0], c, 0, 0, 0);
#pragma unroll
    for (int u = 1; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[u], b.v[u], acc, 0, 0, 0);
    return acc;
}
// max of three as ONE v_max3_f32 (fmaxf() on MFMA outputs costs an extra canonicalising v_max each).
// CAUTION: inline asm is invisible to the compiler's hazard recogniser -- never let it be the first reader
// of an MFMA (or permlane-swap) result; see the forward attention kernel for the guard pattern.
MTMP_DEV float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <typename T> MTMP_DEV void mma(f32x16& acc, const Frag<T>& a, const Frag<T>& b);
template <> MTMP_DEV void mma<bf16>(f32x16& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
template <> MTMP_DEV void mma<float>(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[u], b.v[u], acc, 0, 0, 0);
}

// accumulator registers 8s..8s+7 -> operand fragment of k-step s (see header comment)
template <typename T> MTMP_DEV Frag<T> frag_from_acc(const f32x16& x, int s);
template <> MTMP_DEV Frag<float> frag_from_acc<float>(const f32x16& x, int s) {
    Frag<float> f;
    if (s == 0) f.v = f32x8{x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
    else        f.v = f32x8{x[8], x[9], x[10], x[11], x[12], x[13], x[14], x[15]};
    return f;
}
template <> MTMP_DEV Frag<bf16> frag_from_acc<bf16>(const f32x16& x, int s) {
    Frag<bf16> f;
    f32x8 y = (s == 0) ? f32x8{x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]}
                       : f32x8{x[8], x[9], x[10], x[11], x[12], x[13], x[14], x[15]};
    f.v = __builtin_convertvector(y, bf16x8);     // v_cvt_pk_bf16_f32 (RNE, NaN preserving)
    return f;
}

// swap bits 2 and 3 of a 5-bit index: the A-row permutation described above.
MTMP_DEV int swz23(int i) { return (i & 0x13) | ((i & 4) << 1) | ((i & 8) >> 1); }
// logical (contiguous) index of accumulator register t in lane-half `half` when the
// first product's A rows were read through swz23: 16*(t>>3) + 8*half + (t&7).
MTMP_DEV int acc_row_swz(int t, int half) { return ((t >> 3) << 4) + (half << 3) + (t & 7); }
// natural accumulator row of register t.
MTMP_DEV int acc_row(int t, int half) { return (t & 3) + ((t >> 2) << 3) + (half << 2); }

MTMP_DEV float to_f32(float x) { return x; }
MTMP_DEV float to_f32(bf16 x) { return (float)x; }
template <typename T> MTMP_DEV T from_f32(float x) { return (T)x; }
template <typename T> MTMP_DEV float round_as(float x) { return to_f32((T)x); }

// store 4 consecutive outputs (one accumulator register group) as T
template <typename T> MTMP_DEV void store4(T* p, float a, float b, float c, float d);
template <> MTMP_DEV void store4<float>(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}
template <> MTMP_DEV void store4<bf16>(bf16* p, float a, float b, float c, float d) {
    *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(f32x4{a, b, c, d}, bf16x4);
}
template <typename T> MTMP_DEV f32x4 load4(const T* p);
template <> MTMP_DEV f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> MTMP_DEV f32x4 load4<bf16>(const bf16* p) {
    return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), f32x4);
}

// Wave-wide reductions on the DPP path (round 4).  __shfl_xor lowers to ds_bpermute_b32 -- a round trip through the LDS crossbar,
// ~100+ cycles each, six of them in a dependent chain per reduction: the one-wave-per-row kernels on the step's tail (TIE /
// stream-input backward: a few waves per CU, nothing to hide the chain behind) spent most of their time there.  Here a 16-lane
// row is reduced by four rotate-within-row DPP operands fused into the adds (row_ror:8/4/2/1: every lane of the row ends with
// the row's result), and the four rows meet through v_readlane: ~12 plain vector instructions, no LDS.
// PRECONDITION of dpp_row / wave_sum / wave_max: the whole wave is active (EXEC all ones).  All call sites are wave-uniform.
// (`old` = the lane's own value: a lane without a source lane would keep x -- neutral for max, but it would double-count in a
// sum, and the v_readlane of lanes 0 / 16 / 32 / 48 below reads whatever an inactive lane's register holds: do not call these
// under divergence.)
template <int CTRL> MTMP_DEV float dpp_row(float x) {
    const int xi = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(xi, xi, CTRL, 0xf, 0xf, false));
}
MTMP_DEV float lane_bcast(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
MTMP_DEV float wave_sum(float v) {
    v += dpp_row<0x128>(v);
    v += dpp_row<0x124>(v);
    v += dpp_row<0x122>(v);
    v += dpp_row<0x121>(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
MTMP_DEV float wave_max(float v) {
    v = fmaxf(v, dpp_row<0x128>(v));
    v = fmaxf(v, dpp_row<0x124>(v));
    v = fmaxf(v, dpp_row<0x122>(v));
    v = fmaxf(v, dpp_row<0x121>(v));
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}
// max / sum over the two half-waves (lane i <-> lane i + 32) without the LDS crossbar: v_permlane32_swap
MTMP_DEV float half_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
MTMP_DEV float half_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
MTMP_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// erf, Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7), branch free: the libm erff expands into
// a multi-way branch per element inside GEMM epilogues.
MTMP_DEV float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = 1.0f / fmaf(0.3275911f, ax, 1.0f);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float y = 1.0f - poly * __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    return copysignf(y, x);
}

// GELU of the frozen image encoder's MLP (swin_transformer.py:439, nn.GELU = x * Phi(x)).
//   fp32 (parity build): exact-erf form through erf_as.
//   bf16 (perf build):   x * sigmoid(2z), z = sqrt(2/pi) (x + 0.044715 x^3) -- the tanh form written as a logistic:
//                        7 vector instructions instead of ~20 (the erf epilogue was the bottleneck of the fc1 GEMMs:
//                        77 M activations per stage-1 call at ~90 issue cycles per 4); |error| <= 5e-4 absolute,
//                        below one bf16 ulp of the result wherever it is largest.
template <typename T> MTMP_DEV float gelu(float x);
template <> MTMP_DEV float gelu<float>(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }
template <> MTMP_DEV float gelu<bf16>(float x) {
    const float u = x * fmaf(x * x, 0.0713548163f * 1.4426950408889634f, 1.5957691216f * 1.4426950408889634f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u));
}

// Counter-based dropout mask: element `idx` of a call seeded with `seed` is kept iff
// a 16-bit field of fmix32(group * golden ^ seed) >= p * 2^16.  Stateless, so the backward regenerates the
// same mask from (seed, idx) instead of storing it.
MTMP_DEV unsigned dropout_threshold(float p) { return p <= 0.f ? 0u : (unsigned)((double)p * 65536.0 + 0.5); }
// keep-decisions for the 4 consecutive elements 4g..4g+3 of a call seeded with `seed` (bit i = keep
// element 4g+i): ONE murmur finaliser per group + one extra mixing round; each decision compares a
// 16-bit field with thr = p * 2^16 (the per-element hash of the first version cost more VALU than the
// MFMAs of the projection it sat behind).
MTMP_DEV unsigned dropout_keep4(unsigned seed, unsigned g, unsigned thr) {
    unsigned x = (g * 0x9E3779B1u) ^ seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    unsigned y = x * 0x27D4EB2Fu; y ^= y >> 15;
    return ((x & 0xFFFFu) >= thr ? 1u : 0u) | ((x >> 16) >= thr ? 2u : 0u) | ((y & 0xFFFFu) >= thr ? 4u : 0u) |
           ((y >> 16) >= thr ? 8u : 0u);
}

// The same four decisions as 16-bit fields (element 4g+i is kept iff f[i] >= thr): a caller that compares the fields itself
// feeds the compare result straight into its select instead of building the 4-bit mask and testing it again
// (4 instructions per element less in the GEMM epilogues).
MTMP_DEV void dropout_fields4(unsigned seed, unsigned g, unsigned (&f)[4]) {
    unsigned x = (g * 0x9E3779B1u) ^ seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    unsigned y = x * 0x27D4EB2Fu; y ^= y >> 15;
    f[0] = x & 0xFFFFu; f[1] = x >> 16; f[2] = y & 0xFFFFu; f[3] = y >> 16;
}
// ReLU as ONE integer max (fmaxf() / fmed3 on an MFMA output cost an extra canonicalising v_max each): negative floats
// (and -0.0) are negative integers, everything else passes unchanged (NaNs with the sign bit clear stay NaNs).
MTMP_DEV float relu1(float v) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0)); }

// XCD-aware bijective remap of a 1-D block id: blocks that share an XCD (id % 8)
// get a contiguous chunk of the work list, so neighbours share that XCD's L2.
MTMP_DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ---- grouped launches: ONE grid over the row blocks / tiles of up to three token streams (vital signs, image, text).
// A fusion layer runs the same kernel on three streams of very different lengths (1005 / 54 / 133 tokens); as three launches
// on three HIP streams the small ones hold whole-CU workgroup slots of the big one for 40-90 us each (round 2: every
// vital-sign-stream kernel ran 15-40 % slower inside the step than alone).  As one launch their blocks simply follow the big
// stream's in the same grid.  first[i] = first (XCD-remapped) block of segment i, first[n..GRP_MAX] = the grid size.
constexpr int GRP_MAX = 3;
template <typename A> struct Grouped { A seg[GRP_MAX]; int first[GRP_MAX + 1]; };
template <typename A> MTMP_DEV int grp_find(const Grouped<A>& g, int w) { return (w >= g.first[1] ? 1 : 0) + (w >= g.first[2] ? 1 : 0); }

// ---- streaming-kernel helpers shared by elementwise.hip and the fused dX + LayerNorm-backward GEMM (d_model = 256) ----
MTMP_DEV f32x4 ld4f(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// sum two values across the wave at once
MTMP_DEV void wave_sum2(float& a, float& b) {
    a = wave_sum(a);
    b = wave_sum(b);
}

// block partials: every wave adds its per-lane accumulators acc[NV][4] into LDS, block writes one slab row
template <int NV>
MTMP_DEV void flush_partials(float (&acc)[NV][4], float* slab_row, float* lds, int lane, int wave) {
    // lds: [4 waves][NV*256]
#pragma unroll
    for (int v = 0; v < NV; ++v)
        *reinterpret_cast<f32x4*>(lds + (wave * NV + v) * 256 + 4 * lane) = f32x4{acc[v][0], acc[v][1], acc[v][2], acc[v][3]};
    __syncthreads();
    for (int i = threadIdx.x; i < NV * 256; i += 256)
        slab_row[i] = lds[i] + lds[NV * 256 + i] + lds[2 * NV * 256 + i] + lds[3 * NV * 256 + i];
}

// Row count of a launch on the PACKED vital-sign stream (ops.FusionStackFn, cfg["row_start"]): buffers and grids are sized for the
// padded maximum M -- static, so a captured hipGraph replays -- and the rows in use this step are read from device memory
// (mtmp_row_starts writes them); workgroups whose first row lies past them return at once.
MTMP_DEV int live_rows(int M, const int* m_live) { return m_live ? min(M, *m_live) : M; }

constexpr int RED_GROUPS = 64;            // most row groups of the first level of launch_slab_reduce (sizes its ws tail)
// two-level column sum [rows][cols] -> out[cols] (elementwise.hip); ws_tail: RED_GROUPS * cols floats
void launch_slab_reduce(const float* slab, int rows, int cols, float* ws_tail, float* out, hipStream_t st);

// ---- host side ----
#define MTMP_OK 0
#define MTMP_ERR_ARG 1
#define MTMP_ERR_LAUNCH 2
void mtmp_set_error(const char* fmt, ...);
#define MTMP_CHECK_ARG(cond, ...)                         \
    do {                                                  \
        if (!(cond)) {                                    \
            mtmp_set_error(__VA_ARGS__);                  \
            return MTMP_ERR_ARG;                          \
        }                                                 \
    } while (0)
#define MTMP_CHECK_LAUNCH(name)                                                  \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            mtmp_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return MTMP_ERR_LAUNCH;                                              \
        }                                                                        \
    } while (0)
