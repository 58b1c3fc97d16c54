// MFMA projection kernels of the encoder layer (SURVEY K5+K6, K8) for gfx950.
//
//  mtmp_ln_gemm : Y = act( LN(X) W^T + b ),  X [M,256]  -- the custom LayerNorm of
//      builder/models/src/transformer/module.py:138-144 (unbiased std, eps added to the
//      std) fused as the prologue of
//        * the Q/K/V projections, attention.py:60-62,68-70 (W = [Wq;Wk;Wv], N = 768), and
//        * the first position-wise FFN conv + ReLU, module.py:74-77 (N = 1024).
//      K = d_model = 256 is the whole row, so a wave keeps its 32 normalised rows as MFMA
//      A-fragments in registers (two lanes share a row: statistics need one cross-lane add)
//      and only the weight tiles go through LDS.  Also writes LN(X) (the backward's GEMM
//      operand) and the per-row (mean, 1/(std+eps)).
//  mtmp_gemm_nt : Y = act( A W^T + b ) (+ R) for any K % 64 == 0 -- second FFN conv with the
//      residual add of encoder.py:32 (K = 1024, N = 256).
//
// Both are "NT": activations and weights are contiguous along the contraction index, which
// is exactly the MFMA fragment shape (common.cuh).  Weights arrive in the compute dtype
// (bf16 shadow copy or fp32 master), gamma/beta/bias always fp32.
#include "common.cuh"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, LDW = BK + 8;

template <typename T> struct GemmArgs {
    const T* a; const T* w; const float* bias; const T* res; T* y;
    const float* gamma; const float* beta; T* xn; float* stats;
    int M, N, K, lda, ldy, ldr;
    float eps;
    float drop_p;        // nn.Dropout probability applied after act (0 = off), module.py:77-79
    unsigned seed;       // per-call seed of the counter-based mask (common.cuh: dropout_keep)
};

// 128 x 64 tile of a row-major matrix -> registers (4 x 16 B per thread); rows >= limit give zeros.
template <typename T>
MTMP_DEV void tile_fetch(Frag<T> (&reg)[4], const T* src, int ld, int row0, int limit, int k0, int tid) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int row = row0 + (tid >> 3) + 32 * ps;
        reg[ps] = (row < limit) ? frag_load<T>(src + (size_t)row * ld + k0 + (tid & 7) * 8) : frag_zero<T>();
    }
}
template <typename T> MTMP_DEV void tile_commit(T* dst, const Frag<T> (&reg)[4], int tid) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) frag_store<T>(dst + ((tid >> 3) + 32 * ps) * LDW + (tid & 7) * 8, reg[ps]);
}

template <typename T, bool RELU>
MTMP_DEV void epilogue(const f32x16 (&acc)[4], const GemmArgs<T>& p, int row_base, int n0, int r, int half) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = n0 + 32 * nt + r;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
        const unsigned thr = dropout_threshold(p.drop_p);
        const float keep_scale = 1.0f / (1.0f - p.drop_p);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int row = row_base + acc_row(t, half);
            if (row < p.M) {
                float v = acc[nt][t] + bv;
                if (RELU) v = fmaxf(v, 0.f);
                if (p.drop_p > 0.f)
                    v = dropout_keep(p.seed, (unsigned)row * (unsigned)p.N + (unsigned)col, thr) ? v * keep_scale : 0.f;
                if (p.res) v = round_as<T>(v) + to_f32(p.res[(size_t)row * p.ldr + col]);
                p.y[(size_t)row * p.ldy + col] = from_f32<T>(v);
            }
        }
    }
}

// ---------------------------------------------------------------------------
template <typename T, bool RELU>
__global__ __launch_bounds__(256, 2) void ln_gemm_kernel(GemmArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sW = reinterpret_cast<T*>(smem_raw);                      // [BN][LDW]
    float* sG = reinterpret_cast<float*>(sW + BN * LDW);         // gamma[256], beta[256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    const int row_base = blockIdx.x * BM + wave * 32;
    const int row = row_base + r;
    sG[tid] = p.gamma[tid];
    sG[256 + tid] = p.beta[tid];
    // ---- LayerNorm prologue, in registers: lane (r, half) holds k = 16c + 8*half + j of row r
    Frag<T> af[16];
    float s1 = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        af[c] = (row < p.M) ? frag_load<T>(p.a + (size_t)row * p.lda + 16 * c + 8 * half) : frag_zero<T>();
#pragma unroll
        for (int j = 0; j < 8; ++j) s1 += to_f32(af[c].v[j]);
    }
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * (1.0f / 256.0f);
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = to_f32(af[c].v[j]) - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 32, 64);
    const float sigma = sqrtf(s2 * (1.0f / 255.0f));             // torch.std: Bessel-corrected
    const float rs = 1.0f / (sigma + p.eps);
    __syncthreads();                                             // sG ready
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int k = 16 * c + 8 * half;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            af[c].v[j] = from_f32<T>(fmaf(sG[k + j], (to_f32(af[c].v[j]) - mean) * rs, sG[256 + k + j]));
        if (p.xn && row < p.M) frag_store<T>(p.xn + (size_t)row * 256 + k, af[c]);
    }
    if (p.stats && half == 0 && row < p.M) {
        p.stats[2 * (size_t)row] = mean;
        p.stats[2 * (size_t)row + 1] = rs;
    }
    // ---- Y tiles: for each 128-column block, 4 k-chunks of 64
    const int nsteps = ((p.N + BN - 1) / BN) * 4;
    Frag<T> wreg[4];
    tile_fetch<T>(wreg, p.w, 256, 0, p.N, 0, tid);
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    for (int step = 0; step < nsteps; ++step) {
        const int n0 = (step >> 2) * BN, kc = step & 3;
        __syncthreads();
        tile_commit<T>(sW, wreg, tid);
        __syncthreads();
        if (step + 1 < nsteps) tile_fetch<T>(wreg, p.w, 256, ((step + 1) >> 2) * BN, p.N, ((step + 1) & 3) * BK, tid);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                mma<T>(acc[nt], af[4 * kc + c], frag_load<T>(sW + (32 * nt + r) * LDW + 16 * c + 8 * half));
        if (kc == 3) {
            epilogue<T, RELU>(acc, p, row_base, n0, r, half);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x16{0};
        }
    }
}

// ---------------------------------------------------------------------------
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs<T> p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sA = reinterpret_cast<T*>(smem_raw);   // [BM][LDW]
    T* sW = sA + BM * LDW;                    // [BN][LDW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    const int ntn = (p.N + BN - 1) / BN;
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (w / ntn) * BM, n0 = (w % ntn) * BN;
    const int nk = p.K / BK;
    Frag<T> areg[4], wreg[4];
    tile_fetch<T>(areg, p.a, p.lda, m0, p.M, 0, tid);
    tile_fetch<T>(wreg, p.w, p.K, n0, p.N, 0, tid);
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    for (int kc = 0; kc < nk; ++kc) {
        __syncthreads();
        tile_commit<T>(sA, areg, tid);
        tile_commit<T>(sW, wreg, tid);
        __syncthreads();
        if (kc + 1 < nk) {
            tile_fetch<T>(areg, p.a, p.lda, m0, p.M, (kc + 1) * BK, tid);
            tile_fetch<T>(wreg, p.w, p.K, n0, p.N, (kc + 1) * BK, tid);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const Frag<T> a = frag_load<T>(sA + (32 * wave + r) * LDW + 16 * c + 8 * half);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                mma<T>(acc[nt], a, frag_load<T>(sW + (32 * nt + r) * LDW + 16 * c + 8 * half));
        }
    }
    epilogue<T, RELU>(acc, p, m0 + 32 * wave, n0, r, half);
}

template <typename T>
int launch_ln_gemm(GemmArgs<T> a, int relu, hipStream_t st) {
    const size_t sm = (size_t)BN * LDW * sizeof(T) + 512 * sizeof(float);
    dim3 grid((a.M + BM - 1) / BM);
    if (relu) hipLaunchKernelGGL((ln_gemm_kernel<T, true>), grid, dim3(256), sm, st, a);
    else      hipLaunchKernelGGL((ln_gemm_kernel<T, false>), grid, dim3(256), sm, st, a);
    MTMP_CHECK_LAUNCH("mtmp_ln_gemm");
    return MTMP_OK;
}
template <typename T>
int launch_gemm_nt(GemmArgs<T> a, int relu, hipStream_t st) {
    const size_t sm = (size_t)(BM + BN) * LDW * sizeof(T);
    if (sm > 48 * 1024) {
        const void* f = relu ? (const void*)gemm_nt_kernel<T, true> : (const void*)gemm_nt_kernel<T, false>;
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) {
            mtmp_set_error("mtmp_gemm_nt: cannot raise dynamic LDS to %zu", sm);
            return MTMP_ERR_LAUNCH;
        }
    }
    dim3 grid(((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN));
    if (relu) hipLaunchKernelGGL((gemm_nt_kernel<T, true>), grid, dim3(256), sm, st, a);
    else      hipLaunchKernelGGL((gemm_nt_kernel<T, false>), grid, dim3(256), sm, st, a);
    MTMP_CHECK_LAUNCH("mtmp_gemm_nt");
    return MTMP_OK;
}

}  // namespace

// Y[M,N] = act(LN(X[M,256]; gamma, beta, eps) W[N,256]^T + bias); xn[M,256] and stats[M,2]
// (mean, 1/(std+eps)) are optional outputs.  Replaces module.py:138-144 + attention.py:68-70
// (relu=0, N=768) and module.py:138-144 + module.py:74-77 (relu=1, N=1024).
extern "C" int mtmp_ln_gemm(int dtype, const void* x, const float* gamma, const float* beta, const void* w,
                            const float* bias, void* y, void* xn, float* stats, int M, int N, int ldx, int ldy,
                            float eps, int relu, float drop_p, unsigned seed, void* stream) {
    MTMP_CHECK_ARG(x && gamma && beta && w && y, "mtmp_ln_gemm: null pointer");
    MTMP_CHECK_ARG(M > 0 && N > 0 && N % 32 == 0 && ldx >= 256 && ldx % 8 == 0 && ldy >= N,
                   "mtmp_ln_gemm: bad shape M=%d N=%d ldx=%d ldy=%d (K is fixed at 256)", M, N, ldx, ldy);
    MTMP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (double)M * N < 4294967296.0, "mtmp_ln_gemm: bad dropout %f", drop_p);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        GemmArgs<float> a{(const float*)x, (const float*)w, bias, nullptr, (float*)y, gamma, beta, (float*)xn, stats,
                          M, N, 256, ldx, ldy, 0, eps, drop_p, seed};
        return launch_ln_gemm<float>(a, relu, st);
    }
    if (dtype == 1) {
        GemmArgs<bf16> a{(const bf16*)x, (const bf16*)w, bias, nullptr, (bf16*)y, gamma, beta, (bf16*)xn, stats,
                         M, N, 256, ldx, ldy, 0, eps, drop_p, seed};
        return launch_ln_gemm<bf16>(a, relu, st);
    }
    mtmp_set_error("mtmp_ln_gemm: unknown dtype %d", dtype);
    return MTMP_ERR_ARG;
}

// Y[M,N] = act(A[M,K] W[N,K]^T + bias) (+ R[M,N]).  Replaces module.py:78 + encoder.py:32
// (Conv1d(1024,256,1) + residual) and is the generic NT projection of the path.
extern "C" int mtmp_gemm_nt(int dtype, const void* a, const void* w, const float* bias, const void* res, void* y,
                            int M, int N, int K, int lda, int ldy, int ldr, int relu, float drop_p, unsigned seed,
                            void* stream) {
    MTMP_CHECK_ARG(a && w && y, "mtmp_gemm_nt: null pointer");
    MTMP_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 32 == 0 && lda >= K && lda % 8 == 0 && ldy >= N &&
                       (!res || ldr >= N),
                   "mtmp_gemm_nt: bad shape M=%d N=%d K=%d lda=%d ldy=%d ldr=%d", M, N, K, lda, ldy, ldr);
    MTMP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (double)M * N < 4294967296.0, "mtmp_gemm_nt: bad dropout %f", drop_p);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) {
        GemmArgs<float> g{(const float*)a, (const float*)w, bias, (const float*)res, (float*)y, nullptr, nullptr,
                          nullptr, nullptr, M, N, K, lda, ldy, ldr, 0.f, drop_p, seed};
        return launch_gemm_nt<float>(g, relu, st);
    }
    if (dtype == 1) {
        GemmArgs<bf16> g{(const bf16*)a, (const bf16*)w, bias, (const bf16*)res, (bf16*)y, nullptr, nullptr, nullptr,
                         nullptr, M, N, K, lda, ldy, ldr, 0.f, drop_p, seed};
        return launch_gemm_nt<bf16>(g, relu, st);
    }
    mtmp_set_error("mtmp_gemm_nt: unknown dtype %d", dtype);
    return MTMP_ERR_ARG;
}

namespace {
// g_out[i] = keep(seed, i) ? g_in[i] / (1-p) : 0 -- backward of the epilogue dropout (same mask).
template <typename T>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const T* gi, T* go, size_t n4, unsigned seed, float p) {
    const unsigned thr = dropout_threshold(p);
    const float sc = 1.0f / (1.0f - p);
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 v = load4<T>(gi + 4 * i);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = dropout_keep(seed, (unsigned)(4 * i + k), thr) ? v[k] * sc : 0.f;
        store4<T>(go + 4 * i, v[0], v[1], v[2], v[3]);
    }
}
}  // namespace

// Backward of the dropout applied in the mtmp_gemm_nt / mtmp_ln_gemm epilogue with the same
// (seed, p) on a contiguous [M,N] tensor of n = M*N elements (n % 4 == 0); in place allowed.
extern "C" int mtmp_dropout_bwd(int dtype, const void* g_in, void* g_out, long long n, unsigned seed, float p,
                                void* stream) {
    MTMP_CHECK_ARG(g_in && g_out && n > 0 && n % 4 == 0 && n < 4294967296LL && p >= 0.f && p < 1.f,
                   "mtmp_dropout_bwd: bad argument n=%lld p=%f", n, p);
    const size_t n4 = (size_t)n / 4;
    const int nb = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) hipLaunchKernelGGL(dropout_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)g_in, (float*)g_out, n4, seed, p);
    else if (dtype == 1) hipLaunchKernelGGL(dropout_bwd_kernel<bf16>, dim3(nb), dim3(256), 0, st, (const bf16*)g_in, (bf16*)g_out, n4, seed, p);
    else { mtmp_set_error("mtmp_dropout_bwd: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_dropout_bwd");
    return MTMP_OK;
}
