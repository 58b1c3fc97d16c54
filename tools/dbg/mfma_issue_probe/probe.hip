// How many cycles does one wave per SIMD need per v_mfma_f32_32x32x16_bf16 when vector / LDS instructions stand between the MFMAs?
// (diagnostic for the 64-row attention backward kernels; build: hipcc --offload-arch=gfx950 -O3 probe.hip -o probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF_V(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b))
#define MF_VA(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b))
#define MF_VC(d, a, b, c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c))
#define MF_A(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b))
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define MF16_V(d, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b))
#define V_ADD(x, y) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define V_EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define V_MUL(x, y) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define V_CVT(r, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y))

// V: 0 independent VGPR-dest MFMAs only | 1 + exp, mul, cvt per gap | 2 chains of 4 dependent VGPR-dest MFMAs only | 3 + vector
// 4 independent AGPR-dest only | 5 AGPR-dest + vector | 6 = 1 + two ds_read_b128 per gap | 7 = 3 + two ds_read_b128
// 8 the backward's slot: 8 chained VGPR-dest (B from AGPR) + 8 AGPR-dest, vector on the previous results
// 9 = 8 without the vector work | 10 = 8 with two ds_read_b128 per gap | 11 = 1 with exp only | 12 = 1 with mul + cvt only
// 13 = 3 with D != C first (MF_VC) like the kernel
template <int V> __global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.001f * i;
    __syncthreads();
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
    bf16x8 ba;
    asm volatile("" : "=a"(ba) : "0"(b));
    f32x16 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0}, e0 = {0}, e1 = {0}, e2 = {0}, e3 = {0}, c0 = {0};
    f32x16 x = {0}, y = {0};
    for (int t = 0; t < 16; ++t) { x[t] = 0.001f * (lane + t); y[t] = 1.0f + 0.001f * t; c0[t] = 0.5f; }
    unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 l0 = {0}, l1 = {0};
    const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + lane;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (V == 0 || V == 1 || V == 6 || V == 11 || V == 12) {
                if ((i & 3) == 0) MF_V(d0, a, b); else if ((i & 3) == 1) MF_V(d1, a, b); else if ((i & 3) == 2) MF_V(d2, a, b); else MF_V(d3, a, b);
            } else if (V == 2 || V == 3 || V == 7) {
                if ((i >> 2) == 0) MF_V(d0, a, b); else if ((i >> 2) == 1) MF_V(d1, a, b); else if ((i >> 2) == 2) MF_V(d2, a, b); else MF_V(d3, a, b);
            } else if (V == 13) {
                if (i == 0) MF_VC(d0, a, b, c0); else if (i < 4) MF_V(d0, a, b); else if (i == 4) MF_VC(d1, a, b, c0); else if (i < 8) MF_V(d1, a, b);
                else if (i == 8) MF_VC(d2, a, b, c0); else if (i < 12) MF_V(d2, a, b); else if (i == 12) MF_VC(d3, a, b, c0); else MF_V(d3, a, b);
            } else if (V == 4 || V == 5) {
                if ((i & 3) == 0) MF_A(e0, a, b); else if ((i & 3) == 1) MF_A(e1, a, b); else if ((i & 3) == 2) MF_A(e2, a, b); else MF_A(e3, a, b);
            } else {   // 8, 9, 10: the slot
                if (i < 4) MF_VA(d0, a, ba); else if (i < 8) MF_VA(d1, a, ba);
                else if ((i & 3) == 0) MF_A(e0, a, b); else if ((i & 3) == 1) MF_A(e1, a, b); else if ((i & 3) == 2) MF_A(e2, a, b); else MF_A(e3, a, b);
            }
            if (V == 1 || V == 3 || V == 5 || V == 6 || V == 7 || V == 8 || V == 10 || V == 13) {
                V_EXP(x[i]);
                V_MUL(y[i], x[i]);
                if (i & 1) V_CVT(pk[i >> 1], x[i - 1], x[i]);
            }
            if (V == 11) V_EXP(x[i]);
            if (V == 12) { V_MUL(y[i], x[i]); if (i & 1) V_CVT(pk[i >> 1], x[i - 1], x[i]); }
            if (V == 6 || V == 7 || V == 10) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(l0) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const void*)(lp)));
                asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(l1) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const void*)(lp)));
            }
        }
        if (V == 6 || V == 7 || V == 10) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l0), "+v"(l1));
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(e0), "+a"(e1), "+a"(e2), "+a"(e3));
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 16; ++t) s += d0[t] + d1[t] + d2[t] + d3[t] + e0[t] + e1[t] + e2[t] + e3[t] + x[t] + y[t];
    for (int t = 0; t < 8; ++t) s += (float)pk[t];
    s += l0[0] + l1[1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// The forward attention's mix per 32 x 32 x 16-equivalent of matrix work (one v_mfma_f32_32x32x16_bf16 or two
// v_mfma_f32_16x16x32_bf16): 2 v_exp + 2 v_add + 1 v_cvt_pk.  S: 0 = 32x32x16, 1 = 16x16x32; MIX: vector work or none.
template <int S, int MIX> __global__ __launch_bounds__(256, 1) void probe_shape(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (lane + j)); b[j] = (__bf16)(0.02f * (lane - j)); }
    f32x16 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    f32x4v q0 = {0}, q1 = {0}, q2 = {0}, q3 = {0}, q4 = {0}, q5 = {0}, q6 = {0}, q7 = {0};
    f32x16 x = {0};
    float l0 = 0.f, l1 = 0.f;
    unsigned pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < 16; ++t) x[t] = 0.001f * (lane + t);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (S == 0) {
                if ((i & 3) == 0) MF_V(d0, a, b); else if ((i & 3) == 1) MF_V(d1, a, b); else if ((i & 3) == 2) MF_V(d2, a, b); else MF_V(d3, a, b);
            } else {
                if ((i & 3) == 0) { MF16_V(q0, a, b); } else if ((i & 3) == 1) { MF16_V(q2, a, b); } else if ((i & 3) == 2) { MF16_V(q4, a, b); } else { MF16_V(q6, a, b); }
            }
            if (MIX) { V_EXP(x[i]); V_ADD(l0, x[i]); }
            if (S == 1) {
                if ((i & 3) == 0) { MF16_V(q1, a, b); } else if ((i & 3) == 1) { MF16_V(q3, a, b); } else if ((i & 3) == 2) { MF16_V(q5, a, b); } else { MF16_V(q7, a, b); }
            }
            if (MIX) { V_EXP(x[(i + 8) & 15]); V_ADD(l1, x[(i + 8) & 15]); V_CVT(pk[i & 7], x[i], x[(i + 8) & 15]); }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = l0 + l1;
    for (int t = 0; t < 16; ++t) s += d0[t] + d1[t] + d2[t] + d3[t] + x[t];
    for (int t = 0; t < 4; ++t) s += q0[t] + q1[t] + q2[t] + q3[t] + q4[t] + q5[t] + q6[t] + q7[t];
    for (int t = 0; t < 8; ++t) s += (float)pk[t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int S, int MIX> void run_shape(const char* what, float* out, long long* cyc, int nblk) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) probe_shape<S, MIX><<<nblk, 256>>>(out, cyc, iters);       // (warm: the clock settles under load)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe_shape<S, MIX><<<nblk, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nblk * 4);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double per = sum / h.size() / (iters * 16.0);
    const double ns = ms * 1e6 / (iters * 16.0);
    printf("%-78s %7.2f ticks / 32768 MACs   %7.3f ns   clock %.2f GHz   %.0f TFLOP/s chip\n", what, per, ns, per / ns,
           2.0 * 32768 * 1024 / ns * 1e-3);
}

template <int V> void run(const char* what, float* out, long long* cyc, int nblk) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V><<<nblk, 256>>>(out, cyc, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<V><<<nblk, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nblk * 4);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double per = sum / h.size() / (iters * 16.0);
    printf("V%-2d %-70s %7.2f memtime-ticks/MFMA   %7.3f ns/MFMA\n", V, what, per, ms * 1e6 / (iters * 16.0));
}

int main() {
    const int nblk = 256;
    float* out; long long* cyc;
    hipMalloc(&out, nblk * 256 * sizeof(float));
    hipMalloc(&cyc, nblk * 4 * sizeof(long long));
    run<0>("independent VGPR-dest MFMAs, nothing between", out, cyc, nblk);
    run<1>("independent VGPR-dest + exp, mul, cvt per gap", out, cyc, nblk);
    run<11>("independent VGPR-dest + exp per gap", out, cyc, nblk);
    run<12>("independent VGPR-dest + mul, cvt per gap", out, cyc, nblk);
    run<2>("chains of 4 dependent VGPR-dest, nothing between", out, cyc, nblk);
    run<3>("chains of 4 dependent VGPR-dest + exp, mul, cvt", out, cyc, nblk);
    run<13>("chains of 4 (first with D != C) + exp, mul, cvt", out, cyc, nblk);
    run<4>("independent AGPR-dest, nothing between", out, cyc, nblk);
    run<5>("independent AGPR-dest + exp, mul, cvt", out, cyc, nblk);
    run<6>("independent VGPR-dest + exp, mul, cvt + 2 ds_read_b128", out, cyc, nblk);
    run<7>("chains VGPR-dest + exp, mul, cvt + 2 ds_read_b128", out, cyc, nblk);
    run<9>("slot: 8 chained VGPR-dest (B in AGPR) + 8 AGPR-dest, nothing between", out, cyc, nblk);
    run<8>("slot + exp, mul, cvt", out, cyc, nblk);
    run<10>("slot + exp, mul, cvt + 2 ds_read_b128", out, cyc, nblk);
    printf("-- MFMA shape under the attention forward's vector mix (random-ish operands, every CU busy, one wave per SIMD)\n");
    run_shape<0, 0>("v_mfma_f32_32x32x16_bf16, nothing between", out, cyc, nblk);
    run_shape<1, 0>("2 x v_mfma_f32_16x16x32_bf16, nothing between", out, cyc, nblk);
    run_shape<0, 1>("v_mfma_f32_32x32x16_bf16 + 2 exp, 2 add, 1 cvt_pk", out, cyc, nblk);
    run_shape<1, 1>("2 x v_mfma_f32_16x16x32_bf16 + 2 exp, 2 add, 1 cvt_pk (split around the second MFMA)", out, cyc, nblk);
    return 0;
}
