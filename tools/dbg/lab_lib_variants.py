# scratch: VARIANTS = {name: [(file, old, new), ...]} for tools/lab_lib.py (patched copies of csrc/, timing experiments only)
VARIANTS = {
    "clock": [("Makefile", "-Wno-unused-result -mllvm", "-Wno-unused-result -DMTMP_LAB_CLOCK -mllvm")],
} for tools/lab_lib.py (patched copies of csrc/, timing experiments only)
_G_OLD = """    const float u = x * fmaf(x * x, 0.0713548163f * 1.4426950408889634f, 1.5957691216f * 1.4426950408889634f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u));"""
_G_POLY = """    const float s = __builtin_amdgcn_fmed3f(x, -3.588370285689392f, 3.588370285689392f), s2 = s * s;
    float p = fmaf(s2, -5.669414018398632e-07f, 2.972501561546562e-05f);
    p = fmaf(p, s2, -0.0006558468838514445f);
    p = fmaf(p, s2, 0.008079891918797707f);
    p = fmaf(p, s2, -0.06318190744753219f);
    p = fmaf(p, s2, 0.3969548399538981f);
    return x * __builtin_amdgcn_fmed3f(fmaf(s, p, 0.5f), 0.0f, 1.0f);"""
_G2 = """template <> MTMP_DEV float gelu<bf16>(float x) {"""
_G2_NEW = """typedef float f32x2_t __attribute__((ext_vector_type(2)));
MTMP_DEV void gelu2(float& a, float& b) {
    const f32x2_t x = {a, b};
    const f32x2_t s = {__builtin_amdgcn_fmed3f(a, -3.588370285689392f, 3.588370285689392f), __builtin_amdgcn_fmed3f(b, -3.588370285689392f, 3.588370285689392f)};
    const f32x2_t s2 = s * s;
    f32x2_t p = __builtin_elementwise_fma(s2, (f32x2_t){-5.669414018398632e-07f, -5.669414018398632e-07f}, (f32x2_t){2.972501561546562e-05f, 2.972501561546562e-05f});
    p = __builtin_elementwise_fma(p, s2, (f32x2_t){-0.0006558468838514445f, -0.0006558468838514445f});
    p = __builtin_elementwise_fma(p, s2, (f32x2_t){0.008079891918797707f, 0.008079891918797707f});
    p = __builtin_elementwise_fma(p, s2, (f32x2_t){-0.06318190744753219f, -0.06318190744753219f});
    p = __builtin_elementwise_fma(p, s2, (f32x2_t){0.3969548399538981f, 0.3969548399538981f});
    const f32x2_t q = __builtin_elementwise_fma(s, p, (f32x2_t){0.5f, 0.5f});
    a = x[0] * __builtin_amdgcn_fmed3f(q[0], 0.0f, 1.0f);
    b = x[1] * __builtin_amdgcn_fmed3f(q[1], 0.0f, 1.0f);
}
template <> MTMP_DEV float gelu<bf16>(float x) {"""
_LOOP = """            for (int t = 0; t < 16; ++t) h[t] = gelu<bf16>(h[t]);
            const Frag<bf16> h0 = frag_from_acc<bf16>(h, 0), h1 = frag_from_acc<bf16>(h, 1);
#pragma unroll
            for (int o = 0; o < G::OG; ++o) {
                mma<bf16>(acc2[o], vf[o][0], h0);"""
_LOOP2 = """            for (int t = 0; t < 16; t += 2) { float ga = h[t], gb = h[t + 1]; gelu2(ga, gb); h[t] = ga; h[t + 1] = gb; }
            const Frag<bf16> h0 = frag_from_acc<bf16>(h, 0), h1 = frag_from_acc<bf16>(h, 1);
#pragma unroll
            for (int o = 0; o < G::OG; ++o) {
                mma<bf16>(acc2[o], vf[o][0], h0);"""
VARIANTS = {
    "clock": [("Makefile", "-Wno-unused-result -mllvm", "-Wno-unused-result -DMTMP_LAB_CLOCK -mllvm")],
    "mlp_nogelu": [("common.hip.h", _G_OLD, "    return 0.5f * x;")],
    "mlp_poly": [("common.hip.h", _G_OLD, _G_POLY)],
    "mlp_poly2": [("common.hip.h", _G2, _G2_NEW), ("swin.hip", _LOOP, _LOOP2)],
}

_NT_OLD = """    auto multiply = [&]() {
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
    };"""
_NT_PIPE = """    auto multiply = [&]() {
        Frag<T> a[2][G::RT], wq[2][G::NT];
        auto ldf = [&](int c, int b) __attribute__((always_inline)) {
#pragma unroll
            for (int rt = 0; rt < G::RT; ++rt) a[b][rt] = frag_load<T>(sA + (32 * (G::RT * wr + rt) + r) * LDW + 16 * c + 8 * half);
#pragma unroll
            for (int nt = 0; nt < G::NT; ++nt) wq[b][nt] = frag_load<T>(sW + (foff + 32 * nt + r) * LDW + 16 * c + 8 * half);
        };
        ldf(0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 3) ldf(c + 1, (c + 1) & 1);
#pragma unroll
            for (int nt = 0; nt < G::NT; ++nt)
#pragma unroll
                for (int rt = 0; rt < G::RT; ++rt) mma<T>(acc[rt][nt], wq[c & 1][nt], a[c & 1][rt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };"""
_LB_OLD = "__global__ __launch_bounds__(256, (sizeof(T) == 2 ? (TM == 64 ? 3 : 4) : 1)) void gemm_nt_kernel("
_LB_3 = "__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 3 : 1)) void gemm_nt_kernel("
VARIANTS["nt_pipe4"] = [("gemm.hip", _NT_OLD, _NT_PIPE)]
VARIANTS["nt_pipe3"] = [("gemm.hip", _NT_OLD, _NT_PIPE), ("gemm.hip", _LB_OLD, _LB_3)]
VARIANTS["nt_lb3"] = [("gemm.hip", _LB_OLD, _LB_3)]
