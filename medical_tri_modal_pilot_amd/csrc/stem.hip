// CXR patch-embedding stem of the Swin-T image encoder (first stage) for gfx950:
//   Conv2d(1, 96, kernel 4, stride 4) -> NHWC -> LayerNorm(96, eps 1e-5)
// (builder/models/src/swin_transformer.py:559-567 with the 1-channel stem of :646) as an
// implicit GEMM: rows = 96 output channels (3 MFMA tiles), columns = patches (one per lane),
// K = the 16 pixels of a patch -- one MFMA k-step, no im2col buffer.  A patch's 4x4 pixels
// are four 16-byte runs in the image, and consecutive lanes take consecutive patches of a
// row, so every load is a coalesced 512-byte segment.  With the patch on the lane, the 96
// channels of a patch sit in that lane pair's accumulators: the LayerNorm statistics are an
// in-register sum plus one cross-half add.  Memory-bound (reads the image once, writes the
// feature map once).
#include "common.hip.h"

namespace {

constexpr int C = 96;

template <typename T> MTMP_DEV Frag<T> frag_from_f32(const f32x4& a, const f32x4& b) {
    Frag<T> f;
    f.v[0] = from_f32<T>(a[0]); f.v[1] = from_f32<T>(a[1]); f.v[2] = from_f32<T>(a[2]); f.v[3] = from_f32<T>(a[3]);
    f.v[4] = from_f32<T>(b[0]); f.v[5] = from_f32<T>(b[1]); f.v[6] = from_f32<T>(b[2]); f.v[7] = from_f32<T>(b[3]);
    return f;
}

template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const float* img, const float* w, const float* bias, const float* ln_w,
                                                   const float* ln_b, T* out, int n_img, int H, int W, const int* order,
                                                   const int* rows_live) {
    __shared__ __attribute__((aligned(16))) float sp[3 * C];     // bias | ln_w | ln_b
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, half = lane >> 5;
    for (int i = tid; i < 3 * C; i += 256) sp[i] = i < C ? bias[i] : (i < 2 * C ? ln_w[i - C] : ln_b[i - 2 * C]);
    const int pw = W >> 2, ph = H >> 2, per_img = pw * ph;
    long long total = (long long)n_img * per_img;
    if (rows_live) total = min(total, (long long)*rows_live);     // (patches of the slots in use; see mtmp_image_slots)
    Frag<T> wf[3];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        const float* wr = w + (32 * ct + r) * 16 + 8 * half;     // weight [96][1][4][4]: k = 4*p + q
        wf[ct] = frag_from_f32<T>(*reinterpret_cast<const f32x4*>(wr), *reinterpret_cast<const f32x4*>(wr + 4));
    }
    __syncthreads();
    const long long P = ((long long)blockIdx.x * 4 + wave) * 32 + r;
    const bool ok = P < total;
    Frag<T> pf = frag_zero<T>();
    if (ok) {
        const int b = (int)(P / per_img), rem = (int)(P - (long long)b * per_img), pi = rem / pw, pj = rem - pi * pw;
        const int bs = order ? order[b] : b;                     // slot b holds image order[b] of the batch
        const float* src = img + ((size_t)bs * H + 4 * pi + 2 * half) * W + 4 * pj;   // rows p = 2*half, 2*half+1
        pf = frag_from_f32<T>(*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + W));
    }
    f32x16 acc[3] = {{0}, {0}, {0}};
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) mma<T>(acc[ct], wf[ct], pf);
    // + bias, LayerNorm over the 96 channels of this lane pair's patch
    float s1 = 0.f;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            acc[ct][t] += sp[32 * ct + acc_row(t, half)];
            s1 += acc[ct][t];
        }
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * (1.0f / C);
    float s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int t = 0; t < 16; ++t) { const float d = acc[ct][t] - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = rsqrtf(s2 * (1.0f / C) + 1e-5f);
    if (ok) {
        T* o = out + (size_t)P * C;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = 32 * ct + 8 * g + 4 * half;
                float y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    y[i] = fmaf((acc[ct][4 * g + i] - mean) * rstd, sp[C + ch + i], sp[2 * C + ch + i]);
                store4<T>(o + ch, y[0], y[1], y[2], y[3]);
            }
    }
}

}  // namespace

// out[n_img, H/4, W/4, 96] = LayerNorm(Conv2d_4x4s4(img[n_img,1,H,W]) NHWC); weights/params fp32.
// Replaces swin_transformer.py:559-567 (features[0]) with the 1-channel stem of :646.
extern "C" int mtmp_swin_stem_fwd_live(int dtype, const float* img, const float* w, const float* bias, const float* ln_w,
                                       const float* ln_b, void* out, int n_img, int H, int W, const int32_t* order,
                                       const int32_t* rows_live, void* stream);
extern "C" int mtmp_swin_stem_fwd(int dtype, const float* img, const float* w, const float* bias, const float* ln_w,
                                  const float* ln_b, void* out, int n_img, int H, int W, void* stream) {
    return mtmp_swin_stem_fwd_live(dtype, img, w, bias, ln_w, ln_b, out, n_img, H, W, nullptr, nullptr, stream);
}
// order (int32[n_img] device, may be NULL): output slot i is made from image order[i]; rows_live (may be NULL): a device word
// with the patch rows in use (live slots x (H/4)(W/4)) -- mtmp_image_slots: present images first, the rest not computed.
extern "C" int mtmp_swin_stem_fwd_live(int dtype, const float* img, const float* w, const float* bias, const float* ln_w,
                                       const float* ln_b, void* out, int n_img, int H, int W, const int32_t* order,
                                       const int32_t* rows_live, void* stream) {
    MTMP_CHECK_ARG(img && w && bias && ln_w && ln_b && out, "mtmp_swin_stem_fwd: null pointer");
    MTMP_CHECK_ARG(n_img > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "mtmp_swin_stem_fwd: bad shape %dx%dx%d", n_img, H, W);
    const long long total = (long long)n_img * (H / 4) * (W / 4);
    const int nb = (int)((total + 127) / 128);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) hipLaunchKernelGGL(stem_kernel<float>, dim3(nb), dim3(256), 0, st, img, w, bias, ln_w, ln_b, (float*)out, n_img, H, W, order, rows_live);
    else if (dtype == 1) hipLaunchKernelGGL(stem_kernel<bf16>, dim3(nb), dim3(256), 0, st, img, w, bias, ln_w, ln_b, (bf16*)out, n_img, H, W, order, rows_live);
    else { mtmp_set_error("mtmp_swin_stem_fwd: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_swin_stem_fwd");
    return MTMP_OK;
}
