// Classification head of TRI_MBT_VSLTCLS (K10, tri_mbt_vsltcls.py:59-76 `ie_demo`, :248-255 head) for gfx950:
//     demo = ReLU(LN(age * w[:,0] + gender * w[:,1] + b))                         ie_demo (Linear(2,256), LayerNorm, ReLU)
//     x    = [ LN(cls) | demo ]                                                   layer_norms_after_concat, torch.cat
//     h    = x W1^T + b1 ;  y = BatchNorm1d(h) ;  a = ReLU(y) ;  out = a W2^T + b2  fc_list.{0,1,2,3}
// Everything here is a few hundred KFLOP on B <= 256 rows: the torch modules cost ~65 launches of ~5 us forward +
// backward, all of them on the critical path between the forward and the backward of the step.  Six launches here.
//   forward : head_x (one wave per row: the two LayerNorms) -> head_fc (feature-parallel: 4 features per workgroup,
//             a wave = one feature x all rows -- R = 1, 2 or 4 rows per lane --, so the BatchNorm statistics are wave reductions) -> head_out
//   backward: head_fc_bwd (feature-parallel: BatchNorm / Linear weight gradients, dh) -> head_x_bwd (row-parallel:
//             dx = dh W1, both LayerNorm backwards, per-row partials of the row-summed gradients) -> slab reduce
// fp32 throughout (the reference runs the head in fp32 under autocast too).  Deterministic: no float atomics.
#include "common.hip.h"

namespace {

constexpr int D = 256, DX = 512, MAXB = 256, FPW = 4;  // d_model, head input width, most rows per call, features per workgroup
inline int rows_per_lane(int B) { return B <= 64 ? 1 : (B <= 128 ? 2 : 4); }

MTMP_DEV float block_sum(float v, float* red, int tid) {   // 256 threads -> every thread gets the sum
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

struct HeadParams {
    const float* demo_w; const float* demo_b; const float* demo_g; const float* demo_be;   // ie_demo.{0.weight[256][2],0.bias,1.weight,1.bias}
    const float* ln_g; const float* ln_b;                                                   // layer_norms_after_concat
    const float* w1; const float* b1;                                                       // fc_list.0 [256][512]
    const float* bn_g; const float* bn_b; float* run_mean; float* run_var;                  // fc_list.1
    const float* w2; const float* b2;                                                       // fc_list.3 [1][256]
};

// x[b] = [LN(cls[b]) | ReLU(LN(age*w0 + gen*w1 + b))]; one wave per row, lane owns 4 columns of each half.
template <typename TC>
__global__ __launch_bounds__(256) void head_x_kernel(const TC* cls, const float* age, const float* gen, HeadParams p,
                                                     float* x, int B, float eps, long long* nbt) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;          // BatchNorm1d.num_batches_tracked (one launch less in front of the head)
    if (b >= B) return;
    const int c = 4 * lane;
    f32x4 v = load4<TC>(cls + (size_t)b * D + c);
    float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / D), d[4], sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { d[i] = v[i] - mean; sq += d[i] * d[i]; }
    float rstd = rsqrtf(wave_sum(sq) * (1.0f / D) + eps);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaf(d[i] * rstd, p.ln_g[c + i], p.ln_b[c + i]);
    *reinterpret_cast<f32x4*>(x + (size_t)b * DX + c) = o;
    const float a = age[b], g = gen[b];
    float u[4], su = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { u[i] = fmaf(a, p.demo_w[2 * (c + i)], fmaf(g, p.demo_w[2 * (c + i) + 1], p.demo_b[c + i])); su += u[i]; }
    mean = wave_sum(su) * (1.0f / D);
    sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { d[i] = u[i] - mean; sq += d[i] * d[i]; }
    rstd = rsqrtf(wave_sum(sq) * (1.0f / D) + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaxf(fmaf(d[i] * rstd, p.demo_g[c + i], p.demo_be[c + i]), 0.f);
    *reinterpret_cast<f32x4*>(x + (size_t)b * DX + D + c) = o;
}

// workgroup g owns features 4g..4g+3; thread = (rows b = lane + 64 rr for rr < R, feature jj = wave)
template <int R>
__global__ __launch_bounds__(256) void head_fc_kernel(const float* x, HeadParams p, float* hhat, float* rstd_out, float* partial,
                                                      int B, float eps, float momentum, int training) {
    constexpr int NB = 64 * R;                  // row capacity of this instantiation
    __shared__ float xT[64][NB + 1];            // one 64-column chunk of x, transposed: xT[k][b]
    __shared__ float pl[FPW][NB];
    const int tid = threadIdx.x, lane = tid & 63, jj = __builtin_amdgcn_readfirstlane(tid >> 6), j = blockIdx.x * FPW + jj;
    const float* wrow = p.w1 + (size_t)j * DX;
    float acc[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) acc[rr] = 0.f;
    for (int k0 = 0; k0 < DX; k0 += 64) {
        __syncthreads();
        // NB rows x 64 columns: thread t loads rows (t>>2) + 64 rr, columns 16*(t&3) .. +15
        const int cs = 16 * (tid & 3);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int r = (tid >> 2) + 64 * rr;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < B) v = *reinterpret_cast<const f32x4*>(x + (size_t)r * DX + k0 + cs + 4 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) xT[cs + 4 * q + i][r] = v[i];
            }
        }
        __syncthreads();
#pragma unroll 16
        for (int kk = 0; kk < 64; ++kk) {
            const float wv = wrow[k0 + kk];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) acc[rr] = fmaf(xT[kk][lane + 64 * rr], wv, acc[rr]);
        }
    }
    float h[R];
    bool live[R];
    float sh = 0.f;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        h[rr] = acc[rr] + p.b1[j];
        live[rr] = lane + 64 * rr < B;
        sh += live[rr] ? h[rr] : 0.f;
    }
    float mean, var;
    if (training) {
        mean = wave_sum(sh) / (float)B;
        float sv = 0.f;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) { const float dv = live[rr] ? h[rr] - mean : 0.f; sv += dv * dv; }
        var = wave_sum(sv) / (float)B;                                         // biased, used for normalisation
        if (lane == 0) {                                                       // running statistics (unbiased variance)
            p.run_mean[j] = (1.0f - momentum) * p.run_mean[j] + momentum * mean;
            p.run_var[j] = (1.0f - momentum) * p.run_var[j] + momentum * var * ((float)B / (float)max(B - 1, 1));
        }
    } else {
        mean = p.run_mean[j];
        var = p.run_var[j];
    }
    const float rstd = rsqrtf(var + eps);
    if (lane == 0) rstd_out[j] = rstd;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int b = lane + 64 * rr;
        const float hh = (h[rr] - mean) * rstd;
        const float a = fmaxf(fmaf(hh, p.bn_g[j], p.bn_b[j]), 0.f);
        if (live[rr]) hhat[(size_t)b * D + j] = hh;
        pl[jj][b] = live[rr] ? a * p.w2[j] : 0.f;
    }
    __syncthreads();
    for (int t = tid; t < NB; t += 256) partial[(size_t)blockIdx.x * NB + t] = pl[0][t] + pl[1][t] + pl[2][t] + pl[3][t];
}

__global__ __launch_bounds__(256) void head_out_kernel(const float* partial, const float* b2, float* out, int B, int nb) {
    const int b = threadIdx.x;                  // nb = row stride of `partial` (64 x rows per lane)
    if (b >= B) return;
    float s = 0.f;
    for (int g = 0; g < D / FPW; ++g) s += partial[(size_t)g * nb + b];
    out[b] = s + b2[0];
}

// feature-parallel backward: dW2, db2, dbn_g, dbn_b, db1, dW1, dh[b][j]
template <int R>
__global__ __launch_bounds__(256) void head_fc_bwd_kernel(const float* dout, const float* x, const float* hhat, const float* rstd_in,
                                                          HeadParams p, float* dh_out, float* dw1, float* db1, float* dbn_g,
                                                          float* dbn_b, float* dw2, float* db2, int B, int training) {
    constexpr int NB = 64 * R;
    __shared__ float dhs[FPW][NB];
    const int tid = threadIdx.x, lane = tid & 63, jj = __builtin_amdgcn_readfirstlane(tid >> 6), j = blockIdx.x * FPW + jj;
    const float gam = p.bn_g[j], rstd = rstd_in[j];
    float dl[R], hh[R], dy[R];
    bool live[R];
    float t_w2 = 0.f, t_g = 0.f, t_b = 0.f, t_dl = 0.f;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int b = lane + 64 * rr;
        live[rr] = b < B;
        dl[rr] = live[rr] ? dout[b] : 0.f;
        hh[rr] = live[rr] ? hhat[(size_t)b * D + j] : 0.f;
        const float y = fmaf(hh[rr], gam, p.bn_b[j]);
        const float a = fmaxf(y, 0.f);
        dy[rr] = (live[rr] && y > 0.f) ? dl[rr] * p.w2[j] : 0.f;
        t_w2 += dl[rr] * a; t_g += dy[rr] * hh[rr]; t_b += dy[rr]; t_dl += dl[rr];
    }
    const float s_w2 = wave_sum(t_w2), s_g = wave_sum(t_g), s_b = wave_sum(t_b);
    float t_b1 = 0.f;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int b = lane + 64 * rr;
        const float dh = training ? gam * rstd * (dy[rr] - s_b / (float)B - hh[rr] * s_g / (float)B) : gam * rstd * dy[rr];
        const float dhl = live[rr] ? dh : 0.f;
        t_b1 += dhl;
        if (live[rr]) dh_out[(size_t)b * D + j] = dh;
        dhs[jj][b] = dhl;
    }
    const float s_b1 = wave_sum(t_b1);
    if (lane == 0) { dw2[j] = s_w2; dbn_g[j] = s_g; dbn_b[j] = s_b; db1[j] = s_b1; }
    if (blockIdx.x == 0 && jj == 0) {
        const float s = wave_sum(t_dl);
        if (lane == 0) db2[0] = s;
    }
    __syncthreads();
    // dW1[j][k] = sum_b dh[b][j] x[b][k]: thread t owns columns t and t + 256 of the workgroup's four rows of W1
    float s0[FPW] = {0.f, 0.f, 0.f, 0.f}, s1[FPW] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < B; ++r) {
        const float x0 = x[(size_t)r * DX + tid], x1 = x[(size_t)r * DX + D + tid];
#pragma unroll
        for (int f = 0; f < FPW; ++f) { s0[f] = fmaf(dhs[f][r], x0, s0[f]); s1[f] = fmaf(dhs[f][r], x1, s1[f]); }
    }
#pragma unroll
    for (int f = 0; f < FPW; ++f) {
        dw1[(size_t)(blockIdx.x * FPW + f) * DX + tid] = s0[f];
        dw1[(size_t)(blockIdx.x * FPW + f) * DX + D + tid] = s1[f];
    }
}

// row-parallel backward: workgroup = row b, thread = column k of each half of x.
// slab row b: [dln_g | dln_b | ddemo_g | ddemo_be | ddemo_w0 | ddemo_w1 | ddemo_b] = 7 x 256
// WIL: the two columns of ie_demo.0.weight's gradient interleaved in the slab row ([256][2], the parameter's own layout) instead of
// as two rows of 256 -- the reduction then writes the parameter's gradient slice as it stands.
template <typename TC, bool WIL>
__global__ __launch_bounds__(256) void head_x_bwd_kernel(const float* dh, const TC* cls, const float* age, const float* gen,
                                                         HeadParams p, TC* dcls, float* slab, float eps) {
    __shared__ float dhr[D];
    __shared__ float red[4];
    const int b = blockIdx.x, k = threadIdx.x;
    dhr[k] = dh[(size_t)b * D + k];
    __syncthreads();
    float dxc = 0.f, dxd = 0.f;
    for (int j = 0; j < D; ++j) {
        const float g = dhr[j];
        dxc = fmaf(g, p.w1[(size_t)j * DX + k], dxc);
        dxd = fmaf(g, p.w1[(size_t)j * DX + D + k], dxd);
    }
    float* row = slab + (size_t)b * 7 * D;
    // LayerNorm(cls) backward
    {
        const float v = to_f32(cls[(size_t)b * D + k]);
        const float mean = block_sum(v, red, k) * (1.0f / D);
        const float d = v - mean;
        const float rstd = rsqrtf(block_sum(d * d, red, k) * (1.0f / D) + eps);
        const float xh = d * rstd, gy = dxc * p.ln_g[k];
        const float m1 = block_sum(gy, red, k) * (1.0f / D), m2 = block_sum(gy * xh, red, k) * (1.0f / D);
        dcls[(size_t)b * D + k] = from_f32<TC>(rstd * (gy - m1 - xh * m2));
        row[k] = dxc * xh;
        row[D + k] = dxc;
    }
    // ie_demo backward
    {
        const float a = age[b], g = gen[b];
        const float u = fmaf(a, p.demo_w[2 * k], fmaf(g, p.demo_w[2 * k + 1], p.demo_b[k]));
        const float mean = block_sum(u, red, k) * (1.0f / D);
        const float d = u - mean;
        const float rstd = rsqrtf(block_sum(d * d, red, k) * (1.0f / D) + eps);
        const float xh = d * rstd;
        const float dyd = fmaf(xh, p.demo_g[k], p.demo_be[k]) > 0.f ? dxd : 0.f;
        const float gy = dyd * p.demo_g[k];
        const float m1 = block_sum(gy, red, k) * (1.0f / D), m2 = block_sum(gy * xh, red, k) * (1.0f / D);
        const float du = rstd * (gy - m1 - xh * m2);
        row[2 * D + k] = dyd * xh;
        row[3 * D + k] = dyd;
        if (WIL) {
            row[4 * D + 2 * k] = du * a;
            row[4 * D + 2 * k + 1] = du * g;
        } else {
            row[4 * D + k] = du * a;
            row[5 * D + k] = du * g;
        }
        row[6 * D + k] = du;
    }
}

// out[c] = sum_r slab[r][c], r < rows <= 256: 256 threads over columns
__global__ __launch_bounds__(256) void head_rows_reduce_kernel(const float* slab, int rows, int cols, float* out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += slab[(size_t)r * cols + c];
    out[c] = s;
}

// the same sums with every 256-column block of the slab going to a destination of its own (block i -> dst[i]; blocks 4 and 5 are
// the interleaved ie_demo.0.weight gradient: one destination of 512)
struct HeadDst { float* d[7]; };
// (a block = 64 columns x 4 row lanes combined in a fixed order: 28 blocks -- as 7 blocks of 256 columns walking all rows one after
//  the other, like head_rows_reduce_kernel, this launch took 16 us on the chain between the loss and the backward)
__global__ __launch_bounds__(256) void head_rows_scatter_kernel(const float* slab, int rows, HeadDst t) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f;
    int r = rl;
    for (; r + 4 < rows; r += 8) {
        s0 += slab[(size_t)r * 7 * D + c];
        s1 += slab[(size_t)(r + 4) * 7 * D + c];
    }
    if (r < rows) s0 += slab[(size_t)r * 7 * D + c];
    part[rl][cl] = s0 + s1;
    __syncthreads();
    if (rl == 0) {
        const float s = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        const int blk = c / D, k = c - blk * D;
        if (blk == 5) t.d[4][D + k] = s;
        else t.d[blk][k] = s;
    }
}

// BCEWithLogitsLoss(reduction="mean") (2_train.py:76, trainer.py:128): loss = mean_b [max(o,0) - o t + log(1 + exp(-|o|))];
// dlogit[b] = (sigmoid(o) - t) / n is produced alongside (the backward is a scale by the incoming scalar).
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* o, const float* t, float* loss, float* dlogit, int n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float x = o[i], y = t[i];
        s += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
        dlogit[i] = (1.0f / (1.0f + expf(-x)) - y) / (float)n;
    }
    s = block_sum(s, red, threadIdx.x);
    if (threadIdx.x == 0) loss[0] = s / (float)n;
}

}  // namespace

extern "C" int mtmp_bce_logits_mean(const float* logits, const float* target, float* loss, float* dlogit, int n, void* stream) {
    MTMP_CHECK_ARG(logits && target && loss && dlogit && n > 0, "mtmp_bce_logits_mean: bad argument");
    hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, loss, dlogit, n);
    MTMP_CHECK_LAUNCH("mtmp_bce_logits_mean");
    return MTMP_OK;
}

// ws_fwd: x[B][512] + hhat[B][256] + rstd[256] + partial[64][64 x rows per lane] floats (kept for the backward)
extern "C" int mtmp_head_ws_floats(int B) { return B * DX + B * D + D + (D / FPW) * 64 * rows_per_lane(B); }

// params: 14 device pointers in HeadParams order (float; run_mean / run_var are updated in place when training).
// cls float[B][256], age / gender float[B]; out float[B]; B <= 256 (one, two or four rows per lane).
extern "C" int mtmp_head_fwd(const float* cls, const float* age, const float* gender, const void* const* params, float* out,
                             float* ws, int B, float ln_eps, float bn_eps, float momentum, int training, void* stream) {
    MTMP_CHECK_ARG(cls && age && gender && params && out && ws && B > 0 && B <= MAXB && (training == 0 || B > 1),
                   "mtmp_head_fwd: bad argument (B=%d, needs 1 <= B <= %d, and B > 1 in training mode)", B, MAXB);
    HeadParams p{(const float*)params[0], (const float*)params[1], (const float*)params[2], (const float*)params[3],
                 (const float*)params[4], (const float*)params[5], (const float*)params[6], (const float*)params[7],
                 (const float*)params[8], (const float*)params[9], (float*)params[10], (float*)params[11],
                 (const float*)params[12], (const float*)params[13]};
    hipStream_t st = (hipStream_t)stream;
    float* x = ws; float* hhat = x + (size_t)B * DX; float* rstd = hhat + (size_t)B * D; float* partial = rstd + D;
    hipLaunchKernelGGL(head_x_kernel<float>, dim3((B + 3) / 4), dim3(256), 0, st, cls, age, gender, p, x, B, ln_eps, (long long*)nullptr);
    MTMP_CHECK_LAUNCH("mtmp_head_fwd(x)");
    const int R = rows_per_lane(B);
    auto fc = R == 1 ? head_fc_kernel<1> : (R == 2 ? head_fc_kernel<2> : head_fc_kernel<4>);
    hipLaunchKernelGGL(fc, dim3(D / FPW), dim3(256), 0, st, (const float*)x, p, hhat, rstd, partial, B, bn_eps, momentum, training);
    MTMP_CHECK_LAUNCH("mtmp_head_fwd(fc)");
    hipLaunchKernelGGL(head_out_kernel, dim3(1), dim3(256), 0, st, (const float*)partial, p.b2, out, B, 64 * R);
    MTMP_CHECK_LAUNCH("mtmp_head_fwd(out)");
    return MTMP_OK;
}

// grads (float, overwritten): dcls[B][256]; g_rows[7][256] = dln_g, dln_b, ddemo_g, ddemo_be, ddemo_w[:,0], ddemo_w[:,1],
// ddemo_b; dw1[256][512]; g_feat[4][256] = db1, dbn_g, dbn_b, dw2; db2[1].  ws_bwd: B*256 + B*7*256 floats.
extern "C" int mtmp_head_bwd(const float* d_out, const float* cls, const float* age, const float* gender,
                             const void* const* params, const float* ws_fwd, float* dcls, float* g_rows, float* dw1,
                             float* g_feat, float* db2, float* ws_bwd, int B, float ln_eps, int training, void* stream) {
    MTMP_CHECK_ARG(d_out && cls && age && gender && params && ws_fwd && dcls && g_rows && dw1 && g_feat && db2 && ws_bwd &&
                       B > 0 && B <= MAXB, "mtmp_head_bwd: bad argument (B=%d)", B);
    HeadParams p{(const float*)params[0], (const float*)params[1], (const float*)params[2], (const float*)params[3],
                 (const float*)params[4], (const float*)params[5], (const float*)params[6], (const float*)params[7],
                 (const float*)params[8], (const float*)params[9], (float*)params[10], (float*)params[11],
                 (const float*)params[12], (const float*)params[13]};
    hipStream_t st = (hipStream_t)stream;
    const float* x = ws_fwd; const float* hhat = x + (size_t)B * DX; const float* rstd = hhat + (size_t)B * D;
    float* dh = ws_bwd; float* slab = dh + (size_t)B * D;
    const int R = rows_per_lane(B);
    auto fcb = R == 1 ? head_fc_bwd_kernel<1> : (R == 2 ? head_fc_bwd_kernel<2> : head_fc_bwd_kernel<4>);
    hipLaunchKernelGGL(fcb, dim3(D / FPW), dim3(256), 0, st, d_out, x, hhat, rstd, p, dh, dw1, g_feat, g_feat + D, g_feat + 2 * D,
                       g_feat + 3 * D, db2, B, training);
    MTMP_CHECK_LAUNCH("mtmp_head_bwd(fc)");
    hipLaunchKernelGGL((head_x_bwd_kernel<float, false>), dim3(B), dim3(256), 0, st, (const float*)dh, cls, age, gender, p, dcls, slab, ln_eps);
    MTMP_CHECK_LAUNCH("mtmp_head_bwd(x)");
    hipLaunchKernelGGL(head_rows_reduce_kernel, dim3(7), dim3(256), 0, st, (const float*)slab, B, 7 * D, g_rows);
    MTMP_CHECK_LAUNCH("mtmp_head_bwd(reduce)");
    return MTMP_OK;
}

// The head with the CLS vectors in the fusion stack's own type (ABI 6): cls / dcls [B][256] in `cls_dtype` (MTMP_F32 | MTMP_BF16: no
// cast launch on either side of the head), num_batches_tracked (int64 on the device, may be NULL) incremented by the first launch,
// and every parameter's gradient written to a destination of its own -- dst[12] in the order of the 12 trained parameters
// (ie_demo.0.weight [256][2], ie_demo.0.bias, ie_demo.1.weight, ie_demo.1.bias, layer_norms_after_concat.{weight,bias},
// fc_list.0.{weight [256][512], bias}, fc_list.1.{weight, bias}, fc_list.3.{weight, bias}): slices of the flat gradient buffer,
// or scratch.  ws_bwd: B*256 + B*7*256 floats.
extern "C" int mtmp_head_fwd_t(int cls_dtype, const void* cls, const float* age, const float* gender, const void* const* params,
                               float* out, float* ws, int B, float ln_eps, float bn_eps, float momentum, int training,
                               long long* num_batches_tracked, void* stream) {
    MTMP_CHECK_ARG(cls && age && gender && params && out && ws && B > 0 && B <= MAXB && (training == 0 || B > 1) &&
                       (cls_dtype == 0 || cls_dtype == 1),
                   "mtmp_head_fwd_t: bad argument (B=%d, needs 1 <= B <= %d, and B > 1 in training mode; dtype %d)", B, MAXB, cls_dtype);
    HeadParams p{(const float*)params[0], (const float*)params[1], (const float*)params[2], (const float*)params[3],
                 (const float*)params[4], (const float*)params[5], (const float*)params[6], (const float*)params[7],
                 (const float*)params[8], (const float*)params[9], (float*)params[10], (float*)params[11],
                 (const float*)params[12], (const float*)params[13]};
    hipStream_t st = (hipStream_t)stream;
    float* x = ws; float* hhat = x + (size_t)B * DX; float* rstd = hhat + (size_t)B * D; float* partial = rstd + D;
    if (cls_dtype == 0)
        hipLaunchKernelGGL(head_x_kernel<float>, dim3((B + 3) / 4), dim3(256), 0, st, (const float*)cls, age, gender, p, x, B, ln_eps,
                           num_batches_tracked);
    else
        hipLaunchKernelGGL(head_x_kernel<bf16>, dim3((B + 3) / 4), dim3(256), 0, st, (const bf16*)cls, age, gender, p, x, B, ln_eps,
                           num_batches_tracked);
    MTMP_CHECK_LAUNCH("mtmp_head_fwd_t(x)");
    const int R = rows_per_lane(B);
    auto fc = R == 1 ? head_fc_kernel<1> : (R == 2 ? head_fc_kernel<2> : head_fc_kernel<4>);
    hipLaunchKernelGGL(fc, dim3(D / FPW), dim3(256), 0, st, (const float*)x, p, hhat, rstd, partial, B, bn_eps, momentum, training);
    MTMP_CHECK_LAUNCH("mtmp_head_fwd_t(fc)");
    hipLaunchKernelGGL(head_out_kernel, dim3(1), dim3(256), 0, st, (const float*)partial, p.b2, out, B, 64 * R);
    MTMP_CHECK_LAUNCH("mtmp_head_fwd_t(out)");
    return MTMP_OK;
}
extern "C" int mtmp_head_bwd_scatter(int cls_dtype, const float* d_out, const void* cls, const float* age, const float* gender,
                                     const void* const* params, const float* ws_fwd, void* dcls, float* const* dst, float* ws_bwd,
                                     int B, float ln_eps, int training, void* stream) {
    MTMP_CHECK_ARG(d_out && cls && age && gender && params && ws_fwd && dcls && dst && ws_bwd && B > 0 && B <= MAXB &&
                       (cls_dtype == 0 || cls_dtype == 1),
                   "mtmp_head_bwd_scatter: bad argument (B=%d dtype %d)", B, cls_dtype);
    for (int i = 0; i < 12; ++i) MTMP_CHECK_ARG(dst[i], "mtmp_head_bwd_scatter: destination %d is NULL", i);
    HeadParams p{(const float*)params[0], (const float*)params[1], (const float*)params[2], (const float*)params[3],
                 (const float*)params[4], (const float*)params[5], (const float*)params[6], (const float*)params[7],
                 (const float*)params[8], (const float*)params[9], (float*)params[10], (float*)params[11],
                 (const float*)params[12], (const float*)params[13]};
    hipStream_t st = (hipStream_t)stream;
    const float* x = ws_fwd; const float* hhat = x + (size_t)B * DX; const float* rstd = hhat + (size_t)B * D;
    float* dh = ws_bwd; float* slab = dh + (size_t)B * D;
    const int R = rows_per_lane(B);
    auto fcb = R == 1 ? head_fc_bwd_kernel<1> : (R == 2 ? head_fc_bwd_kernel<2> : head_fc_bwd_kernel<4>);
    hipLaunchKernelGGL(fcb, dim3(D / FPW), dim3(256), 0, st, d_out, x, hhat, rstd, p, dh, dst[6], dst[7], dst[8], dst[9], dst[10], dst[11],
                       B, training);
    MTMP_CHECK_LAUNCH("mtmp_head_bwd_scatter(fc)");
    if (cls_dtype == 0)
        hipLaunchKernelGGL((head_x_bwd_kernel<float, true>), dim3(B), dim3(256), 0, st, (const float*)dh, (const float*)cls, age, gender, p,
                           (float*)dcls, slab, ln_eps);
    else
        hipLaunchKernelGGL((head_x_bwd_kernel<bf16, true>), dim3(B), dim3(256), 0, st, (const float*)dh, (const bf16*)cls, age, gender, p,
                           (bf16*)dcls, slab, ln_eps);
    MTMP_CHECK_LAUNCH("mtmp_head_bwd_scatter(x)");
    // slab blocks: dln_g, dln_b, ddemo_g, ddemo_be, ddemo_w (two blocks, interleaved), ddemo_b
    HeadDst t{{dst[4], dst[5], dst[2], dst[3], dst[0], nullptr, dst[1]}};
    hipLaunchKernelGGL(head_rows_scatter_kernel, dim3(7 * D / 64), dim3(256), 0, st, (const float*)slab, B, t);
    MTMP_CHECK_LAUNCH("mtmp_head_bwd_scatter(reduce)");
    return MTMP_OK;
}
