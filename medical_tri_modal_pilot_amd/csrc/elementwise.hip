// HBM-bound kernels of the hot path for gfx950: custom-LayerNorm backward (K5), TIE/UMSE event
// embedding forward/backward (K1) and the fused AdamW update (K11).  All are one-wave-per-row
// streaming kernels: a lane owns 4 consecutive feature columns (16-byte fp32 / 8-byte bf16
// accesses, 1 KiB / 512 B contiguous per wave instruction), row statistics are wave reductions,
// parameter gradients are accumulated in registers over a grid-stride loop and combined through
// a [blocks][cols] partial slab + a second pass (bitwise reproducible, no float atomics in HBM).
#include "common.hip.h"
#include <hip/hip_fp16.h>

namespace {

constexpr int D = 256;   // d_model, fixed on this path (tri_mbt_vsltcls.py:117,227-228 hard-code it)

// out[c] = sum_r slab[r][c]: a block owns 64 columns; 4 row-lanes sum interleaved rows (coalesced
// 256-byte reads) and combine through LDS in a fixed order (bitwise reproducible).
// blockIdx.y = row group g of gridDim.y: rows g, g + G, g + 2G ... ; out is [gridDim.y][cols].
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* slab, int rows, int cols, float* out) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int G = gridDim.y;
    float s = 0.f;
    if (c < cols) {
#pragma unroll 4
        for (int r = blockIdx.y + rl * G; r < rows; r += 4 * G) s += slab[(size_t)r * cols + c];
    }
    part[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < cols)
        out[(size_t)blockIdx.y * cols + c] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}
// two-level tree: [rows][cols] -> [G][cols] (in ws_tail) -> out[cols].  G is chosen so that the first level has >= 512
// workgroups (a 2048 x 512 slab with 16 groups ran on 128 workgroups, 32 dependent loads per thread: 12-15 us).
}  // namespace
// (also used by the fused dX + LayerNorm-backward GEMM of gemm.hip; declared in common.hip.h)
void launch_slab_reduce(const float* slab, int rows, int cols, float* ws_tail, float* out, hipStream_t st) {
    const int gx = (cols + 63) / 64;
    if (rows <= 64) {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(gx, 1), dim3(256), 0, st, slab, rows, cols, out);
    } else {
        int G = (512 + gx - 1) / gx;
        G = G < 16 ? 16 : (G > RED_GROUPS ? RED_GROUPS : G);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(gx, G), dim3(256), 0, st, slab, rows, cols, ws_tail);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(gx, 1), dim3(256), 0, st, (const float*)ws_tail, G, cols, out);
    }
}
namespace {

// ------------------------------------------------------------------------------------------
// custom LayerNorm backward (module.py:138-144):  y = gamma * (z - mu) / (sigma + eps) + beta,
// sigma = unbiased std.  With xh = (z - mu) * rs, rs = 1/(sigma+eps), g = dy * gamma:
//     dz = (g - mean(g)) * rs - xh * sum(g * xh) / ((D-1) * sigma)
//     dgamma = sum_rows dy * xh ;  dbeta = sum_rows dy
// dz_out = dz (+ d_res): the residual branch of encoder.py:24-32 is added here.
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* z, int ldz, const float* stats, const float* gamma,
                                                     const T* dy, const T* d_res, int ldr, T* dz, int M, float eps,
                                                     float* slab) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 2 * D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4 gm = ld4f(gamma + 4 * lane);
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // A wave owns RPW consecutive rows per trip and issues all of their loads before the first use: one row per
    // trip left a single 512-byte load in flight per wave, i.e. one memory latency per row (58 us for 64 k rows).
    constexpr int RPW = 4;
    const T* res = d_res ? d_res : dy;                     // no residual: read dy again (L1 hit) and scale by 0
    const int ldres = d_res ? ldr : D;
    const float rscale = d_res ? 1.0f : 0.0f;
    for (int row0 = (blockIdx.x * 4 + wave) * RPW; row0 < M; row0 += gridDim.x * 4 * RPW) {
        f32x4 zv[RPW], dv[RPW], rv[RPW];
        float mu[RPW], rs[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const size_t row = (size_t)min(row0 + k, M - 1);
            zv[k] = load4<T>(z + row * ldz + 4 * lane);
            dv[k] = load4<T>(dy + row * D + 4 * lane);
            rv[k] = load4<T>(res + row * ldres + 4 * lane);
            mu[k] = stats[2 * row];
            rs[k] = stats[2 * row + 1];
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const float live = row0 + k < M ? 1.0f : 0.0f;   // rows past M (clamped loads) add nothing and rewrite row M-1 identically
            const float sigma = 1.0f / rs[k] - eps;
            float xh[4], g[4], sg = 0.f, sgx = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xh[i] = (zv[k][i] - mu[k]) * rs[k];
                g[i] = dv[k][i] * gm[i];
                sg += g[i];
                sgx += g[i] * xh[i];
                acc[0][i] += live * dv[k][i] * xh[i];
                acc[1][i] += live * dv[k][i];
            }
            wave_sum2(sg, sgx);
            const float mg = sg * (1.0f / D);
            const float kx = sgx / ((float)(D - 1) * sigma);
            store4<T>(dz + (size_t)min(row0 + k, M - 1) * D + 4 * lane,
                      (g[0] - mg) * rs[k] - xh[0] * kx + rscale * rv[k][0], (g[1] - mg) * rs[k] - xh[1] * kx + rscale * rv[k][1],
                      (g[2] - mg) * rs[k] - xh[2] * kx + rscale * rv[k][2], (g[3] - mg) * rs[k] - xh[3] * kx + rscale * rv[k][3]);
        }
    }
    flush_partials<2>(acc, slab + (size_t)blockIdx.x * 2 * D, lds, lane, wave);
}

// ------------------------------------------------------------------------------------------
// Stream input of the fusion encoder (K4, mbt_encoder.py:697-729 + the concatenation of :745):
//     z[b] = [ bottleneck tokens (nb rows) | dropout(LN(CLS) + PE[0]) | dropout(LN(x[b,t]) + PE[t+1]) ... ]
// nn.LayerNorm (biased variance, eps inside the root) in fp32 on the stream's compute-dtype embeddings, the
// sinusoid rows only for the stream that uses them, dropout by the counter hash of common.hip.h (regenerated in
// the backward), output straight into the [B, nb+1+N, 256] buffer the fusion stack reads -- one launch instead
// of cat + layer_norm + add + dropout + cast + two copies per stream.  One wave per row.
constexpr int NB_MAX = 4;

template <typename T>
__global__ __launch_bounds__(256) void stream_in_fwd_kernel(const T* x, const float* cls, const float* gamma,
                                                            const float* beta, const float* pe, const float* bott, T* out,
                                                            float* stats, int B, int N, int nb, float eps, float p,
                                                            unsigned seed0, const unsigned* seed_dev, const int* row_start,
                                                            const int* kv_len, const T* add = nullptr, int add_L = 1) {
    // add [B * N / add_L][256] (or NULL): row (b N + t - 1) / add_L of it is added to token t of sample b before the LayerNorm, the sum
    // rounded to T as the torch add it replaces rounds it -- the time + modality embedding of the token's image / report
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int R = nb + 1 + N;
    const f32x4 gm = ld4f(gamma + 4 * lane), be = ld4f(beta + 4 * lane);
    const unsigned seed = seed0 ^ (seed_dev ? *seed_dev : 0u);
    const unsigned thr = dropout_threshold(p);
    const float sc = 1.0f / (1.0f - p);
    for (int row = blockIdx.x * 4 + wave; row < B * R; row += gridDim.x * 4) {
        const int b = row / R, r = row - b * R;
        size_t orow = (size_t)row;
        if (row_start) {                                                       // packed output: sample b's kv_len[b] rows start at row_start[b]
            if (r >= kv_len[b]) continue;                                      // (a pad row does not exist there; wave-uniform)
            orow = (size_t)row_start[b] + r;
        }
        T* o = out + orow * D + 4 * lane;
        if (r < nb) {                                                          // wave-uniform
            const f32x4 v = ld4f(bott + r * D + 4 * lane);
            store4<T>(o, v[0], v[1], v[2], v[3]);
            continue;
        }
        const int t = r - nb;                                                  // 0 = CLS
        f32x4 v;
        if (t == 0) {
            const f32x4 c = ld4f(cls + 4 * lane);                              // cls.to(x.dtype) then .float()
            v = f32x4{round_as<T>(c[0]), round_as<T>(c[1]), round_as<T>(c[2]), round_as<T>(c[3])};
        } else {
            const size_t xr = (size_t)b * N + t - 1;
            v = load4<T>(x + xr * D + 4 * lane);
            if (add) {
                const f32x4 av = load4<T>(add + (xr / add_L) * D + 4 * lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = round_as<T>(v[i] + av[i]);
            }
        }
        const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / D);
        float d[4], sq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { d[i] = v[i] - mean; sq += d[i] * d[i]; }
        const float rstd = rsqrtf(wave_sum(sq) * (1.0f / D) + eps);
        float y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = fmaf(d[i] * rstd, gm[i], be[i]);
        if (pe) {
            const f32x4 pv = ld4f(pe + (size_t)t * D + 4 * lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] += pv[i];
        }
        const int lnrow = b * (N + 1) + t;
        if (thr) {
            const unsigned keep = dropout_keep4(seed, (unsigned)lnrow * 64u + lane, thr);
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = (keep >> i) & 1u ? y[i] * sc : 0.f;
        }
        store4<T>(o, y[0], y[1], y[2], y[3]);
        if (lane == 0) { stats[2 * (size_t)lnrow] = mean; stats[2 * (size_t)lnrow + 1] = rstd; }
    }
}

// slab row: [dgamma | dbeta | dcls | dbott[0..3]] = 7 x 256
template <typename T>
MTMP_DEV void stream_in_bwd_body(const T* dz, const T* x, const float* cls, const float* gamma, const float* stats, T* dx, float* slab,
                                 int B, int N, int nb, float p, unsigned seed0, const unsigned* seed_dev, const int* row_start,
                                 const int* kv_len, int block, int nblocks, float* lds, const T* add = nullptr, int add_L = 1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int R = nb + 1 + N;
    const f32x4 gm = ld4f(gamma + 4 * lane);
    const unsigned seed = seed0 ^ (seed_dev ? *seed_dev : 0u);
    const unsigned thr = dropout_threshold(p);
    const float sc = 1.0f / (1.0f - p);
    float acc[7][4];
#pragma unroll
    for (int v = 0; v < 7; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[v][i] = 0.f;
    // Two of the wave's rows are in flight: every load of both rows (gradient, input row, LayerNorm statistics; clamped
    // addresses, no branch around a load) is issued before the first row is used -- one dependent load round per row made this
    // a 52 us launch on the step's tail (1.9 TB/s), where nothing else runs.  Rows are still accumulated in order (same sums).
    constexpr int PF = 2;
    const int stride = nblocks * 4, nrows = B * R;
    for (int row0 = block * 4 + wave; row0 < nrows; row0 += PF * stride) {
        int bb[PF], rr[PF];
        bool live[PF];
        f32x4 gg[PF], vv[PF];
        float mn[PF], rs[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int row = min(row0 + k * stride, nrows - 1);
            const int b = row / R, r = row - b * R;
            bb[k] = b; rr[k] = r;
            live[k] = row0 + k * stride < nrows && !(row_start && r >= kv_len[b]);
            const size_t grow = row_start ? (size_t)row_start[b] + min(r, max(kv_len[b] - 1, 0)) : (size_t)row;
            gg[k] = load4<T>(dz + grow * D + 4 * lane);
            const int t = max(r - nb, 0);
            const size_t xr = (size_t)b * N + max(t - 1, 0);
            vv[k] = load4<T>(x + xr * D + 4 * lane);
            if (add) {                                                         // (the forward's input row: x + its group's add row, rounded to T)
                const f32x4 av = load4<T>(add + (xr / add_L) * D + 4 * lane);
#pragma unroll
                for (int i = 0; i < 4; ++i) vv[k][i] = round_as<T>(vv[k][i] + av[i]);
            }
            const size_t lnrow = (size_t)b * (N + 1) + t;
            mn[k] = stats[2 * lnrow]; rs[k] = stats[2 * lnrow + 1];
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            if (row0 + k * stride >= nrows) continue;
            const int b = bb[k], r = rr[k];
            if (!live[k]) {                                                    // packed gradient rows: a pad row of the padded input x
                if (r > nb) store4<T>(dx + ((size_t)b * N + r - nb - 1) * D + 4 * lane, 0.f, 0.f, 0.f, 0.f);
                continue;
            }
            f32x4 g = gg[k];
            if (r < nb) {                                                      // wave-uniform
#pragma unroll
                for (int q = 0; q < NB_MAX; ++q)
                    if (r == q)
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[3 + q][i] += g[i];
                continue;
            }
            const int t = r - nb, lnrow = b * (N + 1) + t;
            if (thr) {
                const unsigned keep = dropout_keep4(seed, (unsigned)lnrow * 64u + lane, thr);
#pragma unroll
                for (int i = 0; i < 4; ++i) g[i] = (keep >> i) & 1u ? g[i] * sc : 0.f;
            }
            f32x4 v = vv[k];
            if (t == 0) {
                const f32x4 c = ld4f(cls + 4 * lane);
                v = f32x4{round_as<T>(c[0]), round_as<T>(c[1]), round_as<T>(c[2]), round_as<T>(c[3])};
            }
            const float mean = mn[k], rstd = rs[k];
            float xh[4], gy[4], sg = 0.f, sgx = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xh[i] = (v[i] - mean) * rstd;
                acc[0][i] += g[i] * xh[i];
                acc[1][i] += g[i];
                gy[i] = g[i] * gm[i];
                sg += gy[i];
                sgx += gy[i] * xh[i];
            }
            wave_sum2(sg, sgx);
            const float mg = sg * (1.0f / D), mgx = sgx * (1.0f / D);
            float dv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) dv[i] = rstd * (gy[i] - mg - xh[i] * mgx);
            if (t == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[2][i] += dv[i];
            } else {
                store4<T>(dx + ((size_t)b * N + t - 1) * D + 4 * lane, dv[0], dv[1], dv[2], dv[3]);
            }
        }
    }
    flush_partials<7>(acc, slab + (size_t)block * 7 * D, lds, lane, wave);
}
template <typename T>
__global__ __launch_bounds__(256) void stream_in_bwd_kernel(const T* dz, const T* x, const float* cls, const float* gamma,
                                                            const float* stats, T* dx, float* slab, int B, int N, int nb,
                                                            float p, unsigned seed0, const unsigned* seed_dev,
                                                            const int* row_start, const int* kv_len) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 7 * D];              // 28 KiB
    stream_in_bwd_body<T>(dz, x, cls, gamma, stats, dx, slab, B, N, nb, p, seed0, seed_dev, row_start, kv_len, (int)blockIdx.x,
                          (int)gridDim.x, lds);
}
// The three token streams' launches as ONE grid (mtmp_stream_input_bwd_grouped): blocks [first[i], first[i+1]) run stream i's rows
// and write stream i's rows of the partial slab; the slabs lie back to back, so the bottleneck tokens' columns sum over all of them.
constexpr int SI_MAX = 3;
struct StreamInSeg { const void *dz, *x; const float *cls, *gamma, *stats; void* dx; float* slab; const int *row_start, *kv_len;
                     int B, N, nb; float p; unsigned seed; const void* add; int add_L; };
struct StreamInGroup { StreamInSeg seg[SI_MAX]; int first[SI_MAX + 1]; const unsigned* seed_dev; };
template <typename T>
__global__ __launch_bounds__(256) void stream_in_bwd_grouped_kernel(StreamInGroup g) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 7 * D];
    int i = 0;
#pragma unroll
    for (int k = 1; k < SI_MAX; ++k) i += (int)blockIdx.x >= g.first[k] ? 1 : 0;
    const StreamInSeg& q = g.seg[i];
    stream_in_bwd_body<T>((const T*)q.dz, (const T*)q.x, q.cls, q.gamma, q.stats, (T*)q.dx, q.slab, q.B, q.N, q.nb, q.p, q.seed,
                          g.seed_dev, q.row_start, q.kv_len, (int)blockIdx.x - g.first[i], g.first[i + 1] - g.first[i], lds,
                          (const T*)q.add, q.add_L);
}

// ------------------------------------------------------------------------------------------
// TIE / UMSE event embedding (tri_mbt_vsltcls.py:59-71,183-190):
//   E[e,:] = ReLU(LN(v_e * w_v + b_v)) + ReLU(LN(tau_e * w_t + b_t)) + F[f_e]
// events: [n,3] fp32 (time, value, feature index); params: 8 vectors of 256 (w,b,ln_w,ln_b for the
// value chain then the time chain) packed [8][256]; F [20][256].  nn.LayerNorm: biased var, eps 1e-5.
struct TieChain { f32x4 w, b, g, be; };

MTMP_DEV TieChain tie_load_chain(const float* prm, int chain, int lane) {
    const float* base = prm + chain * 4 * D + 4 * lane;
    return TieChain{ld4f(base), ld4f(base + D), ld4f(base + 2 * D), ld4f(base + 3 * D)};
}
// forward of one chain for one event; returns pre-ReLU y, and xh / rstd for the backward
MTMP_DEV void tie_chain_fwd(const TieChain& c, float s, float (&y)[4], float (&xh)[4], float& rstd) {
    float u[4], su = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { u[i] = fmaf(s, c.w[i], c.b[i]); su += u[i]; }
    su = wave_sum(su);
    const float mu = su * (1.0f / D);
    float sv = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float d = u[i] - mu; sv += d * d; }
    sv = wave_sum(sv);
    rstd = rsqrtf(sv * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) { xh[i] = (u[i] - mu) * rstd; y[i] = fmaf(xh[i], c.g[i], c.be[i]); }
}

// Row map of the PACKED event layout (the ragged collate): the batch's events are stored back to
// back, events[cu[b] .. cu[b+1]) belong to sample b, and the embeddings go to the padded [B, t_pad, 256]
// stream layout the fusion stack reads.  Output row (b, t) takes event cu[b] + t when t < cu[b+1] - cu[b]
// and is zero otherwise (a pad row: masked as a key by kv_len, its own output is never read).  With
// cu == nullptr the layout is the reference's padded [n, 3]: row == event.
MTMP_DEV int tie_event_of_row(const int* cu, int t_pad, int row) {
    if (!cu) return row;
    const int b = row / t_pad, t = row - b * t_pad;
    const int lo = cu[b], hi = cu[b + 1];
    return t < hi - lo ? lo + t : -1;
}

// VAL = false: the value chain is left out, E = ReLU(LN(tau*w_t+b_t)) + F[f] -- the time + modality-id embedding
// that is added to every image / text token (tri_mbt_vsltcls.py:216-224: ie_time(t) + ie_feat(18 | 19)).
template <typename T, bool VAL = true>
__global__ __launch_bounds__(256) void tie_fwd_kernel(const float* ev, const float* prm, const float* ftab, T* out, int n,
                                                      const int* cu, int t_pad) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const TieChain cv = tie_load_chain(prm, 0, lane), ct = tie_load_chain(prm, 1, lane);
    for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {
        const int e = tie_event_of_row(cu, t_pad, row);
        if (e < 0) {                                                            // wave-uniform
            store4<T>(out + (size_t)row * D + 4 * lane, 0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const float tau = ev[3 * (size_t)e], val = ev[3 * (size_t)e + 1];
        const int f = min(max((int)ev[3 * (size_t)e + 2], 0), 19);          // x[:,:,2].type(IntTensor), :187
        float yv[4] = {0.f, 0.f, 0.f, 0.f}, yt[4], xh[4], rstd;
        if (VAL) tie_chain_fwd(cv, val, yv, xh, rstd);
        tie_chain_fwd(ct, tau, yt, xh, rstd);
        const f32x4 fe = ld4f(ftab + f * D + 4 * lane);
        store4<T>(out + (size_t)row * D + 4 * lane, fmaxf(yv[0], 0.f) + fmaxf(yt[0], 0.f) + fe[0],
                  fmaxf(yv[1], 0.f) + fmaxf(yt[1], 0.f) + fe[1], fmaxf(yv[2], 0.f) + fmaxf(yt[2], 0.f) + fe[2],
                  fmaxf(yv[3], 0.f) + fmaxf(yt[3], 0.f) + fe[3]);
    }
}

// backward of one chain for one event: accumulates dW, dB, dLNw, dLNb (acc[0..3])
MTMP_DEV void tie_chain_bwd(const TieChain& c, float s, const f32x4& dE, float (&acc)[8][4], int o) {
    float y[4], xh[4], rstd, g[4], sg = 0.f, sgx = 0.f;
    tie_chain_fwd(c, s, y, xh, rstd);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float dy = y[i] > 0.f ? dE[i] : 0.f;
        acc[o + 2][i] += dy * xh[i];
        acc[o + 3][i] += dy;
        g[i] = dy * c.g[i];
        sg += g[i];
        sgx += g[i] * xh[i];
    }
    wave_sum2(sg, sgx);
    const float mg = sg * (1.0f / D), mgx = sgx * (1.0f / D);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float du = rstd * (g[i] - mg - xh[i] * mgx);
        acc[o][i] += du * s;
        acc[o + 1][i] += du;
    }
}

// slab row layout: [8][256] chain grads (same order as prm) then [20][256] feature-table grads
// Second row set (n2 > 0; mtmp_tie_time_embed_bwd_partials): rows n .. n + n2 - 1 are TIME events ev2[n2][3] (the image / text times,
// the time chain and the table only -- their value-chain gradient counts as zero) whose gradient rows come from two tensors,
// dE2a for the first n2a of them and dE2b for the rest: the event embedding's and the time embedding's backward as ONE launch into
// one slab, so the shared time chain / table gradients need no add afterwards.
template <typename T, bool VAL = true>
__global__ __launch_bounds__(256) void tie_bwd_kernel(const float* ev, const float* prm, const T* dE, int n, float* slab,
                                                      const int* cu, int t_pad, const float* ev2 = nullptr,
                                                      const T* dE2a = nullptr, const T* dE2b = nullptr, int n2 = 0, int n2a = 0) {
    // Feature-table gradient: one private [20][256] table per wave (80 KiB), summed in wave order at the end.  A
    // single table with LDS float atomics (the first version) added the four waves' contributions in arrival order:
    // run-to-run 1-ulp differences in d ie_feat (found by the graph-vs-eager bit-equality test once the time
    // embeddings, a handful of events per workgroup, went through this kernel).  The first 32 KiB double as the
    // staging area of flush_partials afterwards.
    __shared__ __attribute__((aligned(16))) float ftab_acc[4 * 20 * D];
    float* lds = ftab_acc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const TieChain cv = tie_load_chain(prm, 0, lane), ct = tie_load_chain(prm, 1, lane);
    for (int i = threadIdx.x; i < 4 * 20 * D; i += 256) ftab_acc[i] = 0.f;
    __syncthreads();
    float* my_tab = ftab_acc + wave * 20 * D;
    float acc[8][4];
#pragma unroll
    for (int v = 0; v < 8; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[v][i] = 0.f;
    // Four of the wave's rows are fetched before the first is used: with 80 KiB of LDS per workgroup only eight waves share a
    // CU, and one dependent global-load round per row (31 rows per wave at config 2) made this a 69 us launch on the step's
    // tail, where nothing else runs.  Rows are still accumulated in order (same sums).
    constexpr int PF = 4;
    const int stride = gridDim.x * 4, nt = n + n2;
    for (int row0 = blockIdx.x * 4 + wave; row0 < nt; row0 += PF * stride) {
        int e[PF];
        bool second[PF];
        float tau[PF], val[PF], ff[PF];
        f32x4 g[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int row = row0 + k * stride;
            second[k] = row >= n && row < nt;                                   // wave-uniform
            if (second[k]) {
                const int r2 = row - n;
                e[k] = r2;
                const float* ep = ev2 + 3 * (size_t)r2;
                tau[k] = ep[0]; val[k] = 0.f; ff[k] = ep[2];
                g[k] = load4<T>((r2 < n2a ? dE2a + (size_t)r2 * D : dE2b + (size_t)(r2 - n2a) * D) + 4 * lane);
            } else {
                e[k] = row < n ? tie_event_of_row(cu, t_pad, row) : -1;
                const size_t eo = 3 * (size_t)max(e[k], 0);
                tau[k] = ev[eo]; val[k] = ev[eo + 1]; ff[k] = ev[eo + 2];
                g[k] = load4<T>(dE + (size_t)min(row, n - 1) * D + 4 * lane);
            }
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            // pad row / past the end: no event -- its gradient row counts as zero (every sum below then adds +0: the same
            // values as skipping it) so that the four rows' chains are straight-line code the scheduler can interleave: with a
            // branch per row the wave ran one dependent chain of ~400 instructions per row at two waves per SIMD (79 us)
            const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 gk = e[k] < 0 ? zero : g[k];
            const int f = min(max((int)ff[k], 0), 19);
            if (VAL) tie_chain_bwd(cv, val[k], second[k] ? zero : gk, acc, 0);
            tie_chain_bwd(ct, tau[k], gk, acc, 4);
            f32x4* tp = reinterpret_cast<f32x4*>(my_tab + f * D + 4 * lane);    // wave-private: plain read-modify-write
            f32x4 tv = *tp;
            tv += gk;
            *tp = tv;
        }
    }
    float* row = slab + (size_t)blockIdx.x * 28 * D;
    __syncthreads();
    for (int i = threadIdx.x; i < 20 * D; i += 256)
        row[8 * D + i] = ((ftab_acc[i] + ftab_acc[20 * D + i]) + ftab_acc[2 * 20 * D + i]) + ftab_acc[3 * 20 * D + i];
    __syncthreads();
    flush_partials<8>(acc, row, lds, lane, wave);
}

// ------------------------------------------------------------------------------------------
// AdamW (2_train.py:110; torch.optim.AdamW single-tensor math) over one flat fp32 buffer, optionally
// refreshing the bf16 shadow copy the MFMA kernels read.
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, bf16* shadow, size_t n,
                                                    float lr, float beta1, float beta2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, float grad_scale) {
    const size_t n4 = n >> 2;
    const float step = lr / bc1, decay = 1.0f - lr * wd;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 pv = ld4f(p + 4 * i), gv = ld4f(g + 4 * i), mv = ld4f(m + 4 * i), vv = ld4f(v + 4 * i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gv[k] * grad_scale;
            pv[k] *= decay;
            mv[k] = mv[k] + (gk - mv[k]) * (1.0f - beta1);          // exp_avg.lerp_(grad, 1-beta1)
            vv[k] = fmaf(vv[k], beta2, (1.0f - beta2) * gk * gk);
            pv[k] -= step * (mv[k] / (sqrtf(vv[k]) / bc2_sqrt + eps));
        }
        *reinterpret_cast<f32x4*>(p + 4 * i) = pv;
        *reinterpret_cast<f32x4*>(m + 4 * i) = mv;
        *reinterpret_cast<f32x4*>(v + 4 * i) = vv;
        if (shadow) *reinterpret_cast<bf16x4*>(shadow + 4 * i) = __builtin_convertvector(pv, bf16x4);
    }
}

int grid_for_rows(int rows) { return max(1, min((rows + 15) / 16, 2048)); }   // 4 waves x 4 rows per trip; <= 8 workgroups per CU

}  // namespace

extern "C" int mtmp_ln_bwd_ws_floats(int M) { return (grid_for_rows(M) + RED_GROUPS) * 2 * D; }

// dz[M,256] = LNbackward(dy; z, stats, gamma) (+ d_res);  dgamma[256], dbeta[256] overwritten.
// ws: mtmp_ln_bwd_ws_floats(M) floats.  Backward of module.py:138-144 (+ the residual of encoder.py:24-32).
extern "C" int mtmp_ln_bwd(int dtype, const void* z, int ldz, const float* stats, const float* gamma, const void* dy,
                           const void* d_res, int ldr, void* dz, float* dgamma_dbeta, float* ws, int M, float eps,
                           void* stream) {
    MTMP_CHECK_ARG(z && stats && gamma && dy && dz && dgamma_dbeta && ws, "mtmp_ln_bwd: null pointer");
    MTMP_CHECK_ARG(M > 0 && ldz >= D && ldz % 4 == 0 && (!d_res || (ldr >= D && ldr % 4 == 0)), "mtmp_ln_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int nb = grid_for_rows(M);
    if (dtype == 0)
        hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)z, ldz, stats, gamma,
                           (const float*)dy, (const float*)d_res, ldr, (float*)dz, M, eps, ws);
    else if (dtype == 1)
        hipLaunchKernelGGL(ln_bwd_kernel<bf16>, dim3(nb), dim3(256), 0, st, (const bf16*)z, ldz, stats, gamma,
                           (const bf16*)dy, (const bf16*)d_res, ldr, (bf16*)dz, M, eps, ws);
    else { mtmp_set_error("mtmp_ln_bwd: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_ln_bwd");
    launch_slab_reduce(ws, nb, 2 * D, ws + (size_t)nb * 2 * D, dgamma_dbeta, st);
    MTMP_CHECK_LAUNCH("mtmp_ln_bwd(reduce)");
    return MTMP_OK;
}

// out[n,256] = TIE embedding of events[n,3]; params [8][256] fp32, ftab [20][256] fp32.
namespace {
int launch_tie_fwd(int dtype, const float* events, const float* params, const float* ftab, void* out, int rows,
                   const int* cu, int t_pad, hipStream_t st, const char* who, bool val = true) {
    const int nb = max(1, min((rows + 3) / 4, 2048));
    if (dtype == 0 && val) hipLaunchKernelGGL(tie_fwd_kernel<float>, dim3(nb), dim3(256), 0, st, events, params, ftab, (float*)out, rows, cu, t_pad);
    else if (dtype == 1 && val) hipLaunchKernelGGL(tie_fwd_kernel<bf16>, dim3(nb), dim3(256), 0, st, events, params, ftab, (bf16*)out, rows, cu, t_pad);
    else if (dtype == 0) hipLaunchKernelGGL((tie_fwd_kernel<float, false>), dim3(nb), dim3(256), 0, st, events, params, ftab, (float*)out, rows, cu, t_pad);
    else if (dtype == 1) hipLaunchKernelGGL((tie_fwd_kernel<bf16, false>), dim3(nb), dim3(256), 0, st, events, params, ftab, (bf16*)out, rows, cu, t_pad);
    else { mtmp_set_error("%s: unknown dtype %d", who, dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH(who);
    return MTMP_OK;
}
int tie_bwd_blocks(int rows) { return max(1, min((rows + 3) / 4, 512)); }
int launch_tie_bwd_partials(int dtype, const float* events, const float* params, const void* d_out, float* ws, int rows, const int* cu,
                            int t_pad, hipStream_t st, const char* who, bool val, const float* ev2 = nullptr, const void* d2a = nullptr,
                            const void* d2b = nullptr, int n2 = 0, int n2a = 0) {
    const int nb = tie_bwd_blocks(rows + n2);
    if (dtype == 0 && val) hipLaunchKernelGGL(tie_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, events, params, (const float*)d_out, rows, ws, cu, t_pad, ev2, (const float*)d2a, (const float*)d2b, n2, n2a);
    else if (dtype == 1 && val) hipLaunchKernelGGL(tie_bwd_kernel<bf16>, dim3(nb), dim3(256), 0, st, events, params, (const bf16*)d_out, rows, ws, cu, t_pad, ev2, (const bf16*)d2a, (const bf16*)d2b, n2, n2a);
    else if (dtype == 0) hipLaunchKernelGGL((tie_bwd_kernel<float, false>), dim3(nb), dim3(256), 0, st, events, params, (const float*)d_out, rows, ws, cu, t_pad, ev2, (const float*)d2a, (const float*)d2b, n2, n2a);
    else if (dtype == 1) hipLaunchKernelGGL((tie_bwd_kernel<bf16, false>), dim3(nb), dim3(256), 0, st, events, params, (const bf16*)d_out, rows, ws, cu, t_pad, ev2, (const bf16*)d2a, (const bf16*)d2b, n2, n2a);
    else { mtmp_set_error("%s: unknown dtype %d", who, dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH(who);
    return MTMP_OK;
}
int launch_tie_bwd(int dtype, const float* events, const float* params, const void* d_out, float* grads, float* ws,
                   int rows, const int* cu, int t_pad, hipStream_t st, const char* who, bool val = true) {
    const int rc = launch_tie_bwd_partials(dtype, events, params, d_out, ws, rows, cu, t_pad, st, who, val);
    if (rc != MTMP_OK) return rc;
    const int nb = tie_bwd_blocks(rows);
    launch_slab_reduce(ws, nb, 28 * D, ws + (size_t)nb * 28 * D, grads, st);
    MTMP_CHECK_LAUNCH(who);
    return MTMP_OK;
}
}  // namespace

extern "C" int mtmp_tie_embed_fwd(int dtype, const float* events, const float* params, const float* ftab, void* out,
                                  int n, void* stream) {
    MTMP_CHECK_ARG(events && params && ftab && out && n > 0, "mtmp_tie_embed_fwd: bad argument");
    return launch_tie_fwd(dtype, events, params, ftab, out, n, nullptr, 0, (hipStream_t)stream, "mtmp_tie_embed_fwd");
}

// Time + modality-id embedding of the image / text tokens (tri_mbt_vsltcls.py:216-224): out[n,256] =
// ReLU(LN(time*w_t+b_t)) + ftab[feature]; events float[n,3] = (time, unused, feature index); same params / grads
// layout as mtmp_tie_embed_* (the value chain's four gradient rows come back zero).
extern "C" int mtmp_time_embed_fwd(int dtype, const float* events, const float* params, const float* ftab, void* out,
                                   int n, void* stream) {
    MTMP_CHECK_ARG(events && params && ftab && out && n > 0, "mtmp_time_embed_fwd: bad argument");
    return launch_tie_fwd(dtype, events, params, ftab, out, n, nullptr, 0, (hipStream_t)stream, "mtmp_time_embed_fwd", false);
}
extern "C" int mtmp_time_embed_bwd(int dtype, const float* events, const float* params, const void* d_out, float* grads,
                                   float* ws, int n, void* stream) {
    MTMP_CHECK_ARG(events && params && d_out && grads && ws && n > 0, "mtmp_time_embed_bwd: bad argument");
    return launch_tie_bwd(dtype, events, params, d_out, grads, ws, n, nullptr, 0, (hipStream_t)stream, "mtmp_time_embed_bwd",
                          false);
}

// Packed (ragged) input: events float[cu[B]][3] back to back, cu int32[B+1] on the device (cu[0] = 0);
// out [B, t_pad, 256] with zero rows past each sample's length (rows past t_pad are dropped: the caller
// truncates to --TIE-len before packing, as dataset_new.py:2020-2021 does).
extern "C" int mtmp_tie_embed_packed_fwd(int dtype, const float* events, const int32_t* cu_seqlens, int B, int t_pad,
                                         const float* params, const float* ftab, void* out, void* stream) {
    MTMP_CHECK_ARG(events && cu_seqlens && params && ftab && out && B > 0 && t_pad > 0 && (long long)B * t_pad < (1ll << 31),
                   "mtmp_tie_embed_packed_fwd: bad argument (B=%d t_pad=%d)", B, t_pad);
    return launch_tie_fwd(dtype, events, params, ftab, out, B * t_pad, cu_seqlens, t_pad, (hipStream_t)stream,
                          "mtmp_tie_embed_packed_fwd");
}

extern "C" int mtmp_tie_bwd_ws_floats(int n) { return (max(1, min((n + 3) / 4, 512)) + RED_GROUPS) * 28 * D; }

// grads[28][256] fp32 (8 chain vectors in params order, then the 20 feature-table rows) overwritten.
extern "C" int mtmp_tie_embed_bwd(int dtype, const float* events, const float* params, const void* d_out, float* grads,
                                  float* ws, int n, void* stream) {
    MTMP_CHECK_ARG(events && params && d_out && grads && ws && n > 0, "mtmp_tie_embed_bwd: bad argument");
    return launch_tie_bwd(dtype, events, params, d_out, grads, ws, n, nullptr, 0, (hipStream_t)stream, "mtmp_tie_embed_bwd");
}

// d_out [B, t_pad, 256]; ws: mtmp_tie_bwd_ws_floats(B * t_pad) floats.
extern "C" int mtmp_tie_embed_packed_bwd(int dtype, const float* events, const int32_t* cu_seqlens, int B, int t_pad,
                                         const float* params, const void* d_out, float* grads, float* ws, void* stream) {
    MTMP_CHECK_ARG(events && cu_seqlens && params && d_out && grads && ws && B > 0 && t_pad > 0 &&
                       (long long)B * t_pad < (1ll << 31),
                   "mtmp_tie_embed_packed_bwd: bad argument (B=%d t_pad=%d)", B, t_pad);
    return launch_tie_bwd(dtype, events, params, d_out, grads, ws, B * t_pad, cu_seqlens, t_pad, (hipStream_t)stream,
                          "mtmp_tie_embed_packed_bwd");
}

// The event embedding's and the time embedding's backward as one launch, PARTIAL slabs only: ws receives
// [mtmp_tie_bwd_slab_rows(n + n_time)][28][256] floats (8 chain vectors in params order, then the 20 table rows), to be summed by
// mtmp_reduce_scatter straight into the parameters' gradient slices.  time_events [n_time][3] = (time, unused, feature index);
// d_time_a: gradient rows of the first n_time_a time events, d_time_b: of the rest (tri_mbt_vsltcls.py:216-224: image / text times).
extern "C" int mtmp_tie_bwd_slab_rows(int n) { return tie_bwd_blocks(n); }
extern "C" int mtmp_tie_time_embed_bwd_partials(int dtype, const float* events, int n, const float* time_events, int n_time,
                                                int n_time_a, const float* params, const void* d_out, const void* d_time_a,
                                                const void* d_time_b, float* ws, void* stream) {
    MTMP_CHECK_ARG(events && params && d_out && ws && n > 0 && n_time >= 0 && n_time_a >= 0 && n_time_a <= n_time &&
                       (n_time == 0 || time_events) && (n_time_a == 0 || d_time_a) && (n_time_a == n_time || d_time_b) &&
                       (long long)n + n_time < (1ll << 31),
                   "mtmp_tie_time_embed_bwd_partials: bad argument (n=%d n_time=%d n_time_a=%d)", n, n_time, n_time_a);
    return launch_tie_bwd_partials(dtype, events, params, d_out, ws, n, nullptr, 0, (hipStream_t)stream,
                                   "mtmp_tie_time_embed_bwd_partials", true, time_events, d_time_a, d_time_b, n_time, n_time_a);
}

extern "C" int mtmp_stream_input_ws_floats(int rows) { return (max(1, min((rows + 15) / 16, 1024)) + RED_GROUPS) * 7 * D; }

// z [B, nb+1+N, 256] (dtype) from x [B, N, 256] (dtype); cls, gamma, beta [256], pe [>= N+1][256] or NULL,
// bott [nb][256] fp32; stats float[B*(N+1)][2] (mean, 1/sqrt(var+eps)) kept for the backward.
// row_start != NULL (with kv_len, both int32 device, mtmp_row_starts): PACKED output -- rows r < kv_len[b] of sample b go to
// rows row_start[b] + r of `out` (same allocation), the others are not written.
extern "C" int mtmp_stream_input_fwd(int dtype, const void* x, const float* cls, const float* gamma, const float* beta,
                                     const float* pe, const float* bott, void* out, float* stats, int B, int N, int nb,
                                     float eps, float p, unsigned seed, const unsigned* seed_dev, const int32_t* row_start,
                                     const int32_t* kv_len, void* stream) {
    MTMP_CHECK_ARG(x && cls && gamma && beta && out && stats && B > 0 && N > 0 && nb >= 0 && nb <= NB_MAX && (nb == 0 || bott) &&
                       p >= 0.f && p < 1.f && (long long)B * (nb + 1 + N) < (1ll << 25) && (!row_start || kv_len),
                   "mtmp_stream_input_fwd: bad argument (B=%d N=%d nb=%d p=%f)", B, N, nb, p);
    hipStream_t st = (hipStream_t)stream;
    const int rows = B * (nb + 1 + N), nbk = max(1, min((rows + 3) / 4, 2048));
    if (dtype == 0)
        hipLaunchKernelGGL(stream_in_fwd_kernel<float>, dim3(nbk), dim3(256), 0, st, (const float*)x, cls, gamma, beta, pe,
                           bott, (float*)out, stats, B, N, nb, eps, p, seed, seed_dev, row_start, kv_len);
    else if (dtype == 1)
        hipLaunchKernelGGL(stream_in_fwd_kernel<bf16>, dim3(nbk), dim3(256), 0, st, (const bf16*)x, cls, gamma, beta, pe,
                           bott, (bf16*)out, stats, B, N, nb, eps, p, seed, seed_dev, row_start, kv_len);
    else { mtmp_set_error("mtmp_stream_input_fwd: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_stream_input_fwd");
    return MTMP_OK;
}

// mtmp_stream_input_fwd with `add` [B N / add_L][256] (dtype, may be NULL): row (b N + t) / add_L is added to input token (b, t) in
// front of the LayerNorm, the sum rounded to `dtype` (= the torch add it replaces: tri_mbt_vsltcls.py:216-224, the time + modality
// embedding of the token's image / report); the backward (mtmp_stream_input_bwd_grouped with the same add) recomputes it.
extern "C" int mtmp_stream_input_fwd_add(int dtype, const void* x, const float* cls, const float* gamma, const float* beta,
                                         const float* pe, const float* bott, void* out, float* stats, int B, int N, int nb,
                                         float eps, float p, unsigned seed, const unsigned* seed_dev, const int32_t* row_start,
                                         const int32_t* kv_len, const void* add, int add_L, void* stream) {
    MTMP_CHECK_ARG(x && cls && gamma && beta && out && stats && B > 0 && N > 0 && nb >= 0 && nb <= NB_MAX && (nb == 0 || bott) &&
                       p >= 0.f && p < 1.f && (long long)B * (nb + 1 + N) < (1ll << 25) && (!row_start || kv_len) && add_L > 0 &&
                       (!add || ((long long)B * N) % add_L == 0),
                   "mtmp_stream_input_fwd_add: bad argument (B=%d N=%d nb=%d p=%f add_L=%d)", B, N, nb, p, add_L);
    hipStream_t st = (hipStream_t)stream;
    const int rows = B * (nb + 1 + N), nbk = max(1, min((rows + 3) / 4, 2048));
    if (dtype == 0)
        hipLaunchKernelGGL(stream_in_fwd_kernel<float>, dim3(nbk), dim3(256), 0, st, (const float*)x, cls, gamma, beta, pe,
                           bott, (float*)out, stats, B, N, nb, eps, p, seed, seed_dev, row_start, kv_len, (const float*)add, add_L);
    else if (dtype == 1)
        hipLaunchKernelGGL(stream_in_fwd_kernel<bf16>, dim3(nbk), dim3(256), 0, st, (const bf16*)x, cls, gamma, beta, pe,
                           bott, (bf16*)out, stats, B, N, nb, eps, p, seed, seed_dev, row_start, kv_len, (const bf16*)add, add_L);
    else { mtmp_set_error("mtmp_stream_input_fwd_add: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_stream_input_fwd_add");
    return MTMP_OK;
}

// dx [B, N, 256] (dtype); grads float[7][256] = dgamma, dbeta, dcls, dbott[0..3] (rows past nb are zero),
// overwritten; ws: mtmp_stream_input_ws_floats(B * (nb+1+N)) floats.  row_start / kv_len: dz is PACKED as the forward's output
// was; dx stays padded (its pad rows are written as zeros).
namespace {
int stream_in_bwd_blocks(int rows) { return max(1, min((rows + 15) / 16, 1024)); }
int launch_stream_in_bwd(int dtype, const void* dz, const void* x, const float* cls, const float* gamma, const float* stats, void* dx,
                         float* ws, int B, int N, int nb, float p, unsigned seed, const unsigned* seed_dev, const int32_t* row_start,
                         const int32_t* kv_len, hipStream_t st, const char* who) {
    const int nbk = stream_in_bwd_blocks(B * (nb + 1 + N));
    if (dtype == 0)
        hipLaunchKernelGGL(stream_in_bwd_kernel<float>, dim3(nbk), dim3(256), 0, st, (const float*)dz, (const float*)x, cls,
                           gamma, stats, (float*)dx, ws, B, N, nb, p, seed, seed_dev, row_start, kv_len);
    else if (dtype == 1)
        hipLaunchKernelGGL(stream_in_bwd_kernel<bf16>, dim3(nbk), dim3(256), 0, st, (const bf16*)dz, (const bf16*)x, cls,
                           gamma, stats, (bf16*)dx, ws, B, N, nb, p, seed, seed_dev, row_start, kv_len);
    else { mtmp_set_error("%s: unknown dtype %d", who, dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH(who);
    return MTMP_OK;
}
}  // namespace
extern "C" int mtmp_stream_input_bwd(int dtype, const void* dz, const void* x, const float* cls, const float* gamma,
                                     const float* stats, void* dx, float* grads, float* ws, int B, int N, int nb, float p,
                                     unsigned seed, const unsigned* seed_dev, const int32_t* row_start, const int32_t* kv_len,
                                     void* stream) {
    MTMP_CHECK_ARG(dz && x && cls && gamma && stats && dx && grads && ws && B > 0 && N > 0 && nb >= 0 && nb <= NB_MAX &&
                       p >= 0.f && p < 1.f && (long long)B * (nb + 1 + N) < (1ll << 25) && (!row_start || kv_len),
                   "mtmp_stream_input_bwd: bad argument (B=%d N=%d nb=%d p=%f)", B, N, nb, p);
    hipStream_t st = (hipStream_t)stream;
    const int rc = launch_stream_in_bwd(dtype, dz, x, cls, gamma, stats, dx, ws, B, N, nb, p, seed, seed_dev, row_start, kv_len, st,
                                        "mtmp_stream_input_bwd");
    if (rc != MTMP_OK) return rc;
    const int nbk = stream_in_bwd_blocks(B * (nb + 1 + N));
    launch_slab_reduce(ws, nbk, 7 * D, ws + (size_t)nbk * 7 * D, grads, st);
    MTMP_CHECK_LAUNCH("mtmp_stream_input_bwd(reduce)");
    return MTMP_OK;
}
// out[j, :] = sum_{t < L} dx[(j L + t), :] for up to two tensors in one launch: the gradient of a per-image / per-report vector that
// was added to every token of its L-token group (the time + modality embedding, tri_mbt_vsltcls.py:216-224).  One workgroup per
// output row: wave w sums tokens w, w + 4, ... in fp32, the four waves combine in a fixed order; result rounded to the tensors' type.
namespace {
struct TokenSums { const void* dx[2]; void* out[2]; int rows[2], L[2]; };
template <typename T>
__global__ __launch_bounds__(256) void token_sums_kernel(TokenSums q) {
    __shared__ __attribute__((aligned(16))) float part[4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = (int)blockIdx.x >= q.rows[0] ? 1 : 0, j = (int)blockIdx.x - (s ? q.rows[0] : 0), L = q.L[s];
    const T* base = (const T*)q.dx[s] + (size_t)j * L * D + 4 * lane;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    int t = wave;
    for (; t + 4 < L; t += 8) {                                    // two rows in flight per wave
        a0 += load4<T>(base + (size_t)t * D);
        a1 += load4<T>(base + (size_t)(t + 4) * D);
    }
    if (t < L) a0 += load4<T>(base + (size_t)t * D);
    *reinterpret_cast<f32x4*>(&part[wave][4 * lane]) = a0 + a1;
    __syncthreads();
    if (wave == 0) {
        const f32x4 v = (*reinterpret_cast<const f32x4*>(&part[0][4 * lane]) + *reinterpret_cast<const f32x4*>(&part[1][4 * lane])) +
                        (*reinterpret_cast<const f32x4*>(&part[2][4 * lane]) + *reinterpret_cast<const f32x4*>(&part[3][4 * lane]));
        store4<T>((T*)q.out[s] + (size_t)j * D + 4 * lane, v[0], v[1], v[2], v[3]);
    }
}
}  // namespace
extern "C" int mtmp_token_sums(int dtype, int n, const void* const* dx, void* const* out, const int* rows, const int* L, void* stream) {
    MTMP_CHECK_ARG(n > 0 && n <= 2 && dx && out && rows && L, "mtmp_token_sums: bad argument (n=%d)", n);
    TokenSums q;
    int total = 0;
    for (int i = 0; i < 2; ++i) {
        const int k = i < n ? i : 0;
        MTMP_CHECK_ARG(dx[k] && out[k] && rows[k] > 0 && L[k] > 0 && (long long)rows[k] * L[k] < (1ll << 25), "mtmp_token_sums: bad entry %d", k);
        q.dx[i] = dx[k]; q.out[i] = out[k]; q.rows[i] = i < n ? rows[k] : 0; q.L[i] = L[k];
        total += q.rows[i];
    }
    if (dtype == 0) hipLaunchKernelGGL(token_sums_kernel<float>, dim3(total), dim3(256), 0, (hipStream_t)stream, q);
    else if (dtype == 1) hipLaunchKernelGGL(token_sums_kernel<bf16>, dim3(total), dim3(256), 0, (hipStream_t)stream, q);
    else { mtmp_set_error("mtmp_token_sums: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_token_sums");
    return MTMP_OK;
}

// The stream-input backward of up to three token streams in ONE launch, partial slabs only: ws receives the streams' slabs back to
// back -- stream i's [mtmp_stream_input_slab_rows(B[i] * (nb[i]+1+N[i]))][7][256] floats behind those of the streams in front of it
// -- for ONE mtmp_reduce_scatter (the bottleneck tokens' columns 768.. sum over all rows of ws when every nb[i] is the same).
// All arrays are HOST arrays of n entries; row_start / kv_len may be NULL as a whole or per entry.
extern "C" int mtmp_stream_input_bwd_grouped(int dtype, int n, const void* const* dz, const void* const* x, const float* const* cls,
                                             const float* const* gamma, const float* const* stats, void* const* dx, float* ws,
                                             const int* B, const int* N, const int* nb, const float* p, const unsigned* seed,
                                             const unsigned* seed_dev, const int32_t* const* row_start,
                                             const int32_t* const* kv_len, const void* const* add, const int* add_L, void* stream) {
    MTMP_CHECK_ARG(n > 0 && n <= SI_MAX && dz && x && cls && gamma && stats && dx && ws && B && N && nb && p && seed,
                   "mtmp_stream_input_bwd_grouped: bad argument (n=%d)", n);
    StreamInGroup g;
    g.seed_dev = seed_dev;
    int total = 0;
    for (int i = 0; i < SI_MAX; ++i) {
        const int k = i < n ? i : 0;
        if (i < n) {
            const int32_t* rs = row_start ? row_start[k] : nullptr;
            const int32_t* kv = kv_len ? kv_len[k] : nullptr;
            MTMP_CHECK_ARG(dz[k] && x[k] && cls[k] && gamma[k] && stats[k] && dx[k] && B[k] > 0 && N[k] > 0 && nb[k] >= 0 &&
                               nb[k] <= NB_MAX && p[k] >= 0.f && p[k] < 1.f && (long long)B[k] * (nb[k] + 1 + N[k]) < (1ll << 25) &&
                               (!rs || kv),
                           "mtmp_stream_input_bwd_grouped: bad stream %d (B=%d N=%d nb=%d p=%f)", k, B[k], N[k], nb[k], p[k]);
            const void* ad = add ? add[k] : nullptr;
            const int aL = (ad && add_L) ? add_L[k] : 1;
            MTMP_CHECK_ARG(aL > 0 && (!ad || ((long long)B[k] * N[k]) % aL == 0), "mtmp_stream_input_bwd_grouped: stream %d: add rows of %d tokens do not divide B N", k, aL);
            g.seg[i] = StreamInSeg{dz[k], x[k], cls[k], gamma[k], stats[k], dx[k], ws + (size_t)total * 7 * D, rs, kv,
                                   B[k], N[k], nb[k], p[k], seed[k], ad, aL};
            g.first[i] = total;
            total += stream_in_bwd_blocks(B[k] * (nb[k] + 1 + N[k]));
        } else {
            g.seg[i] = g.seg[0];
            g.first[i] = total;                    // (empty: no block index reaches it)
        }
    }
    g.first[SI_MAX] = total;
    if (dtype == 0) hipLaunchKernelGGL(stream_in_bwd_grouped_kernel<float>, dim3(total), dim3(256), 0, (hipStream_t)stream, g);
    else if (dtype == 1) hipLaunchKernelGGL(stream_in_bwd_grouped_kernel<bf16>, dim3(total), dim3(256), 0, (hipStream_t)stream, g);
    else { mtmp_set_error("mtmp_stream_input_bwd_grouped: unknown dtype %d", dtype); return MTMP_ERR_ARG; }
    MTMP_CHECK_LAUNCH("mtmp_stream_input_bwd_grouped");
    return MTMP_OK;
}
// The same launch, PARTIAL slabs only: ws receives [mtmp_stream_input_slab_rows(B * (nb+1+N))][7][256] floats (dgamma, dbeta, dcls,
// dbott[0..3]) for mtmp_reduce_scatter.
extern "C" int mtmp_stream_input_slab_rows(int rows) { return stream_in_bwd_blocks(rows); }
extern "C" int mtmp_stream_input_bwd_partials(int dtype, const void* dz, const void* x, const float* cls, const float* gamma,
                                              const float* stats, void* dx, float* ws, int B, int N, int nb, float p, unsigned seed,
                                              const unsigned* seed_dev, const int32_t* row_start, const int32_t* kv_len,
                                              void* stream) {
    MTMP_CHECK_ARG(dz && x && cls && gamma && stats && dx && ws && B > 0 && N > 0 && nb >= 0 && nb <= NB_MAX &&
                       p >= 0.f && p < 1.f && (long long)B * (nb + 1 + N) < (1ll << 25) && (!row_start || kv_len),
                   "mtmp_stream_input_bwd_partials: bad argument (B=%d N=%d nb=%d p=%f)", B, N, nb, p);
    return launch_stream_in_bwd(dtype, dz, x, cls, gamma, stats, dx, ws, B, N, nb, p, seed, seed_dev, row_start, kv_len,
                                (hipStream_t)stream, "mtmp_stream_input_bwd_partials");
}

// In-place AdamW step over flat fp32 buffers of n elements (n % 4 == 0); bf16_shadow may be null.
// grad_scale multiplies the gradient first (1/world_size after an all-reduce SUM).
extern "C" int mtmp_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* bf16_shadow,
                               long long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                               int step, float grad_scale, void* stream) {
    MTMP_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && (n % 4) == 0 && step >= 1,
                   "mtmp_adamw_step: bad argument (n=%lld must be a positive multiple of 4, step >= 1)", n);
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const int nb = (int)max((long long)1, min((n / 4 + 255) / 256, (long long)2048));
    hipLaunchKernelGGL(adamw_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (bf16*)bf16_shadow, (size_t)n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2),
                       grad_scale);
    MTMP_CHECK_LAUNCH("mtmp_adamw_step");
    return MTMP_OK;
}

// ------------------------------------------------------------------------------------------
// Bottleneck exchange between the three modality streams (K9, mbt_encoder.py:764-779).  Every stream
// buffer is [B, 4 + N_m, 256] with the four bottleneck tokens in rows 0..3; the exchanged tokens are
// the per-sample weighted mean over the modalities that are present (missing_num 0: all three, 1: vslt +
// image, 2: vslt + text, 3: vslt only; weights below), optionally averaged with the previous layer's
// (--residual-bottlenecks 1), and are written back into rows 0..3 of all three buffers in place.
// One thread per (sample, token, feature); products and sums in the order of the torch expression
// (bo * w).sum(1), un-fused, so fp32 results are bit-identical to it.
namespace {

__device__ const float kExchangeW[4][3] = {{1.0f / 3.0f, 1.0f / 3.0f, 1.0f / 3.0f}, {0.5f, 0.5f, 0.0f}, {0.5f, 0.0f, 0.5f}, {1.0f, 0.0f, 0.0f}};

template <typename T> struct ExchangeArgs {
    T* z[3];
    long long bstride[3];     // elements between consecutive samples of stream m ((4 + N_m) * 256)
    const long long* missing; // [B] pattern ids 0..3
    const float* prev;        // [B,4,256] fp32 previous exchange result (resbottle) or null
    float* keep;              // [B,4,256] fp32 copy of the new tokens (next layer's prev / backward) or null
    const float* d_prev_in;   // backward: gradient flowing into this exchange's output through the next residual, or null
    float* d_prev_out;        // backward: gradient for the previous exchange's output (resbottle), or null
    int resbottle;
    const int* row_start0;    // stream 0 PACKED: its sample b starts at row row_start0[b] (else null: b * bstride[0])
};

template <typename T> MTMP_DEV size_t exchange_base(const ExchangeArgs<T>& p, int m, int b) {
    return (m == 0 && p.row_start0) ? (size_t)p.row_start0[b] * 256 : (size_t)b * (size_t)p.bstride[m];
}

template <typename T> __global__ __launch_bounds__(256) void exchange_fwd_kernel(ExchangeArgs<T> p) {
    const int b = blockIdx.x >> 2, r = blockIdx.x & 3, f = threadIdx.x;
    // ids outside 0..3 raise IndexError on the host (trainer.py, like the reference's gather); clamped here so that a
    // caller that skipped the check still cannot read outside the table
    const long long pat = min(max(p.missing[b], 0LL), 3LL);
    const size_t o = (size_t)r * 256 + f, ko = ((size_t)b * 4 + r) * 256 + f;
    float v = __fmul_rn(to_f32(p.z[0][exchange_base(p, 0, b) + o]), kExchangeW[pat][0]);
    v = __fadd_rn(v, __fmul_rn(to_f32(p.z[1][b * p.bstride[1] + o]), kExchangeW[pat][1]));
    // two-stream encoder (BimodalTransformerEncoder_MBT, mbt_encoder.py:629-632): z[2] is NULL and the caller maps its
    // patterns {0: mean of both, 1: stream 0} onto rows {1, 3} of the table, whose third weight is 0
    if (p.z[2]) v = __fadd_rn(v, __fmul_rn(to_f32(p.z[2][b * p.bstride[2] + o]), kExchangeW[pat][2]));
    if (p.resbottle) v = __fmul_rn(__fadd_rn(v, p.prev[ko]), 0.5f);
    if (p.keep) p.keep[ko] = v;
    const T t = from_f32<T>(v);
#pragma unroll
    for (int m = 0; m < 3; ++m)
        if (p.z[m]) p.z[m][exchange_base(p, m, b) + o] = t;
}

// z[m] hold dL/d(input of the next layer): rows 0..3 are the gradients w.r.t. the exchanged tokens as seen by
// each consumer.  d_new = sum_m rows (+ d_prev_in); resbottle: half goes to the previous exchange (d_prev_out);
// rows 0..3 of z[m] become d_new * w[m] = gradient w.r.t. stream m's own bottleneck outputs.
template <typename T> __global__ __launch_bounds__(256) void exchange_bwd_kernel(ExchangeArgs<T> p) {
    const int b = blockIdx.x >> 2, r = blockIdx.x & 3, f = threadIdx.x;
    const long long pat = min(max(p.missing[b], 0LL), 3LL);
    const size_t o = (size_t)r * 256 + f, ko = ((size_t)b * 4 + r) * 256 + f;
    float d = to_f32(p.z[0][exchange_base(p, 0, b) + o]);
    d = __fadd_rn(d, to_f32(p.z[1][b * p.bstride[1] + o]));
    if (p.z[2]) d = __fadd_rn(d, to_f32(p.z[2][b * p.bstride[2] + o]));
    if (p.d_prev_in) d = __fadd_rn(d, p.d_prev_in[ko]);
    if (p.resbottle) {
        d = __fmul_rn(d, 0.5f);
        p.d_prev_out[ko] = d;
    }
#pragma unroll
    for (int m = 0; m < 3; ++m)
        if (p.z[m]) p.z[m][exchange_base(p, m, b) + o] = from_f32<T>(__fmul_rn(d, kExchangeW[pat][m]));
}

}  // namespace

// In-place bottleneck exchange over the three stream buffers z_m [B, n_m, 256] (n_m = 4 + tokens, rows 0..3 =
// bottleneck tokens).  missing: int64[B] in 0..3.  resbottle != 0: new = (mean + prev) / 2 with prev [B,4,256]
// fp32.  keep (optional) receives the fp32 result.  Replaces mbt_encoder.py:764-779.  row_start_v (int32[B] device, may be NULL): the
// first buffer is PACKED -- its sample b starts at row row_start_v[b] instead of b * n_v (mtmp_row_starts).
extern "C" int mtmp_bottleneck_exchange_fwd(int dtype, void* z_v, void* z_i, void* z_t, int B, int n_v, int n_i, int n_t,
                                            const long long* missing, int resbottle, const float* prev, float* keep,
                                            const int32_t* row_start_v, void* stream) {
    MTMP_CHECK_ARG(z_v && z_i && missing && B > 0 && n_v >= 4 && n_i >= 4 && (!z_t || n_t >= 4) && (!resbottle || prev),
                   "mtmp_bottleneck_exchange_fwd: bad argument (B=%d rows %d/%d/%d)", B, n_v, n_i, n_t);
    if (dtype == 0) {
        ExchangeArgs<float> a{{(float*)z_v, (float*)z_i, (float*)z_t}, {(long long)n_v * D, (long long)n_i * D, (long long)n_t * D},
                              missing, prev, keep, nullptr, nullptr, resbottle, row_start_v};
        hipLaunchKernelGGL(exchange_fwd_kernel<float>, dim3(B * 4), dim3(256), 0, (hipStream_t)stream, a);
    } else if (dtype == 1) {
        ExchangeArgs<bf16> a{{(bf16*)z_v, (bf16*)z_i, (bf16*)z_t}, {(long long)n_v * D, (long long)n_i * D, (long long)n_t * D},
                             missing, prev, keep, nullptr, nullptr, resbottle, row_start_v};
        hipLaunchKernelGGL(exchange_fwd_kernel<bf16>, dim3(B * 4), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        mtmp_set_error("mtmp_bottleneck_exchange_fwd: unknown dtype %d", dtype);
        return MTMP_ERR_ARG;
    }
    MTMP_CHECK_LAUNCH("mtmp_bottleneck_exchange_fwd");
    return MTMP_OK;
}

// Backward of the exchange, in place on the gradient buffers dz_m [B, n_m, 256] (rows 0..3): see
// exchange_bwd_kernel.  d_prev_in / d_prev_out: [B,4,256] fp32, used with resbottle (d_prev_in may be null).
extern "C" int mtmp_bottleneck_exchange_bwd(int dtype, void* dz_v, void* dz_i, void* dz_t, int B, int n_v, int n_i, int n_t,
                                            const long long* missing, int resbottle, const float* d_prev_in,
                                            float* d_prev_out, const int32_t* row_start_v, void* stream) {
    MTMP_CHECK_ARG(dz_v && dz_i && missing && B > 0 && n_v >= 4 && n_i >= 4 && (!dz_t || n_t >= 4) && (!resbottle || d_prev_out),
                   "mtmp_bottleneck_exchange_bwd: bad argument (B=%d rows %d/%d/%d)", B, n_v, n_i, n_t);
    if (dtype == 0) {
        ExchangeArgs<float> a{{(float*)dz_v, (float*)dz_i, (float*)dz_t}, {(long long)n_v * D, (long long)n_i * D, (long long)n_t * D},
                              missing, nullptr, nullptr, d_prev_in, d_prev_out, resbottle, row_start_v};
        hipLaunchKernelGGL(exchange_bwd_kernel<float>, dim3(B * 4), dim3(256), 0, (hipStream_t)stream, a);
    } else if (dtype == 1) {
        ExchangeArgs<bf16> a{{(bf16*)dz_v, (bf16*)dz_i, (bf16*)dz_t}, {(long long)n_v * D, (long long)n_i * D, (long long)n_t * D},
                             missing, nullptr, nullptr, d_prev_in, d_prev_out, resbottle, row_start_v};
        hipLaunchKernelGGL(exchange_bwd_kernel<bf16>, dim3(B * 4), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        mtmp_set_error("mtmp_bottleneck_exchange_bwd: unknown dtype %d", dtype);
        return MTMP_ERR_ARG;
    }
    MTMP_CHECK_LAUNCH("mtmp_bottleneck_exchange_bwd");
    return MTMP_OK;
}

// ---------------------------------------------------------------------------
// Batched 2-D transposes in ONE launch: dst_i[cols_i][rows_i] = src_i[rows_i][cols_i]^T for up to TB_MAX matrices of 16- or
// 32-bit elements.  The encoder layers' K-contiguous backward operands (W2^T, Wqkv^T, W1^T of every block, rebuilt after
// each optimizer step) were three stack + strided-copy pairs of ~30 us on the critical path of every step; the pointers travel
// by value in the kernel arguments, so a captured hipGraph needs no device-side table.
constexpr int TB_MAX = 64;
struct TransposeBatch { const void* src[TB_MAX]; void* dst[TB_MAX]; int rows[TB_MAX], cols[TB_MAX]; };
template <typename E>
__global__ __launch_bounds__(256) void transpose_batch_kernel(TransposeBatch t) {
    __shared__ E tile[64][65];
    const int b = blockIdx.y, R = t.rows[b], C = t.cols[b];
    const int tiles_c = (C + 63) / 64, tiles = ((R + 63) / 64) * tiles_c;
    const E* src = static_cast<const E*>(t.src[b]);
    E* dst = static_cast<E*>(t.dst[b]);
    for (int w = blockIdx.x; w < tiles; w += gridDim.x) {
        const int r0 = (w / tiles_c) * 64, c0 = (w % tiles_c) * 64;
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + ty + 4 * i, c = c0 + tx;
            if (r < R && c < C) tile[ty + 4 * i][tx] = src[(size_t)r * C + c];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = c0 + ty + 4 * i, r = r0 + tx;
            if (r < R && c < C) dst[(size_t)c * R + r] = tile[tx][ty + 4 * i];
        }
        __syncthreads();
    }
}
// elem_bytes: 2 or 4; src / dst / rows / cols: HOST arrays of n entries (read here, at launch time).
extern "C" int mtmp_transpose_batch(int elem_bytes, const void* const* src, void* const* dst, const int* rows, const int* cols,
                                    int n, void* stream) {
    MTMP_CHECK_ARG(src && dst && rows && cols && n > 0 && (elem_bytes == 2 || elem_bytes == 4), "mtmp_transpose_batch: bad argument");
    for (int base = 0; base < n; base += TB_MAX) {
        TransposeBatch t;
        const int m = min(TB_MAX, n - base);
        int max_tiles = 1;
        for (int i = 0; i < TB_MAX; ++i) {
            const int k = base + (i < m ? i : 0);
            MTMP_CHECK_ARG(src[k] && dst[k] && rows[k] > 0 && cols[k] > 0, "mtmp_transpose_batch: bad entry %d", k);
            t.src[i] = src[k]; t.dst[i] = dst[k]; t.rows[i] = rows[k]; t.cols[i] = cols[k];
            max_tiles = max(max_tiles, ((rows[k] + 63) / 64) * ((cols[k] + 63) / 64));
        }
        dim3 grid(min(max_tiles, 64), m);
        if (elem_bytes == 2) hipLaunchKernelGGL(transpose_batch_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, t);
        else                 hipLaunchKernelGGL(transpose_batch_kernel<unsigned>, grid, dim3(256), 0, (hipStream_t)stream, t);
        MTMP_CHECK_LAUNCH("mtmp_transpose_batch");
    }
    return MTMP_OK;
}

// ---------------------------------------------------------------------------
// Up to CB_MAX device-to-device copies in ONE launch, optionally rounding fp32 data through fp16 on the way
// (x.half().float(): the reference stores its event / time inputs as fp16, trainer.py:26-27, 2_train.py:164).  The replayed
// training step copies its eleven input tensors into the graph's static buffers with this: a dozen ~10 us eager launches on the
// host's critical path between two steps become one.  Pointers travel by value in the kernel arguments.
constexpr int CB_MAX = 16;
struct CopyBatch { const void* src[CB_MAX]; void* dst[CB_MAX]; long long bytes[CB_MAX]; int round16[CB_MAX]; };
__global__ __launch_bounds__(256) void copy_batch_kernel(CopyBatch t) {
    const int b = blockIdx.y;
    const long long n16 = t.bytes[b] >> 4;
    const uint4* s = static_cast<const uint4*>(t.src[b]);
    uint4* d = static_cast<uint4*>(t.dst[b]);
    const bool r16 = t.round16[b] != 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        uint4 v = s[i];
        if (r16) {
            float f[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
            for (int k = 0; k < 4; ++k) f[k] = __half2float(__float2half_rn(f[k]));
            v = uint4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
        }
        d[i] = v;
    }
    if (blockIdx.x == 0) {                                       // tail (< 16 bytes; whole floats when rounding)
        const long long done = n16 << 4, rest = t.bytes[b] - done;
        const char* sc = static_cast<const char*>(t.src[b]) + done;
        char* dc = static_cast<char*>(t.dst[b]) + done;
        if (r16) {
            if ((long long)threadIdx.x * 4 < rest)
                reinterpret_cast<float*>(dc)[threadIdx.x] = __half2float(__float2half_rn(reinterpret_cast<const float*>(sc)[threadIdx.x]));
        } else if (threadIdx.x < rest) {
            dc[threadIdx.x] = sc[threadIdx.x];
        }
    }
}
// src / dst / bytes / round16: HOST arrays of n <= 16 entries; src and dst 16-byte aligned; round16[i] != 0: buffer i holds
// fp32 values (bytes % 4 == 0), written as float(half(x)).
extern "C" int mtmp_copy_batch(const void* const* src, void* const* dst, const long long* bytes, const int* round16, int n,
                               void* stream) {
    MTMP_CHECK_ARG(src && dst && bytes && round16 && n > 0 && n <= CB_MAX, "mtmp_copy_batch: bad argument (n=%d)", n);
    CopyBatch t;
    long long most = 0;
    for (int i = 0; i < CB_MAX; ++i) {
        const int k = i < n ? i : 0;
        MTMP_CHECK_ARG(src[k] && dst[k] && bytes[k] > 0 && ((uintptr_t)src[k] & 15) == 0 && ((uintptr_t)dst[k] & 15) == 0 &&
                           (!round16[k] || bytes[k] % 4 == 0), "mtmp_copy_batch: bad entry %d", k);
        t.src[i] = src[k]; t.dst[i] = dst[k]; t.bytes[i] = bytes[k]; t.round16[i] = round16[k];
        most = most > bytes[k] ? most : bytes[k];
    }
    const long long blocks = (most / 16 + 255) / 256;
    dim3 grid((unsigned)(blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks)), n);
    hipLaunchKernelGGL(copy_batch_kernel, grid, dim3(256), 0, (hipStream_t)stream, t);
    MTMP_CHECK_LAUNCH("mtmp_copy_batch");
    return MTMP_OK;
}

// ---------------------------------------------------------------------------
// Valid-key counts of the three streams in one launch (mbt_encoder.py:703-714): plain = length + 1 (the CLS token), the text
// stream's 3 -> 0 (an absent report); fused = plain + n_bott (the bottleneck prefix).  out: int32 [2][3][B] (plain | fused);
// a NULL length pointer leaves that stream's rows untouched (an unmasked stream has no count).  Replaces ~14 one-element-wide
// torch kernels in front of the fusion stack.
__global__ void stream_lengths_kernel(const long long* lv, const long long* li, const long long* lt, int* out, int B, int n_bott,
                                      int txt_idx) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const long long* src[3] = {lv, li, lt};
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        if (!src[m]) continue;
        int v = (int)src[m][b] + 1;
        if (m == txt_idx && v == 3) v = 0;
        out[m * B + b] = v;
        out[(3 + m) * B + b] = v + n_bott;
    }
}
extern "C" int mtmp_stream_lengths(const long long* len_v, const long long* len_i, const long long* len_t, int* out, int B,
                                   int n_bott, int txt_idx, void* stream) {
    MTMP_CHECK_ARG(out && B > 0 && (len_v || len_i || len_t), "mtmp_stream_lengths: bad argument (B=%d)", B);
    hipLaunchKernelGGL(stream_lengths_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, len_v, len_i, len_t, out, B,
                       n_bott, txt_idx);
    MTMP_CHECK_LAUNCH("mtmp_stream_lengths");
    return MTMP_OK;
}

// Row map of a PACKED token stream: out[b] = sum of min(max(kv_len[0..b), 0), n_max) -- sample b's first row when the samples'
// kv_len[b] valid rows (bottleneck prefix + CLS + events) are stored back to back with no pad rows in between -- and out[B] =
// the rows in use.  That last word is what the row-panel / weight-gradient kernels take as `rows_live`; the buffers and
// the launch grids keep the padded size B * n_max, so a captured hipGraph replays whatever the lengths are.
// out[B + 1 .. 2 B + 1) = the order in which the attention kernels walk the samples: their grids are cut into eight contiguous
// chunks, one per XCD (common.hip.h xcd_remap), and the cost of a sample grows with the square of its length -- in batch order one
// XCD gets the eight longest samples of a ragged batch and the launch waits for it.  The samples are ranked by length and
// dealt round-robin: slot i of XCD x (x = 0..7) holds the sample of rank 8 i + x, so every XCD gets the same mix and starts
// with its longest samples.
__global__ __launch_bounds__(256) void row_starts_kernel(const int* kv_len, int* out, int B, int n_max) {
    __shared__ int s[256];
    for (int b = threadIdx.x; b < B; b += 256) {
        const int v = min(max(kv_len[b], 0), n_max);
        int rank = B > 2048 ? b : 0;                                  // (ranking is O(B^2) in one workgroup: batch order beyond that)
        for (int c = 0; c < (B > 2048 ? 0 : B); ++c) {
            const int u = min(max(kv_len[c], 0), n_max);
            rank += (u > v || (u == v && c < b)) ? 1 : 0;
        }
        const int x = rank & 7, j = rank >> 3;
        int pos = j;                                                  // slots of the XCDs in front of x, then j
        for (int xx = 0; xx < x; ++xx) pos += (B - xx + 7) >> 3;
        out[B + 1 + pos] = b;
    }
    int carry = 0;
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int b = b0 + (int)threadIdx.x;
        const int v = b < B ? min(max(kv_len[b], 0), n_max) : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int t = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (b < B) out[b] = carry + s[threadIdx.x] - v;
        carry += s[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[B] = carry;
}
extern "C" int mtmp_row_starts(const int32_t* kv_len, int32_t* out, int B, int n_max, void* stream) {
    MTMP_CHECK_ARG(kv_len && out && B > 0 && n_max > 0 && (long long)B * n_max < (1ll << 31), "mtmp_row_starts: bad argument (B=%d n_max=%d)", B, n_max);
    hipLaunchKernelGGL(row_starts_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, kv_len, out, B, n_max);
    MTMP_CHECK_LAUNCH("mtmp_row_starts");
    return MTMP_OK;
}

// Slots of the frozen image encoder for a batch in which some samples have no image (their encoder output is read by nothing:
// the bottleneck exchange gives the image stream weight 0 for them, mbt_encoder.py:764-779 with missing_num 2 / 3).  pattern:
// int64[B] missing_num ids, sample b HAS an image iff pattern[b] < present_below; out: int32[2 B + 1 + 15]:
//   out[i], i < B          slot i works on image out[i] of the batch (present images first, in batch order)
//   out[B + b]             the slot of sample b's image, or B (a slot the caller keeps zero) when it has none
//   out[2 B]               present images
//   out[2 B + 1 + 5 p + s] rows in use at encoder stage s = 0..3 (hw0 >> 2 s rows per image; s = 4: one row per image) for
//                          p = 0: the whole batch, p = 1 / 2: its first / second half of B / 2 slots (the two-stream tail)
__global__ __launch_bounds__(256) void image_slots_kernel(const long long* pattern, int present_below, int* out, int B, int hw0) {
    __shared__ int s[256];
    int carry = 0;
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int b = b0 + (int)threadIdx.x;
        const int v = b < B && pattern[b] >= 0 && pattern[b] < present_below ? 1 : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int t = (int)threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (b < B) out[B + b] = v ? carry + s[threadIdx.x] - 1 : -(b - (carry + s[threadIdx.x])) - 1;   // missing: -(rank among the missing) - 1
        carry += s[255];
        __syncthreads();
    }
    const int live = carry;
    for (int b = threadIdx.x; b < B; b += 256) {
        const int v = out[B + b];
        const int slot = v >= 0 ? v : live + (-v - 1);
        out[slot] = b;
        out[B + b] = v >= 0 ? v : B;
    }
    if (threadIdx.x < 15) {
        const int p = threadIdx.x / 5, st = threadIdx.x % 5, half = B / 2;
        const int n = p == 0 ? live : p == 1 ? min(live, half) : max(live - half, 0);
        out[2 * B + 1 + threadIdx.x] = n * (st < 4 ? hw0 >> (2 * st) : 1);
    }
    if (threadIdx.x == 0) out[2 * B] = live;
}
extern "C" int mtmp_image_slots(const long long* pattern, int present_below, int32_t* out, int B, int hw0, void* stream) {
    MTMP_CHECK_ARG(pattern && out && B > 0 && hw0 > 0 && hw0 % 64 == 0 && (long long)B * hw0 < (1ll << 31),
                   "mtmp_image_slots: bad argument (B=%d hw0=%d)", B, hw0);
    hipLaunchKernelGGL(image_slots_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pattern, present_below, out, B, hw0);
    MTMP_CHECK_LAUNCH("mtmp_image_slots");
    return MTMP_OK;
}

// pair[0] <- bits of *value, pair[1] += 1: a scalar (the step's loss) and a sequence number, published together so that the host
// can take the value from pinned memory as soon as THIS point of the stream is reached (graph.py, GraphedTrainStep.publish_loss).
__global__ void publish_scalar_kernel(const float* value, unsigned* pair) {
    pair[0] = __float_as_uint(*value);
    pair[1] = pair[1] + 1u;
}
extern "C" int mtmp_publish_scalar(const float* value, unsigned* pair, void* stream) {
    MTMP_CHECK_ARG(value && pair, "mtmp_publish_scalar: null pointer");
    hipLaunchKernelGGL(publish_scalar_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, value, pair);
    MTMP_CHECK_LAUNCH("mtmp_publish_scalar");
    return MTMP_OK;
}

// Stream-ordered time stamp: one lane stores the 100 MHz wall clock into *slot.  Works inside a captured hipGraph, where HIP
// events cannot be timed: bench.py brackets the roofline kernel of the replayed steps with two of these (its in-step duration).
__global__ void timestamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

extern "C" int mtmp_timestamp(unsigned long long* slot, void* stream) {
    MTMP_CHECK_ARG(slot, "mtmp_timestamp: null slot");
    hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot);
    MTMP_CHECK_LAUNCH("mtmp_timestamp");
    return MTMP_OK;
}
