/* mtmp.h -- C ABI of libmtmp_hip.so: the MI355X (gfx950) kernels behind the tri-modal
 * training hot path of AITRICS/Medical_Tri_Modal_Pilot (model tri_mbt_vsltcls).
 *
 * The reference is 100 % Python/PyTorch: there is no FFI in it to mirror.  The boundary it
 * does have is the sequence of ATen calls made by its nn.Modules; each entry point below
 * replaces one such chain and cites it (paths relative to the reference root).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types.  All pointers are DEVICE pointers owned by the
 *     caller (including workspaces and saved-for-backward buffers); nothing here allocates,
 *     synchronises or keeps global state.  Calls are asynchronous on `stream` (a hipStream_t
 *     passed as void*), re-entrant across streams/devices.
 *   - dtype: MTMP_F32 = parity build (fp32 storage, v_mfma_f32_32x32x2_f32, exact fp32 fma
 *     chains), MTMP_BF16 = performance build (bf16 storage, v_mfma_f32_32x32x16_bf16, fp32
 *     accumulate / softmax / LayerNorm statistics).  Same kernels, same indexing.
 *   - return 0 on success; non-zero = argument or launch error, text via mtmp_last_error()
 *     (thread-local).  Row strides ("ld") are in elements.
 *   - d_model = 256 = 4 heads x 64 is fixed on this path (tri_mbt_vsltcls.py:117,227-228).
 */
#ifndef MTMP_H
#define MTMP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MTMP_F32 0
#define MTMP_BF16 1

int mtmp_abi_version(void);
const char* mtmp_last_error(void);

/* Modality-aware multi-head attention forward.
 * Replaces builder/models/src/transformer/attention.py:24-49 (bmm, /sqrt(d), masked_fill(-65504),
 * softmax, bmm), the head split/merge of attention.py:72-82 and the key-pad mask of
 * builder/models/src/transformer/utils.py:79-125.
 * q,k,v: [B,N,ld_qkv] (head h = columns [64h,64h+64)); o: [B,N,ld_o]; kv_len: int32[B] valid
 * keys per sample (NULL = unmasked; 0 = fully masked -> uniform average, as the reference);
 * lse: float[B,H,N] out (log2 units, consumed by mtmp_attn_bwd).  If res/o_res are non-NULL,
 * o_res = o + res (the "outputs += residual" of encoder.py:27).
 * key_norms (may be NULL): float[ceil(B N / 32)][H], max ||k_h||_2 over each block of 32 consecutive rows of the [B N] token
 * space (mtmp_key_norms, or the epilogue of the projection that produced k).  With it a wave whose queries satisfy
 * ||q|| max||k|| scale log2(e) <= 64 skips the running maximum of the softmax altogether (exp2 cannot leave the f32 range
 * and the softmax is shift invariant: same result, ~40 % fewer vector instructions); other waves, and every wave when it is
 * NULL, run the online-maximum form. */
int mtmp_attn_fwd(int dtype, const void* q, const void* k, const void* v, void* o, const void* res, void* o_res,
                  float* lse, const int32_t* kv_len, const float* key_norms, int B, int N, int H, int ld_qkv, int ld_o,
                  float scale, void* stream);
/* Grouped forms: up to three token streams (vital signs / image / text of one fusion layer) in ONE launch -- the blocks of the
 * short streams follow the long stream's in the same grid instead of occupying workgroup slots beside it from other HIP
 * streams.  All pointer / int arrays are HOST arrays of n entries (1 <= n <= 3); res / o_res / kv_len / key_norms may be NULL
 * as a whole or per entry; B, H, scale are common.  Same kernels, same results as n calls of the single forms.
 * row_start (may be NULL as a whole or per entry): stream i is PACKED -- row_start[i] is int32[B] on the device
 * (mtmp_row_starts), sample b's kv_len[i][b] tokens are rows row_start[i][b] .. of the q / k / v / o / res / d_o / dq / dk / dv
 * buffers, with no pad rows between samples (kv_len[i][b] is then also the sample's query count); N[i] is the longest
 * sample the buffers were sized for, and stays the stride of lse / delta_ws ([B, H, N[i]]). */
int mtmp_attn_fwd_grouped(int dtype, int n, const void* const* q, const void* const* k, const void* const* v, void* const* o,
                          const void* const* res, void* const* o_res, float* const* lse, const int32_t* const* kv_len,
                          const int32_t* const* row_start, const float* const* key_norms, const int* N, const int* ld_qkv,
                          const int* ld_o, int B, int H, float scale, void* stream);
int mtmp_attn_bwd_grouped(int dtype, int n, const void* const* q, const void* const* k, const void* const* v,
                          const void* const* o, const void* const* d_o, const float* const* lse, const int32_t* const* kv_len,
                          const int32_t* const* row_start, void* const* dq, void* const* dk, void* const* dv,
                          float* const* delta_ws, const int* N, const int* ld_qkv, const int* ld_o, const int* ld_do,
                          const int* ld_dqkv, int B, int H, float scale, void* stream);
/* Attention of ONE query row per sample -- the CLS token of the last fusion layer of the vital-sign stream: tri_mbt_vsltcls.py:248
 * reads nothing of the encoder's result but outputs[0][:, 0, :], so in the LAST layer only that row's attention output, FFN and
 * residuals feed the loss (its keys / values still come from all rows).  q / k / v [rows, ld_qkv], res [rows, ld_res] (the layer
 * input: residual of encoder.py:27); kv_len / row_start as in mtmp_attn_fwd_grouped (both may be NULL: padded [B, N], all keys
 * valid); cls_tok = the query's token index inside a sample.  _fwd: o_cls, r1_cls [B, H * 64] (the attention output and output +
 * residual, both in `dtype`), lse float[B, H].  _bwd: d_o [B, H * 64] = gradient w.r.t. o_cls -> dq, dk, dv written for EVERY row
 * of every sample (dense [rows, ld_dqkv]; dq is zero outside row cls_tok, rows past kv_len are zero): what mtmp_gemm_tn /
 * mtmp_gemm_lnbwd read next.  Replaces attention.py:24-84 (+ autograd) for that one row. */
int mtmp_attn_cls_fwd(int dtype, const void* q, const void* k, const void* v, int ld_qkv, const void* res, int ld_res, void* o_cls,
                      void* r1_cls, float* lse, const int32_t* kv_len, const int32_t* row_start, int B, int N, int H, int cls_tok,
                      float scale, void* stream);
int mtmp_attn_cls_bwd(int dtype, const void* q, const void* k, const void* v, int ld_qkv, const void* o_cls, const void* d_o,
                      const float* lse, const int32_t* kv_len, const int32_t* row_start, void* dq, void* dk, void* dv, int ld_dqkv,
                      int B, int N, int H, int cls_tok, float scale, void* stream);
/* out[ceil(rows / 32)][H] = max over each 32-row block of ||k[row, 64h : 64h + 64]||_2 (k: [rows, ld]). */
long long mtmp_key_norms_floats(long long rows, int H);
int mtmp_key_norms(int dtype, const void* k, float* out, long long rows, int H, int ld, void* stream);

/* Backward of the above: dq,dk,dv [B,N,ld_dqkv] from d_o [B,N,ld_do]; o is the forward output.
 * delta_ws: float[B*H*N] scratch. */
int mtmp_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                  const float* lse, const int32_t* kv_len, void* dq, void* dk, void* dv, float* delta_ws, int B,
                  int N, int H, int ld_qkv, int ld_o, int ld_do, int ld_dqkv, float scale, void* stream);

/* y[M,N] = act(LN(x[M,256]) w[N,256]^T + bias): custom LayerNorm (module.py:138-144: unbiased
 * std, eps added to std) fused into the Q/K/V projections (attention.py:60-62,68-70; w = [Wq;Wk;Wv])
 * or the first FFN conv + ReLU (module.py:74-77).  gamma/beta/bias fp32; w in `dtype`.
 * N % 64 == 0 (bf16) / N % 32 == 0 (fp32): features are walked in whole weight panels.
 * Optional outputs: xn[M,256] = LN(x) (dtype), stats[M,2] = (mean, 1/(std+eps)).
 * drop_p > 0 applies nn.Dropout (module.py:77-79 drop1) to the activated output with the
 * counter-based mask keep(seed ^ *seed_dev, row*N+col); the backward regenerates it (mtmp_dropout_bwd).
 * seed_dev (may be NULL) is a device word the host advances every step, so that a captured hipGraph
 * (whose scalar arguments are frozen) still draws fresh masks on every replay. */
int mtmp_ln_gemm(int dtype, const void* x, const float* gamma, const float* beta, const void* w, const float* bias,
                 void* y, void* xn, float* stats, int M, int N, int ldx, int ldy, float eps, int relu, float drop_p,
                 unsigned seed, const unsigned* seed_dev, void* stream);

/* mtmp_ln_gemm for the Q/K/V projection of an encoder layer (w = [Wq; Wk; Wv], N = 768, ldy = 768, no activation) that
 * also fills the key-norm table of mtmp_attn_fwd from its epilogue: key_norms[ceil(M / 32)][4] (mtmp_key_norms_floats(M, 4)
 * floats).  Replaces module.py:138-144 + attention.py:68-70. */
int mtmp_ln_gemm_qkv(int dtype, const void* x, const float* gamma, const float* beta, const void* w, const float* bias,
                     void* y, void* xn, float* stats, float* key_norms, int M, int ldx, float eps, void* stream);

/* y[M,N] = drop(act(a[M,K] w[N,K]^T + bias)) (+ res[M,N]); K % 8 == 0, N % 32 == 0; res must not alias y.
 * Second FFN conv + drop2 + residual (module.py:78-80, encoder.py:32).  With gate[M,N] != NULL the
 * result is gated: y = gate > 0 ? y * gate_scale : 0 -- the backward of ReLU (+ drop1) applied to
 * dH = dY W2 in the same pass (gate = the stored post-dropout activations). */
/* act: 0 none, 1 ReLU, 2 exact GELU; K % 8 == 0 (a 64-wide chunk's tail reads as zero);
 * row_scale[M / rows_per_scale] (may be NULL) multiplies row blocks before the residual add
 * (row-mode StochasticDepth of the Swin blocks). */
int mtmp_gemm_nt(int dtype, const void* a, const void* w, const float* bias, const void* res, void* y, int M, int N,
                 int K, int lda, int ldy, int ldr, int act, float drop_p, unsigned seed, const unsigned* seed_dev,
                 const void* gate, float gate_scale, const float* row_scale, int rows_per_scale, void* stream);

/* Sign bits of a ReLU projection (bf16 build, N % 64 == 0): one bit per output, "y > 0", in the row-panel kernels' private
 * order; mtmp_sign_bits_bytes(M, N) bytes (= M N / 8).
 *   mtmp_ln_gemm_signs  = mtmp_ln_gemm with relu = 1 (module.py:74-77 + drop1) that also writes them;
 *   mtmp_gemm_nt_signs  : y[M,N] = signs ? (a[M,256] w[N,256]^T) * gate_scale : 0 -- the FFN backward's dH = dY W2 taken
 *     through the ReLU + drop1 of module.py:77-79 (autograd) without re-reading the M x N hidden activation, whose sign is
 *     all that step needs: 4 MB of gate traffic per launch at config 2 instead of 132 MB.
 * Other dtypes return MTMP_ERR_ARG (the fp32 build gates on the stored activation through mtmp_gemm_nt). */
long long mtmp_sign_bits_bytes(int M, int N);
int mtmp_ln_gemm_signs(int dtype, const void* x, const float* gamma, const float* beta, const void* w, const float* bias,
                       void* y, void* xn, float* stats, int M, int N, int ldx, int ldy, float eps, float drop_p,
                       unsigned seed, const unsigned* seed_dev, void* signs, void* stream);
int mtmp_gemm_nt_signs(int dtype, const void* a, const void* w, void* y, int M, int N, int lda, int ldy,
                       const void* signs, float gate_scale, void* stream);
/* mtmp_gemm_nt_signs with the backward of a dropout on its A operand folded in (autograd of module.py:78-80 in front of dH):
 * a' = keep ? a / (1 - drop_p) : 0 with mtmp_dropout_bwd's mask for (seed, seed_dev, drop_p) on the contiguous [M,256] tensor,
 * y = signs ? (a' w^T) * gate_scale : 0, a_out [M,256] (may be NULL) = a' (the weight-gradient product's operand). */
int mtmp_gemm_nt_signs_drop(int dtype, const void* a, const void* w, void* y, int M, int N, int lda, int ldy,
                            const void* signs, float gate_scale, float drop_p, unsigned seed, const unsigned* seed_dev,
                            void* a_out, void* stream);

/* Weight / bias gradient of the Linear and k=1 Conv1d layers (attention.py:60-62, module.py:74-78):
 * dw[N,K] (fp32) = dy[M,N]^T x[M,K];  db[N] (fp32, may be NULL) = column sums of dy.
 * N % 128 == 0, K % 128 == 0; the contraction runs over the M tokens (split over workgroups,
 * partial slabs in ws, one reduce pass).  ws: mtmp_gemm_tn_ws_floats(M,N,K) floats. */
long long mtmp_gemm_tn_ws_floats(int M, int N, int K);
int mtmp_gemm_tn(int dtype, const void* dy, const void* x, float* dw, float* db, float* ws, int M, int N, int K,
                 int ldy, int ldx, void* stream);
/* The same with rows_live (may be NULL): a DEVICE word with the rows in use (<= M) -- a packed token stream, see the grouped
 * forms below; split count, workspace and grid stay those of M, the splits share the live rows. */
int mtmp_gemm_tn_live(int dtype, const void* dy, const void* x, float* dw, float* db, float* ws, int M, int N, int K,
                      int ldy, int ldx, const int32_t* rows_live, void* stream);

/* g_out[i] = keep(seed,i) ? g_in[i]/(1-p) : 0 over n contiguous elements: backward of the epilogue
 * dropout above (n = M*N of that call, n % 4 == 0). */
int mtmp_dropout_bwd(int dtype, const void* g_in, void* g_out, long long n, unsigned seed, const unsigned* seed_dev,
                     float p, void* stream);

/* Backward of the custom LayerNorm (module.py:138-144), plus the residual-branch gradient:
 * dz[M,256] = LNbwd(dy[M,256]; z, stats, gamma) (+ d_res); dgamma_dbeta float[512] overwritten.
 * ws: mtmp_ln_bwd_ws_floats(M) floats. */
int mtmp_ln_bwd_ws_floats(int M);
int mtmp_ln_bwd(int dtype, const void* z, int ldz, const float* stats, const float* gamma, const void* dy,
                const void* d_res, int ldr, void* dz, float* dgamma_dbeta, float* ws, int M, float eps, void* stream);

/* The dX product of a LayerNorm-fed projection fused with that LayerNorm's backward -- autograd of
 * module.py:138-144 in front of attention.py:68-70 (K = 768) / module.py:74-77 (K = 1024):
 * dz[M,256] = LNbwd(dy[M,K] wt[256,K]^T; z, stats, gamma) (+ d_res); dgamma_dbeta float[512] overwritten.
 * wt = W^T of y = LN(z) W^T, K-contiguous; the M x 256 product never goes to HBM.
 * ws: mtmp_gemm_lnbwd_ws_floats(M) floats. */
int mtmp_gemm_lnbwd_ws_floats(int M);
int mtmp_gemm_lnbwd(int dtype, const void* dy, const void* wt, const void* z, int ldz, const float* stats,
                    const float* gamma, const void* d_res, int ldr, void* dz, float* dgamma_dbeta, float* ws, int M,
                    int K, int ldy, float eps, void* stream);

/* TIE/UMSE event embedding (tri_mbt_vsltcls.py:59-71,183-190):
 * out[n,256] = ReLU(LN(value*w_v+b_v)) + ReLU(LN(time*w_t+b_t)) + ftab[feature].
 * events float[n,3] = (time, value, feature index); params float[8][256] = ie_vslt.{0.weight,0.bias,
 * 1.weight,1.bias} then ie_time.{same}; ftab float[20][256] = ie_feat.weight. */
int mtmp_tie_embed_fwd(int dtype, const float* events, const float* params, const float* ftab, void* out, int n,
                       void* stream);
/* grads float[28][256] overwritten (8 parameter vectors in `params` order, then d ftab). */
int mtmp_tie_bwd_ws_floats(int n);
int mtmp_tie_embed_bwd(int dtype, const float* events, const float* params, const void* d_out, float* grads,
                       float* ws, int n, void* stream);

/* Time + modality-id embedding added to every image / text token (tri_mbt_vsltcls.py:216-224):
 * out[n,256] = ReLU(LN(time*w_t+b_t)) + ftab[feature], i.e. the event embedding without its value chain;
 * same events / params / grads layout as above (ws: mtmp_tie_bwd_ws_floats(n)). */
int mtmp_time_embed_fwd(int dtype, const float* events, const float* params, const float* ftab, void* out, int n,
                        void* stream);
int mtmp_time_embed_bwd(int dtype, const float* events, const float* params, const void* d_out, float* grads,
                        float* ws, int n, void* stream);

/* The same on the PACKED (ragged) batch layout of the collate (replaces the zero-padded
 * [B, TIE_len, 3] tensor of dataset_new.py:2177 + the [:, :max_len] trim of trainer.py:41-42):
 * events float[cu[B]][3] back to back, cu_seqlens int32[B+1] (device), out / d_out [B, t_pad, 256];
 * output rows past a sample's length are zero and receive no gradient.
 * ws: mtmp_tie_bwd_ws_floats(B * t_pad) floats. */
int mtmp_tie_embed_packed_fwd(int dtype, const float* events, const int32_t* cu_seqlens, int B, int t_pad,
                              const float* params, const float* ftab, void* out, void* stream);
int mtmp_tie_embed_packed_bwd(int dtype, const float* events, const int32_t* cu_seqlens, int B, int t_pad,
                              const float* params, const void* d_out, float* grads, float* ws, void* stream);

/* Stream input of the fusion encoder (mbt_encoder.py:697-729 and the concatenation of :745) in one launch:
 * out[b] = [ bott (nb rows) | dropout(LN(cls) + pe[0]) | dropout(LN(x[b,t]) + pe[t+1]), t = 0..N-1 ],
 * nn.LayerNorm semantics (biased variance, eps inside the root) in fp32; x, out, dz, dx in `dtype`;
 * cls / gamma / beta float[256], pe float[>= N+1][256] or NULL, bott float[nb][256] (nb <= 4);
 * stats float[B*(N+1)][2] is written by the forward and read by the backward; dropout as in mtmp_dropout_bwd
 * (seed ^ *seed_dev). grads float[7][256] = dgamma, dbeta, dcls, dbott[0..3], overwritten.
 * ws: mtmp_stream_input_ws_floats(B * (nb+1+N)) floats.
 * row_start / kv_len (int32 device, NULL together): PACKED output -- row r < kv_len[b] of sample b is written to row
 * row_start[b] + r of `out` (read from there in dz), the other rows are not written; x / dx keep the padded [B,N,256]
 * layout (dx rows of pad events are written as zeros). */
int mtmp_stream_input_ws_floats(int rows);
int mtmp_stream_input_fwd(int dtype, const void* x, const float* cls, const float* gamma, const float* beta,
                          const float* pe, const float* bott, void* out, float* stats, int B, int N, int nb, float eps,
                          float p, unsigned seed, const unsigned* seed_dev, const int32_t* row_start, const int32_t* kv_len,
                          void* stream);
int mtmp_stream_input_bwd(int dtype, const void* dz, const void* x, const float* cls, const float* gamma,
                          const float* stats, void* dx, float* grads, float* ws, int B, int N, int nb, float p,
                          unsigned seed, const unsigned* seed_dev, const int32_t* row_start, const int32_t* kv_len,
                          void* stream);

/* Classification head (tri_mbt_vsltcls.py:59-76 ie_demo, :248-255): out[b] = fc3(ReLU(BatchNorm1d(fc0([LN(cls[b]) |
 * ReLU(LN(ie_demo.0(age, gender)))])))), fp32, B <= 256, six launches forward + backward instead of ~65 torch kernels.
 * params: 14 device pointers (float): ie_demo.0.weight[256][2], ie_demo.0.bias, ie_demo.1.weight, ie_demo.1.bias,
 * layer_norms_after_concat.{weight,bias}, fc_list.0.{weight[256][512],bias}, fc_list.1.{weight,bias,running_mean,
 * running_var} (the running statistics are updated in place when training != 0), fc_list.3.{weight[256],bias[1]}.
 * ws_fwd: mtmp_head_ws_floats(B) floats, written by the forward and read by the backward.
 * backward outputs (overwritten): dcls[B][256]; g_rows[7][256] = d layer_norms_after_concat.{weight,bias}, d ie_demo.1.{weight,
 * bias}, d ie_demo.0.weight[:,0], [:,1], d ie_demo.0.bias; dw1[256][512]; g_feat[4][256] = d fc_list.0.bias, d fc_list.1.{weight,
 * bias}, d fc_list.3.weight; db2[1].  ws_bwd: B*256*8 floats. */
int mtmp_head_ws_floats(int B);
int mtmp_head_fwd(const float* cls, const float* age, const float* gender, const void* const* params, float* out, float* ws,
                  int B, float ln_eps, float bn_eps, float momentum, int training, void* stream);
int mtmp_head_bwd(const float* d_out, const float* cls, const float* age, const float* gender, const void* const* params,
                  const float* ws_fwd, float* dcls, float* g_rows, float* dw1, float* g_feat, float* db2, float* ws_bwd, int B,
                  float ln_eps, int training, void* stream);
/* The head with the CLS vectors in the fusion stack's own type (ABI 6): cls / dcls [B][256] in cls_dtype (no cast launch on either
 * side of the head), num_batches_tracked (device int64, may be NULL) incremented by the first launch, and every parameter's
 * gradient written to a destination of its own: dst[12] in the order of the 12 trained parameters of `params` (ie_demo.0.weight
 * [256][2], ie_demo.0.bias, ie_demo.1.{weight,bias}, layer_norms_after_concat.{weight,bias}, fc_list.0.{weight,bias},
 * fc_list.1.{weight,bias}, fc_list.3.{weight,bias}) -- slices of the flat gradient buffer, or scratch.  Same math, workspaces and
 * limits as mtmp_head_fwd / mtmp_head_bwd. */
int mtmp_head_fwd_t(int cls_dtype, const void* cls, const float* age, const float* gender, const void* const* params, float* out,
                    float* ws, int B, float ln_eps, float bn_eps, float momentum, int training, long long* num_batches_tracked,
                    void* stream);
int mtmp_head_bwd_scatter(int cls_dtype, const float* d_out, const void* cls, const float* age, const float* gender,
                          const void* const* params, const float* ws_fwd, void* dcls, float* const* dst, float* ws_bwd, int B,
                          float ln_eps, int training, void* stream);

/* BCEWithLogitsLoss(reduction="mean") (2_train.py:76, trainer.py:128): loss[0] = mean_b [max(o,0) - o t + log1p(exp(-|o|))],
 * dlogit[b] = (sigmoid(o_b) - t_b) / n; logits / target / dlogit float[n]. */
int mtmp_bce_logits_mean(const float* logits, const float* target, float* loss, float* dlogit, int n, void* stream);

/* Stream-ordered time stamp: a one-lane kernel on `stream` stores the 100 MHz wall clock into *slot.  Also works inside a
 * captured hipGraph, where HIP events cannot be timed -- bench.py brackets the roofline kernel of the replayed steps with it. */
int mtmp_timestamp(unsigned long long* slot, void* stream);

/* Swin-T patch-embedding stem: Conv2d(1,96,4,stride 4) -> NHWC -> LayerNorm(96)
 * (builder/models/src/swin_transformer.py:559-567,646) as an implicit GEMM.
 * img float[n_img,1,H,W]; out [n_img,H/4,W/4,96] in `dtype`. */
int mtmp_swin_stem_fwd(int dtype, const float* img, const float* w, const float* bias, const float* ln_w,
                       const float* ln_b, void* out, int n_img, int H, int W, void* stream);

/* nn.LayerNorm(C) over rows (Swin norm1/norm2/final norm, swin_transformer.py:428-449,611-612);
 * merge != 0 fuses the 2x2 patch-merging gather of swin_transformer.py:34-44: x is [n,H,W,C/4],
 * rows = n*(H/2)*(W/2).  w,b fp32; C even, <= 1536. */
int mtmp_layernorm_rows(int dtype, const void* x, const float* w, const float* b, void* y, long long rows, int C,
                        float eps, int merge, int H, int W, void* stream);

/* y[M,N] = LayerNorm(x[M,C]; ln_w, ln_b, eps) W[N,C]^T + bias: norm1 + the qkv projection of a Swin block in one launch
 * (swin_transformer.py:428-449 + :115-225; replaces mtmp_layernorm_rows + mtmp_gemm_nt for the narrow stages).
 * bf16 only (dtype 1), C = 96 or 192, N % 32 == 0; W bf16; ln_w, ln_b, bias fp32 (bias may be NULL). */
int mtmp_swin_ln_linear(int dtype, const void* x, const float* ln_w, const float* ln_b, const void* w, const float* bias,
                        void* y, long long M, int C, int N, float eps, void* stream);

/* MLP half of a Swin block in one launch (swin_transformer.py:428-449, torchvision MLP keys mlp.0 / mlp.3):
 * y[M,C] = x + row_scale[row / rows_per_scale] * (gelu(LayerNorm(x; ln_w, ln_b, eps) W1^T + b1) W2^T + b2).
 * Replaces mtmp_layernorm_rows + mtmp_gemm_nt(act = GELU) + mtmp_gemm_nt(residual, row_scale) for the narrow stages, whose
 * 4C-wide hidden activation then never reaches HBM.  bf16 only (dtype 1), C = 96 or 192; w1 [4C,C], w2 [C,4C] bf16;
 * ln_w, ln_b, b1, b2 fp32; row_scale (per-image StochasticDepth factor) may be NULL; y must not alias x. */
int mtmp_swin_mlp(int dtype, const void* x, const float* ln_w, const float* ln_b, const void* w1, const float* b1,
                  const void* w2, const float* b2, const float* row_scale, int rows_per_scale, void* y, long long M, int C,
                  float eps, void* stream);

/* Shifted-window attention (swin_transformer.py:115-225, V1 branch) on the un-shifted NHWC map:
 * qkv [n,H,W,3C] -> out [n,H,W,C]; window 7x7, head_dim 32, H % 7 == W % 7 == 0.
 * table [4 window types][heads][64][64] in `dtype` = relative-position bias + shift mask
 * (type = 2*(last window row) + (last window col), 0 when shift == 0), -30000 on pad keys. */
int mtmp_swin_window_attn(int dtype, const void* qkv, const void* table, void* out, int n_img, int H, int W, int C,
                          int heads, int shift, float scale, void* stream);

/* Fused AdamW (2_train.py:110, torch.optim.AdamW math) over flat fp32 buffers of n elements
 * (n % 4 == 0); optional bf16 shadow copy of the parameters; grad is multiplied by grad_scale
 * first (1/world_size after an RCCL all-reduce SUM). */
int mtmp_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* bf16_shadow,
                    long long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    float grad_scale, void* stream);

/* Bottleneck exchange between the modality streams (mbt_encoder.py:764-779), in place on the stream
 * buffers z_m [B, n_m, 256] whose rows 0..3 are the four bottleneck tokens (n_m = 4 + 1 + tokens):
 * rows 0..3 of all three become the per-sample weighted mean over the present modalities
 * (missing[b] 0: vslt+img+txt, 1: vslt+img, 2: vslt+txt, 3: vslt) -- with resbottle != 0 averaged with
 * prev [B,4,256] fp32 (mbt_encoder.py:741-742,778-779).  keep (may be NULL) receives the fp32 result.
 * _bwd transforms the gradient buffers dz_m the same way in place: rows 0..3 hold the consumers'
 * gradients on entry and each stream's own bottleneck-output gradient on exit; d_prev_in (may be NULL) /
 * d_prev_out carry the residual path between consecutive exchanges.
 * z_t / dz_t may be NULL: the two-stream encoder (BimodalTransformerEncoder_MBT, mbt_encoder.py:519-634), whose
 * patterns {0: mean of both, 1: stream 0} the caller passes as rows 1 and 3.
 * row_start_v (int32[B] device, may be NULL): the first buffer is PACKED, its sample b starts at row row_start_v[b]. */
int mtmp_bottleneck_exchange_fwd(int dtype, void* z_v, void* z_i, void* z_t, int B, int n_v, int n_i, int n_t,
                                 const long long* missing, int resbottle, const float* prev, float* keep,
                                 const int32_t* row_start_v, void* stream);
int mtmp_bottleneck_exchange_bwd(int dtype, void* dz_v, void* dz_i, void* dz_t, int B, int n_v, int n_i, int n_t,
                                 const long long* missing, int resbottle, const float* d_prev_in, float* d_prev_out,
                                 const int32_t* row_start_v, void* stream);

/* Grouped forms of the fusion layer's row-wise kernels (bf16): the same operation of up to three token streams (vital signs /
 * image / text) in ONE launch -- see mtmp_attn_fwd_grouped.  All pointer / int arrays are HOST arrays of n entries (1 <= n <= 3),
 * scalars are common to the streams; same kernels and results as n calls of the single forms.  The parity (fp32) build keeps
 * the single forms (these return an error for dtype 0).
 *   mtmp_ln_gemm_qkv_grouped        = mtmp_ln_gemm_qkv per stream
 *   mtmp_ln_gemm_signs_grouped      = mtmp_ln_gemm_signs per stream (one dropout seed per stream)
 *   mtmp_gemm_nt_grouped            = mtmp_gemm_nt per stream without gate / row scale (FFN2 + drop2 + residual)
 *   mtmp_gemm_nt_signs_drop_grouped = mtmp_gemm_nt_signs_drop per stream
 *   mtmp_gemm_lnbwd_grouped         = mtmp_gemm_lnbwd per stream, partial slabs only (reduce with mtmp_reduce_batch)
 *   mtmp_gemm_tn_grouped            = mtmp_gemm_tn per stream on the LDS-DMA kernel, partial slabs only: ws[i] holds
 *                                     splits[i] x (N K + N) floats with splits from mtmp_gemm_tn_group_plan (non-zero return:
 *                                     these shapes have no grouped form -- call mtmp_gemm_tn per stream)
 * rows_live (may be NULL as a whole or per entry): rows_live[i] is a DEVICE word holding the rows of stream i that are in use
 * this step (<= M[i]) -- the packed vital-sign stream (mtmp_row_starts), whose buffers and launch grids keep the padded size
 * M[i] so that a captured hipGraph replays for any lengths: rows past it are neither read nor written, workgroups that own
 * none return at once (their gradient partials count as zeros), and mtmp_gemm_tn_grouped's splits share the live rows. */
int mtmp_ln_gemm_qkv_grouped(int dtype, int n, const void* const* x, const float* const* gamma, const float* const* beta,
                             const void* const* w, const float* const* bias, void* const* y, void* const* xn, float* const* stats,
                             float* const* key_norms, const int* M, const int* ldx, float eps, const int32_t* const* rows_live,
                             void* stream);
int mtmp_ln_gemm_signs_grouped(int dtype, int n, const void* const* x, const float* const* gamma, const float* const* beta,
                               const void* const* w, const float* const* bias, void* const* y, void* const* xn, float* const* stats,
                               void* const* signs, const int* M, int N, const int* ldx, float eps, float drop_p,
                               const unsigned* seeds, const unsigned* seed_dev, const int32_t* const* rows_live, void* stream);
int mtmp_gemm_nt_grouped(int dtype, int n, const void* const* a, const void* const* w, const float* const* bias,
                         const void* const* res, void* const* y, const int* M, int N, int K, const int* lda, const int* ldy,
                         const int* ldr, int act, float drop_p, const unsigned* seeds, const unsigned* seed_dev,
                         const int32_t* const* rows_live, void* stream);
int mtmp_gemm_nt_signs_drop_grouped(int dtype, int n, const void* const* a, const void* const* w, void* const* y, const int* M, int N,
                                    const int* lda, const void* const* signs, float gate_scale, float drop_p, const unsigned* seeds,
                                    const unsigned* seed_dev, void* const* a_out, const int32_t* const* rows_live, void* stream);
int mtmp_gemm_lnbwd_grouped(int dtype, int n, const void* const* dy, const void* const* wt, const void* const* z, const int* ldz,
                            const float* const* stats, const float* const* gamma, const void* const* d_res, const int* ldr,
                            void* const* dz, float* const* ws, const int* M, int K, const int* ldy, float eps,
                            const int32_t* const* rows_live, void* stream);
int mtmp_gemm_tn_group_plan(int n, const int* M, int N, int K, int* splits_out);
int mtmp_gemm_tn_grouped(int dtype, int n, const void* const* dy, const void* const* x, float* const* ws, const int* M, int N, int K,
                         const int* ldy, const int* ldx, const int* splits, const int32_t* const* rows_live, void* stream);

/* Deferred reductions: mtmp_gemm_tn with dw == NULL (and db == NULL) and mtmp_gemm_lnbwd with dgamma_dbeta == NULL leave their
 * partial slabs in ws -- [mtmp_gemm_tn_slab_rows(dtype,M,N,K)][N K + N] and [mtmp_gemm_lnbwd_slab_rows(M)][512] floats -- and
 * mtmp_reduce_batch sums up to 8 such slabs in ONE launch: out_a[i][c] = sum_r slab[i][r][c] for c < split[i], the remaining
 * columns go to out_b[i] (may be NULL).  The gradient reductions of one encoder layer's backward (autograd of attention.py:60-62,
 * module.py:74-78, :138-144): seven launches become one.  All arrays are HOST arrays of n entries. */
int mtmp_gemm_tn_slab_rows(int dtype, int M, int N, int K);
int mtmp_gemm_lnbwd_slab_rows(int M);
int mtmp_reduce_batch(const float* const* slab, const int* rows, const long long* cols, float* const* out_a, const long long* split,
                      float* const* out_b, int n, void* stream);

/* ---- The input nodes' backward with ONE closing launch (ABI 6).  The backward of the three stream-input launches and of the event /
 * time embeddings (autograd of mbt_encoder.py:697-729,745 and tri_mbt_vsltcls.py:183-190,216-224) is the tail of a training step,
 * where nothing else runs; each of them used to end in two reduction levels + a multi-tensor copy into the flat gradient, and the
 * parameters the nodes share (ie_time, ie_feat, the bottleneck tokens) in accumulation launches.
 * mtmp_stream_input_bwd_partials = mtmp_stream_input_bwd without the reduction: ws receives
 *   [mtmp_stream_input_slab_rows(B * (nb+1+N))][7][256] floats of partial sums (dgamma, dbeta, dcls, dbott[0..3]).
 * mtmp_tie_time_embed_bwd_partials = mtmp_tie_embed_bwd of events[n][3] / d_out[n][256] AND mtmp_time_embed_bwd of
 *   time_events[n_time][3], whose gradient rows are d_time_a[n_time_a][256] followed by d_time_b[n_time - n_time_a][256], as one
 *   launch: ws receives [mtmp_tie_bwd_slab_rows(n + n_time)][28][256] floats (grads layout of mtmp_tie_embed_bwd).
 * mtmp_reduce_scatter: out[i][c] = sum_{r < rows[i]} src[i][r * ld[i] + c], c < cols[i], for up to 12 column ranges of such slabs
 *   in one launch (src[i] = first column of the range inside its slab, ld[i] = the slab's row length in floats; HOST arrays) --
 *   every destination is a parameter's slice of the flat gradient buffer, or scratch. */
int mtmp_stream_input_slab_rows(int rows);
int mtmp_stream_input_bwd_partials(int dtype, const void* dz, const void* x, const float* cls, const float* gamma, const float* stats,
                                   void* dx, float* ws, int B, int N, int nb, float p, unsigned seed, const unsigned* seed_dev,
                                   const int32_t* row_start, const int32_t* kv_len, void* stream);
/* mtmp_stream_input_bwd_partials of up to three token streams as ONE launch: ws receives the streams' slabs back to back (stream i's
 * mtmp_stream_input_slab_rows(B[i] * (nb[i]+1+N[i])) rows behind those in front of it), so one mtmp_reduce_scatter entry over all
 * rows sums the bottleneck tokens' columns (768..) of all streams.  HOST arrays of n <= 3 entries; row_start / kv_len may be NULL
 * as a whole or per entry. */
int mtmp_stream_input_bwd_grouped(int dtype, int n, const void* const* dz, const void* const* x, const float* const* cls,
                                  const float* const* gamma, const float* const* stats, void* const* dx, float* ws, const int* B,
                                  const int* N, const int* nb, const float* p, const unsigned* seed, const unsigned* seed_dev,
                                  const int32_t* const* row_start, const int32_t* const* kv_len, const void* const* add,
                                  const int* add_L, void* stream);
/* mtmp_stream_input_fwd with `add` [B N / add_L][256] (dtype; may be NULL): row (b N + t) / add_L is added to input token (b, t) in
 * front of the LayerNorm, the sum rounded to dtype -- the time + modality embedding of the token's image / report
 * (tri_mbt_vsltcls.py:216-224) without a torch add launch or a second copy of the tokens; mtmp_stream_input_bwd_grouped takes the
 * same add / add_L (per stream, NULL entries allowed) and recomputes the sum; x then is the tensor WITHOUT the add. */
int mtmp_stream_input_fwd_add(int dtype, const void* x, const float* cls, const float* gamma, const float* beta, const float* pe,
                              const float* bott, void* out, float* stats, int B, int N, int nb, float eps, float p, unsigned seed,
                              const unsigned* seed_dev, const int32_t* row_start, const int32_t* kv_len, const void* add,
                              int add_L, void* stream);
/* out[i][j][:] = sum_{t < L[i]} dx[i][j L[i] + t][:] (256 columns, fp32 accumulation, `dtype` in and out) for n <= 2 tensors in one
 * launch: the gradient of the per-image / per-report time embedding that was added to each of its L tokens
 * (tri_mbt_vsltcls.py:216-224) -- torch autograd's sum over the token axis.  rows[i] = output rows of tensor i.  HOST arrays. */
int mtmp_token_sums(int dtype, int n, const void* const* dx, void* const* out, const int* rows, const int* L, void* stream);
int mtmp_tie_bwd_slab_rows(int n);
int mtmp_tie_time_embed_bwd_partials(int dtype, const float* events, int n, const float* time_events, int n_time, int n_time_a,
                                     const float* params, const void* d_out, const void* d_time_a, const void* d_time_b, float* ws,
                                     void* stream);
int mtmp_reduce_scatter(const float* const* src, const int* rows, const long long* ld, const long long* cols, float* const* out, int n,
                        void* stream);

/* pair[0] <- bit pattern of *value (fp32), pair[1] += 1 (both uint32, device memory): a scalar and a sequence number published
 * together.  The training step copies the pair to pinned host memory right behind its loss (2_train.py:76, trainer.py:128), so the
 * reference's `loss.item()` hands the value over when the forward pass is done instead of when the whole step has drained. */
int mtmp_publish_scalar(const float* value, unsigned* pair, void* stream);

/* Up to 16 device-to-device copies in one launch: dst[i] <- src[i] (bytes[i] bytes, both 16-byte aligned); round16[i] != 0: the
 * buffer holds fp32 values and is written as float(half(x)) -- the fp16 round trip the reference applies to its event / time
 * inputs (trainer.py:26-27, 2_train.py:164).  All arrays are HOST arrays of n entries, read at launch time. */
int mtmp_copy_batch(const void* const* src, void* const* dst, const long long* bytes, const int* round16, int n, void* stream);

/* Batched 2-D transposes in one launch: dst[i] [cols[i]][rows[i]] = src[i] [rows[i]][cols[i]]^T, elements of elem_bytes = 2 | 4.
 * src / dst / rows / cols are HOST arrays of n entries (read at launch time; the pointers travel in the kernel arguments).
 * The K-contiguous backward operands (W2^T, Wqkv^T, W1^T: autograd of attention.py:60-62 / module.py:74-78) of all encoder
 * blocks, rebuilt after every optimizer step. */
int mtmp_transpose_batch(int elem_bytes, const void* const* src, void* const* dst, const int* rows, const int* cols, int n,
                         void* stream);

/* Valid-key counts of the three streams in one launch (mbt_encoder.py:703-714): plain = len + 1 (CLS), stream txt_idx's
 * 3 -> 0; fused = plain + n_bott.  len_*: int64 [B] or NULL (unmasked stream: rows left untouched); out: int32 [2][3][B]. */
int mtmp_stream_lengths(const long long* len_v, const long long* len_i, const long long* len_t, int* out, int B, int n_bott,
                        int txt_idx, void* stream);
/* The frozen image encoder on a batch in which some samples have no image.  Their encoder output is read by nothing (the
 * bottleneck exchange gives the image stream weight 0 for missing_num 2 / 3, mbt_encoder.py:764-779; the reference pushes a zero
 * image through all of swin_transformer.py:503-654 for them).  mtmp_image_slots moves the present images to the front of the
 * encoder's batch and tabulates the rows in use per encoder stage; the *_live forms of the encoder's kernels take one word of that
 * table as `rows_live` (DEVICE pointer, may be NULL = all rows): buffers and grids keep the size of the whole batch, so a captured
 * hipGraph replays for any number of present images; rows past the live ones are neither read nor written.
 *   pattern: int64[B] device, the step's missing_num ids; sample b has an image iff 0 <= pattern[b] < present_below (2 for the
 *   tri-modal ids: 0 all three, 1 vslt + image).  out: int32[2 B + 16] device:
 *     out[i], i < B            slot i of the encoder's batch works on image out[i] (present images first, in batch order)
 *     out[B + b]               the slot of sample b's image, or B (a slot the caller keeps zero) when it has none
 *     out[2 B]                 number of present images
 *     out[2 B + 1 + 5 p + s]   rows in use at stage s = 0..3 (hw0 >> 2 s rows per image; s = 4: one row per image), p = 0: the whole
 *                              batch, p = 1 / 2: its first / second half of B / 2 slots (the encoder's two-stream tail)
 * mtmp_swin_stem_fwd_live: `order` (= out, may be NULL) maps slot -> image.  mtmp_swin_window_attn_live: rows_live counts token
 * rows (live images = *rows_live / (H W)). */
int mtmp_image_slots(const long long* pattern, int present_below, int32_t* out, int B, int hw0, void* stream);
int mtmp_swin_stem_fwd_live(int dtype, const float* img, const float* w, const float* bias, const float* ln_w, const float* ln_b,
                            void* out, int n_img, int H, int W, const int32_t* order, const int32_t* rows_live, void* stream);
int mtmp_layernorm_rows_live(int dtype, const void* x, const float* w, const float* b, void* y, long long rows, int C, float eps,
                             int merge, int H, int W, const int32_t* rows_live, void* stream);
int mtmp_swin_ln_linear_live(int dtype, const void* x, const float* ln_w, const float* ln_b, const void* w, const float* bias,
                             void* y, long long M, int C, int N, float eps, const int32_t* rows_live, void* stream);
int mtmp_swin_mlp_live(int dtype, const void* x, const float* ln_w, const float* ln_b, const void* w1, const float* b1,
                       const void* w2, const float* b2, const float* row_scale, int rows_per_scale, void* y, long long M, int C,
                       float eps, const int32_t* rows_live, void* stream);
int mtmp_swin_window_attn_live(int dtype, const void* qkv, const void* table, void* out, int n_img, int H, int W, int C, int heads,
                               int shift, float scale, const int32_t* rows_live, void* stream);

/* ---- Backward of the image encoder's building blocks (ABI 5): the sibling models that TRAIN the Swin-T encoder -- the reference
 * calls self.img_encoder(img) without torch.no_grad() in builder/models/8_missing_models/bi_vsltimg_mbt_v1.py:203-206,
 * tri_mbt_v2.py:208-211, tri_mbt_vmulti.py:145 -- replace torch autograd of builder/models/src/swin_transformer.py as follows
 * (the GEMM-shaped parts, dX = dY W and dW = dY^T X, run on mtmp_gemm_nt / mtmp_gemm_tn):
 *
 * mtmp_layernorm_rows_bwd: autograd of nn.LayerNorm(C, eps) as mtmp_layernorm_rows computes it (:428-449 norm1 / norm2, :34-85
 *   the patch-merging norm, :611 the final norm).  x, dy, dx [rows, C] (dtype), w float[C]; slab float[slab_rows][2][C] receives one
 *   row of partial (dw | db) sums per workgroup -- the caller adds the rows (mtmp_layernorm_rows_bwd_slab_rows tells how many).
 * mtmp_gelu_fwd / mtmp_gelu_bwd: nn.GELU of the MLP (:437-439) as its own pass, and dx = dy GELU'(x); n % 8 == 0.
 * mtmp_swin_window_attn_bwd: autograd of mtmp_swin_window_attn (:115-225).  dout [n,H,W,C] -> dqkv [n,H,W,3C] (every element
 *   written once; H, W multiples of 7) and dtab float[4][heads][64][64], the gradient of the additive table, which the caller
 *   ZEROES first (float atomics). */
int mtmp_layernorm_rows_bwd_slab_rows(long long rows, int C);
int mtmp_layernorm_rows_bwd(int dtype, const void* x, const float* w, const void* dy, void* dx, float* slab, long long rows, int C,
                            float eps, void* stream);
int mtmp_gelu_fwd(int dtype, const void* x, void* y, long long n, void* stream);
int mtmp_gelu_bwd(int dtype, const void* x, const void* dy, void* dx, long long n, void* stream);
int mtmp_swin_window_attn_bwd(int dtype, const void* qkv, const void* table, const void* dout, void* dqkv, float* dtab, int n_img,
                              int H, int W, int C, int heads, int shift, float scale, void* stream);

/* Attention half of a Swin block in ONE launch (ABI 4): out = x + row_scale[image] * (proj(window_attention(qkv(norm1(x)))) + b_proj).
 * Replaces builder/models/src/swin_transformer.py:428-449 (norm1, attn, stochastic_depth, residual) with :115-225
 * (shifted_window_attention) for maps that are multiples of the 7x7 window.  bf16 only (dtype MTMP_BF16); x, out [n_img, H, W, C]
 * (out != x), C = 96 or 192, heads = C / 32; wqkv [3C][C], wproj [C][C] bf16; ln_w, ln_b [C], bqkv [3C], bproj [C] fp32; table
 * [4 window types][heads][64][64] bf16 as mtmp_swin_window_attn's, but with the KEY columns of every group of 16 keys in
 * accumulator-register order: position 8 h + j (h = 0, 1; j = 0..7) of group g holds key 16 g + (j & 3) + 8 (j >> 2) + 4 h.
 * row_scale: fp32[n_img] (row-mode StochasticDepth) or NULL; rows_live: NULL or the device word of mtmp_image_slots. */
int mtmp_swin_attn_block(int dtype, const void* x, const float* ln_w, const float* ln_b, float eps, const void* wqkv,
                         const float* bqkv, const void* table, const void* wproj, const float* bproj, const float* row_scale,
                         void* out, int n_img, int H, int W, int C, int heads, int shift, float scale, const int32_t* rows_live,
                         void* stream);
int mtmp_gemm_nt_live(int dtype, const void* a, const void* w, const float* bias, const void* res, void* y, int M, int N, int K,
                      int lda, int ldy, int ldr, int act, float drop_p, unsigned seed, const unsigned* seed_dev, const void* gate,
                      float gate_scale, const float* row_scale, int rows_per_scale, const int32_t* rows_live, void* stream);

/* Row map of a PACKED token stream ("skip padded key tiles and padded query rows"; the reference pads every sample to
 * the batch maximum, trainer.py:41-42, and masks keys, utils.py:79-125): out[b] = sum of min(max(kv_len[0..b), 0), n_max) =
 * sample b's first row when the samples' valid rows are stored back to back; out[B] = the rows in use (the `rows_live` word of
 * the grouped kernels); out[B + 1 .. 2 B + 1) = the order in which the attention kernels walk the samples of a packed stream
 * (ranked by length, dealt round-robin over the eight XCD chunks of their grids).  kv_len: int32[B] device (a fused count of
 * mtmp_stream_lengths); out: int32[2 B + 1] device -- the `row_start` argument of the packed forms is this whole array. */
int mtmp_row_starts(const int32_t* kv_len, int32_t* out, int B, int n_max, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MTMP_H */
