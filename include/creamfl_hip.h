/*
 * creamfl_hip.h -- C ABI of libcreamfl_hip.so, the MI355X (gfx950) implementation of the
 * CreamFL contrastive hot path.
 *
 * The reference (FLAIR-THU/CreamFL) is pure Python on PyTorch and has no FFI; the boundary
 * it would bind for this path is therefore designed here and each entry
 * point cites the reference code it replaces (paths relative to the reference checkout).
 * INTEGRATION.md shows the ctypes stub a maintainer would add on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; tensors are dense
 *     row-major fp32 unless stated; int64 index arrays are `const long long*`.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *     enqueued asynchronously on it; nothing synchronises, allocates or frees.
 *   - scratch memory is provided by the caller: cfl_*_ws_bytes() returns the size needed.
 *   - return value: 0 on success, a hipError_t (> 0) for a HIP failure, or a negative
 *     CFL_E* code for an argument error.  Nothing throws across the boundary.
 */
#ifndef CREAMFL_HIP_H
#define CREAMFL_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFL_EINVAL   (-1)   /* bad size / null pointer */
#define CFL_EALIGN   (-2)   /* pointer not 4-byte aligned */
#define CFL_ELIMIT   (-3)   /* size outside the supported range (documented per call) */

/* ---- library / profiling ------------------------------------------------------------- */
int         cfl_version(void);                 /* 100*major + minor */
const char* cfl_arch(void);                    /* "gfx950" */
int         cfl_num_kernels(void);
const char* cfl_kernel_name(int kernel_id);    /* name as it appears in rocprofv3 traces */
/* When enabled every kernel launch is bracketed by hipEvents recorded on the launch
 * stream; cfl_prof_query drains finished events (it synchronises on them). */
int cfl_prof_enable(int on);
int cfl_prof_select(int kernel_id);           /* time only this kernel (-1 = all): two hipEventRecords per launch cost host time */
int cfl_prof_reset(void);
int cfl_prof_query(int kernel_id, long long* launches, double* total_ms);

/* ---- A1: all-pairs soft-contrastive loss ------------------------------------------------
 * Replaces MCSoftContrastiveLoss.forward / _compute_loss / pairwise_sampling / full_sampling /
 * batchwise_cdist / soft_contrastive_nll  (src/criterions/probemb.py:7-86,150-256) for 2-D
 * features, uniform_lambda = vib_beta = 0.
 *
 * fwd: I, T [N, D]; a_dev/b_dev: device scalars negative_scale / shift (the criterion's learnable
 *      parameters stay on the device: no host sync per step).  d_ij = sqrt(|I_i - T_j|^2 + eps), s = -a d + b, m = +1 (i == j) / -1,
 *      NLL = softplus(-2 m s).  out[8] = { loss (= 2*(pos+neg)), pos, neg, dL/da, dL/db, 0,0,0 }
 *      (pos/neg are the one-direction sums: i2t_pos_loss == t2i_pos_loss == pos).
 *      If coef != NULL (a [2, N, N] buffer) it also writes coef[0] = (dL/dd_ij) / d_ij (both directions, not
 *      scaled by any upstream gradient) and coef[1] = its transpose, and leaves their row/column sums and the
 *      transposed features in ws for bwd.
 * bwd: dI = gout * (I * rowsum(coef) - coef @ T),  dT = gout * (T * colsum(coef) - coef^T @ I).
 *      gout_dev is a device scalar (the upstream gradient of the loss).
 * ws must be the same buffer for fwd and the following bwd.
 */
size_t cfl_pair_loss_ws_bytes(int N, int D);
int cfl_pair_loss_fwd(const float* I, const float* T, int N, int D, const float* a_dev,
                      const float* b_dev, float eps,
                      float* out8, float* coef, void* ws, void* stream);
int cfl_pair_loss_bwd(const float* I, const float* T, const float* coef, int N, int D,
                      const float* gout_dev, float* dI, float* dT, void* ws, void* stream);

/* ---- A3: inter-modal contrast against the frozen global bank ------------------------------
 * Replaces  logits = matmul(f, G.T)/0.5 ; CrossEntropyLoss()(logits, d_idx)
 * (src/algorithms/ClientTrainer.py:388,400-401,493-502; MMClientTrainer.py:194-201,301-308).
 *
 * fwd: F [B, D], G [M, D], idx [B] (int64, 0 <= idx < M).  lse[b] = logsumexp_m(inv_tau F_b.G_m),
 *      pos[b] = inv_tau F_b.G_idx[b] (exact fp32 dot), loss[0] = mean_b(lse - pos).
 *      If logits_t != NULL the scaled logits are stored TRANSPOSED, logits_t[M, B], for bwd.
 * bwd: dF[b,:] = coef * (sum_m softmax_bm G_m - G_idx[b]),  coef = inv_tau / B * gout.
 */
size_t cfl_bank_ws_bytes(int B, int M, int D);
int cfl_bank_lse_fwd(const float* F, const float* G, const long long* idx, int B, int M, int D,
                     float inv_tau, float* lse, float* pos, float* loss, float* logits_t,
                     void* ws, void* stream);
int cfl_bank_lse_bwd(const float* logits_t, const float* G, const long long* idx, const float* lse,
                     int B, int M, int D, float inv_tau, const float* gout_dev, float* dF,
                     void* ws, void* stream);

/* ---- A4: intra-modal (MOON style) contrast ------------------------------------------------
 * Replaces pos = sum(f*G_same[d_idx]), neg = sum(f*f_old), CE([pos,neg]/0.5, 0)
 * (ClientTrainer.py:404-414,458-468; MMClientTrainer.py:173-191,246-264).
 * loss[0] = mean_b softplus((neg-pos)*inv_tau);  dF_unit[b,:] = sigmoid(z_b)*inv_tau/B *
 * (Fold_b - G_same[idx_b])  (gradient for an upstream gradient of 1; may be NULL).
 * `B_div` is the CE mean divisor (B, or 2B when two modalities are stacked, MMClientTrainer.py:188).
 */
/* ---- A3 + A4 fused, single pass over the bank (csrc/bank_attn.hip: finish + backward; csrc/bank_gsplit.h: the bank pass) ----
 * Replaces the loop body src/algorithms/ClientTrainer.py:386-419 (and the inter-only :493-502 / intra-only :458-468
 * variants, and MMClientTrainer.py:173-206 with B_div = 2B) for D <= 768, D % 4 == 0 (cfl_bank_gsplit_supported); the entry
 * point is cfl_client_contrast_img_fwd below (the bank as a pre-split image):
 *   inter:  li = mean_b [ LSE_m(inv_tau F_b.G_other_m) - inv_tau F_b.G_other_idx[b] ]          (mode bit 0)
 *   intra:  lm = (1/B_div) sum_b softplus(inv_tau (F_b.Fold_b - F_b.G_same_idx[b]))             (mode bit 1)
 *   loss = (lm + li) w | (lm + li / (li/lm)) w with mode bit 2 (--loss_scale) | li | lm
 *   mode bit 3 (multi-modal client, MMClientTrainer.py:164-206 / :246-264 / :301-308): out5 holds on entry the terms of the
 *   OTHER modality (the previous call on this stream, same out5); li and lm become the totals over both modalities
 *   (loss_inter = loss_1_inter + loss_2_inter; loss_intra = CE over the stacked [2B, 2] logits, B_div = 2B in both calls)
 *   before the combination, and out5 = {loss, li_total, lm_total, c_inter, c_moon, loss} serves the backward of both.
 * G is streamed ONCE: the same pass accumulates softmax . G, so the gradient needs no second pass and no [B, M] tensor.
 * Logits use 3 x bf16-split MFMA (hi.hi + lo.hi + hi.lo, fp32 accumulation, |error| ~ 1e-6 on unit-norm features); the
 * positive dot and the intra term are exact fp32.
 * out5 (>= 6 floats) = {loss, li, lm, c_inter, c_moon, loss again}; dF_inter / dF_moon [B, D] are the UNIT gradients of li / lm
 * (want_grad), and
 * cfl_client_contrast_bwd writes dF = gout * (c_inter dF_inter + c_moon dF_moon).
 * lse [B] (required with bit 0), pos [B] (optional).  ws >= cfl_bank_gsplit_ws_bytes; `sync` points at an int that is 0
 * before the first call (left 0).  idx outside [0, M) contributes a zero positive (as cfl_bank_lse_fwd).
 */
/* Precision of the dense kernels that exist in two forms (cfl_pair_loss_*, cfl_bank_lse_*, cfl_conw_logprob):
 * 0 = 3 x bf16-split MFMA with fp32 accumulation (default; dot products of unit-norm rows to ~1e-6), 1 = exact fp32 MFMA.
 * Process-wide; initialised from the environment variable CFL_GEMM_EXACT. */
int cfl_get_exact_gemm(void);
int cfl_set_exact_gemm(int exact);

int cfl_client_contrast_bwd(const float* dF_inter, const float* dF_moon, const float* out5, const float* gout_dev, int B, int D,
                            float* dF, void* stream);

/* ---- A3 + A4 on a PRE-SPLIT bank image (round 3, csrc/bank_gsplit.h) -----------------------
 * The global bank is frozen while a client trains (ClientTrainer.py:369-372 takes the global features once per round and
 * every step of the round contrasts against them, :386-419), so its fp32 -> (bf16 hi, bf16 lo) split is done ONCE:
 *   cfl_bank_image_build(G [M, D] fp32) -> image of cfl_bank_image_bytes(M, D) bytes (= the fp32 bank's size, rows padded to 16
 *   and columns to 128 / 256): 16-row slots, [plane][row][column] bf16, 16-byte pieces XOR-swizzled -- the LDS image itself.
 * cfl_client_contrast_img_fwd = the fused step described above with the bank pass running on that image: 32 (64 beyond
 * D = 256) feature rows per workgroup, so a client batch of 128 needs 64 bank splits.  G_other (fp32) is still needed for the
 * exact positive rows G_other[idx]; with mode bit 0 clear (intra only) image_other / G_other may be NULL and only the finish
 * launch runs.  D <= 768, D % 4 == 0 (cfl_bank_gsplit_supported); ws >= cfl_bank_gsplit_ws_bytes.
 * (Rounds 1-3 kept two earlier generations of the bank pass as A/B references; round 4 keeps ONE: the exact-fp32 two-pass
 * kernels cfl_bank_lse_fwd / _bwd, which also serve D % 4 != 0 and D > 768.)
 * want_grad: 0 no gradient; 1 the two UNIT gradients (dF_inter, dF_moon; cfl_client_contrast_bwd combines them with the
 * coefficients of out5 and the upstream gradient); 2 (round 6, "direct") the FINAL gradient for an upstream gradient of 1 --
 * the loss is the root of loss.backward() as at ClientTrainer.py:420 / MMClientTrainer.py:207 -- written once into dF_inter
 * ([B, D]; dF_moon unused), no backward launch.  Only without mode bit 2 (--loss_scale makes the coefficients data-dependent:
 * CFL_EINVAL).
 */
size_t cfl_bank_image_bytes(int M, int D);
int cfl_bank_image_build(const float* G, int M, int D, void* image, void* stream);
int cfl_bank_gsplit_supported(int B, int M, int D);
size_t cfl_bank_gsplit_ws_bytes(int B, int M, int D, int want_grad);
int cfl_client_contrast_img_fwd(const float* F, const void* image_other, const float* G_other, const float* G_same,
                                const long long* idx, const float* F_old, int B, int M, int D, int B_div, float inv_tau,
                                float weight, int mode, int want_grad, float* out5, float* lse, float* pos, float* dF_inter,
                                float* dF_moon, void* ws, int* sync, void* stream);

size_t cfl_intra_ws_bytes(int B);
int cfl_intra_fwd(const float* F, const float* Gsame, const long long* idx, const float* Fold,
                  int B, int D, int M, int B_div, float inv_tau, float* loss, float* dF_unit, void* ws,
                  void* stream);   /* M = rows of Gsame: idx outside [0, M) contributes a zero positive */

/* ---- KD distillation term -------------------------------------------------------
 * Replaces  target = agg[d_idx, :]; loss += kd_weight * nn.MSELoss()(out, target)   (src/algorithms/MMFL.py:352-378).
 * loss[0] = weight * mean_{b,d} (out[b,d] - agg[idx[b],d])^2;  dout_unit = d loss / d out (may be NULL).
 * ws: cfl_intra_ws_bytes(B). */
int cfl_kd_mse(const float* out, const float* agg, const long long* idx, int B, int D, int M, float weight,
               float* loss, float* dout_unit, void* ws, void* stream);

/* ---- BERT text tower glue (row A2: src/networks/models/pcme.py:36-44 builds transformers' BertModel) ------------
 * The element-wise work between the GEMMs of a BertLayer (third-party `transformers`, restated in csrc/bertfuse.hip):
 *   daln : z = LayerNorm(dropout(g + bias) + residual)            BertSelfOutput / BertOutput
 *   bgelu: h = gelu(g + bias)   (exact erf form)                  BertIntermediate
 * g is the bias-free GEMM output.  All activations bf16 row-major [T, H] / [T, I]; gamma/beta fp32; bias bf16 or fp32
 * (bias_bf16 flag; NULL = none).  Dropout keep flags are a counter hash of (seed, element index), never stored.
 * daln_fwd: writes z, and for a later backward s = bf16(dropout(g+bias) + residual), mean[T], rstd[T] (s/mean/rstd
 *           may be NULL).  H % 4 == 0, H <= 2048, 0 <= p < 1.
 * daln_bwd: dz = dz_a + dz_b (dz_b may be NULL): the two gradients reaching z (through the next GEMM and through
 *           the residual connection).  Writes ds = d/d residual, dy = d/d g (NULL when p == 0: identical to ds),
 *           dgamma_dbeta[2H] fp32, dbias[H] (dbias_bf16 selects its dtype; NULL = skip).  ws: cfl_daln_ws_bytes.
 * bias_gelu_bwd: du = dh * gelu'(g + bias), dbias[I] = column sums of du.  I % 8 == 0.  ws: cfl_bias_gelu_ws_bytes.
 * dropout_mask: test helper, the keep flags of n elements (n % 4 == 0). */
size_t cfl_daln_ws_bytes(int T, int H);
int cfl_daln_fwd(const void* g, const void* bias, int bias_bf16, const void* residual, const float* gamma, const float* beta,
                 int T, int H, float eps, float p, unsigned seed, void* z, void* s, float* mean, float* rstd, void* stream);
int cfl_daln_bwd(const void* s, const void* dz_a, const void* dz_b, const float* gamma, const float* mean, const float* rstd,
                 int T, int H, float p, unsigned seed, void* ds, void* dy, float* dgamma_dbeta, void* dbias, int dbias_bf16,
                 void* ws, void* stream);
/* Pre-LN blocks (the build-defined ViT trunk of BASELINE.json configs[4]; the reference ships no ViT -- same element-wise glue as the
 * BertLayer lines above): s = g + bias + residual is the next residual, z = LayerNorm(s) the next GEMM's input (cfl_daln_fwd, p = 0,
 * writes both); preln_bwd: ds = LayerNorm'(dz) + ds_direct (ds_direct NULL = none) = d/d g = d/d residual, dgamma_dbeta[2H], dbias. */
int cfl_preln_bwd(const void* s, const void* dz, const void* ds_direct, const float* gamma, const float* mean, const float* rstd, int T,
                  int H, void* ds, float* dgamma_dbeta, void* dbias, int dbias_bf16, void* ws, void* stream);
size_t cfl_bias_gelu_ws_bytes(long long T, int I);
int cfl_bias_gelu_fwd(const void* g, const void* bias, int bias_bf16, long long T, int I, void* h, void* stream);
int cfl_bias_gelu_bwd(const void* g, const void* bias, int bias_bf16, const void* dh, long long T, int I, void* du, void* dbias,
                      int dbias_bf16, void* ws, void* stream);
int cfl_dropout_mask(unsigned seed, float p, long long n, unsigned char* keep, void* stream);
/* Steps replayed from a HIP graph (the server loop, retrieval_trainer.py:192-214, whose BERT tower draws dropout masks): the `seed`
 * arguments above are constants of a captured launch.  With a tick word registered here (device memory the caller increments on the
 * stream once per step; NULL = off, the default) every launch of cfl_daln_fwd / cfl_daln_bwd / cfl_dropout_mask uses
 * seed + *tick * 0x85EBCA6B.  Process-wide; the pointer is read when a launch is issued (or captured). */
int cfl_set_dropout_tick(const unsigned* tick_dev);

/* ---- BERT self-attention for short sequences (L <= 32 tokens, head_dim 64) -------------------------------------
 * softmax(Q K^T / sqrt(64) + key-padding mask) V per (batch, head): BertSelfAttention of the BertModel built at
 * src/networks/models/pcme.py:36-38 (captions are ~12-30 tokens).  One wavefront per (batch, head), MFMA, nothing
 * saved for the backward but the inputs.  bf16; element (b, t, h*64 + d) of q/k/v at b*bs + t*ld + h*64 + d (so the
 * three may be slices of one fused [B, L, 3H] projection), of o/dout at b*bso + t*ldo + ..., of dq/dk/dv at
 * b*bsg + t*ldg + ...; mask [B][L] bytes (1 = key takes part) or NULL.  All strides % 8 == 0, 16-byte aligned bases. */
int cfl_attn_small_fwd(const void* q, const void* k, const void* v, long long ld, long long bs, const unsigned char* mask,
                       int B, int L, int heads, int head_dim, void* o, long long ldo, long long bso, void* stream);
int cfl_attn_small_bwd(const void* q, const void* k, const void* v, long long ld, long long bs, const unsigned char* mask,
                       int B, int L, int heads, int head_dim, const void* dout, long long ldo, long long bso, void* dq, void* dk,
                       void* dv, long long ldg, long long bsg, void* stream);
/* The same on PACKED tokens (round 6): the reference pads every caption of a batch to the longest one and masks the padding
 * (pcme.py:43-57 builds the attention mask from the lengths); here the text tower may run on the T = sum(lengths) real tokens only,
 * [T][...] tensors, sequence b = rows cu_seqlens[b] .. cu_seqlens[b + 1] - 1 (B + 1 device ints, cu_seqlens[0] = 0, <= 32 tokens
 * each).  Per sequence the arithmetic is the padded kernel's with its key-padding mask. */
int cfl_attn_small_fwd_varlen(const void* q, const void* k, const void* v, long long ld, const int* cu_seqlens, int B, int heads,
                              int head_dim, void* o, long long ldo, void* stream);
int cfl_attn_small_bwd_varlen(const void* q, const void* k, const void* v, long long ld, const int* cu_seqlens, int B, int heads,
                              int head_dim, const void* dout, long long ldo, void* dq, void* dk, void* dv, long long ldg,
                              void* stream);

/* ---- 3 x 3 / stride 1 / padding 1 convolution of fp32 channels_last tensors on the bf16 matrix pipe at fp32-class accuracy --------
 * (round 6, csrc/conv3x3_x3.hip).  The BasicBlock convolutions of the clients' ResNet-18 (src/networks/resnet_client.py:33-66,
 * 102-201), fp32 in the reference (cuDNN): y[n,h,w,co] = sum x[n,h+kh-1,w+kw-1,ci] w[co,kh,kw,ci]; x [N,H,W,Ci], w [Co,3,3,Ci],
 * y [N,H,W,Co] dense fp32 (= torch channels_last memory).  Every operand element is split into two bf16 while it is staged and each
 * product runs as three bf16 MFMAs (tile_x3.h): ~1e-6 relative per product.  Ci % 32 == 0, Co % 64 == 0.  variant 0 = the tile
 * choice of the library (22 / 42: 128 / 256 positions x 128 channels, 21 / 41: x 64 channels).
 * The data gradient is the same call on dY and the rotated, transposed weight: cfl_conv3x3_x3_rot_weight writes
 * w_rot[ci][kh][kw][co] = w[co][2-kh][2-kw][ci]. */
int cfl_conv3x3_x3_supported(int N, int H, int W, int Ci, int Co);
int cfl_conv3x3_x3_fwd(const float* x, const float* w, int N, int H, int W, int Ci, int Co, float* y, int variant, void* stream);
int cfl_conv3x3_x3_rot_weight(const float* w, int Ci, int Co, float* w_rot, void* stream);
/* Version 3 of the forward kernel (W <= 63) takes the weight as an IMAGE of its own LDS stage, written once per call by
 * cfl_conv3x3_x3_wimage: img[((tap * Ci/32 + c) * Co + co) * 128 B] = [32 hi | 32 lo] bf16 of w[co][tap][32 c .. 32 c + 31], 16-byte
 * pieces XOR-swizzled by (co >> 1) & 7; cfl_conv3x3_x3_wimage_bytes(Ci, Co) bytes (= the fp32 weight's size).  variant 0 = chosen by
 * the library, 2MN = a tile of 64 M positions x 64 N channels (222, 242, 221, 241, 212, 211).  Data gradient: image of the rotated weight. */
size_t cfl_conv3x3_x3_wimage_bytes(int Ci, int Co);
int cfl_conv3x3_x3_wimage(const float* w, int Ci, int Co, void* img, void* stream);
int cfl_conv3x3_x3_wimage_rot(const float* w, int Ci, int Co, void* img, void* stream);   /* image of w_rot (Co x 9 Ci -> Ci x 9 Co) in one pass */
int cfl_conv3x3_x3_fwd_img(const float* x, const void* wimg, int N, int H, int W, int Ci, int Co, float* y, int variant, void* stream);
/* Round 6, stride 2 / padding 1 (the three down-sampling 3x3 convolutions of the clients' ResNet-18, resnet_client.py:33-66 with stride 2):
 * forward only, even H and W <= 62, y [N, H/2, W/2, Co]; the same weight image; data / weight gradients stay on the library. */
int cfl_conv3x3_x3_fwd_img_s2(const float* x, const void* wimg, int N, int H, int W, int Ci, int Co, float* y, void* stream);
/* The weight gradient of the same convolutions (csrc/wgrad3x3_x3.hip; the reference: autograd through cuDNN's fp32 backward-filter,
 * src/algorithms/ClientTrainer.py:420 loss.backward()): dw[co,kh,kw,ci] = sum over (n,h,w) dy[n,h,w,co] x[n,h+kh-1,w+kw-1,ci], fp32 in
 * and out, products as three bf16 MFMAs on operands split while they are staged.  H = W in {7, 14, 28, 56}, Ci % 64 == 0,
 * Co % 64 == 0; split-K over row ranges into fp32 partials in ws (cfl_conv3x3_x3_wgrad_ws_bytes) + a fixed-order reduce:
 * deterministic.  cfl_conv3x3_x3_wgrad_splits(n > 0) overrides the number of row ranges (0 = default; returns the old value). */
int cfl_conv3x3_x3_wgrad_supported(int N, int H, int W, int Ci, int Co);
size_t cfl_conv3x3_x3_wgrad_ws_bytes(int N, int H, int W, int Ci, int Co);
int cfl_conv3x3_x3_wgrad(const float* dy, const float* x, int N, int H, int W, int Ci, int Co, float* dw, void* ws, void* stream);
int cfl_conv3x3_x3_wgrad_splits(int splits);

/* ---- ResNet stem max pooling, 3x3 / stride 2 / pad 1, NHWC bf16 ---------------------------------------------------
 * torchvision ResNet.maxpool of the trunk built at src/networks/models/image_encoder.py:27-36 (and the client trunk,
 * src/networks/resnet_client.py:19).  x [N,H,W,C] -> y [N,Ho,Wo,C], Ho = (H-1)/2+1; idx: one byte per output element
 * (the winning tap 0..8, first maximum in row-major window order, NaN wins -- torch's rule).  C % 8 == 0.
 * bwd: dx [N,H,W,C] from dy and idx, gather form (no atomics). */
/* Stem space-to-depth: the 7x7 / stride 2 / pad 3 convolution on 3 channels (torchvision ResNet.conv1 inside
 * image_encoder.py:27-36) is exactly a 4x4 / stride-1 convolution of out[n, i, j, (2p+q)*3 + c] = x[n, 2(i-2)+p, 2(j-2)+q, c]
 * (zero for i, j outside [2, H/2+2) and in channels 12..15).  x [N,H,W,3] bf16 or (x_f32 != 0) fp32, rounded to bf16 here
 * (H, W even), out [N, H/2+3, W/2+3, 16] bf16. */
int cfl_stem_s2d(const void* x, int x_f32, int N, int H, int W, void* out, void* stream);
int cfl_maxpool3s2_fwd(const void* x, int N, int H, int W, int C, void* y, void* idx, void* stream);
int cfl_maxpool3s2_bwd(const void* dy, const void* idx, int N, int H, int W, int C, void* dx, void* stream);

/* ---- bf16 MFMA GEMMs for the 1x1 convolutions of the ResNet trunk -------------------------------------------------
 * nt: C[M,N] = A[M,K] * B[N,K]^T, bf16 in / bf16 out, fp32 accumulation.  On the NHWC-flattened activation a 1x1
 * convolution of the torchvision Bottleneck blocks (src/networks/models/image_encoder.py:27-36) is this GEMM:
 * forward A = x[M,Ci], B = w[Co,Ci]; data gradient A = dy[M,Co], B = w^T[Ci,Co].  It ties MIOpen's forward and beats
 * its backward-data kernels on every ResNet-101 shape, so the product uses it for the DATA GRADIENT (ops.conv1x1).
 * lda/ldb/ldc in elements; K % 64 == 0, N % 8 == 0, lda % 8 == ldb % 8 == ldc % 8 == 0, 16-byte aligned pointers.
 * variant 0 = pick by shape; 21 / 22 / 41 / 42 / 44 select the wave tile (TM,TN) for benchmarking; 90 = the B-resident
 * streaming kernel (K in {64, 128, 256, 512, 1024}, N % 64 == 0, N / tile width divides 32) or CFL_ELIMIT. */
int cfl_gemm_bf16_nt(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc,
                     int M, int N, int K, int variant, void* stream);
/* The same GEMM with the gradient JOIN of a residual block in its epilogue (round 3): C = (A B^T + add) . mask, C / add dense
 * [M, N] bf16, mask = 1 bit per element as cfl_bn_fwd writes it (byte i = elements 8 i .. 8 i + 7).  Used for the data gradient
 * of a bottleneck's first 1x1 convolution (torchvision resnet.py Bottleneck.forward: out += identity; relu): the BatchNorm
 * backward of the layer below then reads one pre-masked gradient (cfl_bn_bwd with relu = 0, has_residual = 0). */
int cfl_gemm_bf16_nt_join(const void* A, long long lda, const void* B, long long ldb, void* C, const void* add,
                          const unsigned char* mask, int M, int N, int K, void* stream);
/* Forward of a 1x1 convolution that a training-mode BatchNorm follows (torchvision Bottleneck conv3 -> bn3,
 * src/networks/models/image_encoder.py:27-36): C = A B^T on the B-resident streaming kernel with the BatchNorm's batch statistics in
 * the epilogue -- pstat[0][nblk][N] / pstat[1][nblk][N] = per-column sum / sum of squares of the STORED bf16 outputs, one partial
 * row per wave.  cfl_gemm_bf16_nt_stats_nblk = nblk for a shape (K in {64, 128, 256}, N % 128 == 0, N / 128 divides 32), 0 when
 * the shape is not taken.  C dense [M, N]; pstat 2 * nblk * N floats, 16-byte aligned.  cfl_bn_fwd_pre is cfl_bn_fwd without its
 * statistics pass. */
int cfl_gemm_bf16_nt_stats_nblk(int M, int N, int K);
int cfl_gemm_bf16_nt_stats(const void* A, long long lda, const void* B, long long ldb, void* C, int M, int N, int K, float* pstat,
                           void* stream);
int cfl_bn_fwd_pre(const void* x, const void* residual, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, long long R, int C, float eps, float momentum, int relu, void* y, float* save_mean,
                   float* save_invstd, unsigned char* relu_mask, const float* psum, const float* psq, int nblk, void* stream);
/* Measurement / test knob: the join entry runs the B-resident streaming kernel (weight tile resident in LDS, A rows streamed
 * through registers) when M >= min_m and K <= 256, the tile kernel otherwise.  Default 32768 (CFL_GEMM_BRES_MIN_M overrides it;
 * CFL_GEMM_NO_BRES=1 disables the streaming kernel).  Returns the previous value; min_m < 0 only queries. */
int cfl_gemm_bf16_bres_min_m(int min_m);
/* dst[C][R] = src[R][C]^T, dense bf16 (the weight transpose the data gradient needs). */
int cfl_transpose_bf16(const void* src, int R, int C, void* dst, void* stream);
/* every weight transpose of a backward pass in one launch: meta = device array of ntensors 40-byte records
 * {const void* src; void* dst; int R; int C; int tile0; int tiles_c; int lds; int ldd} (tile0 ascending; 64x64 tiles):
 * dst[c * ldd + r] = src[r * lds + c].  Dense matrices: lds = C, ldd = R; the taps of a k x k channels_last weight are
 * strided [Co, Ci] matrices (ops.prepare_weight_transposes builds the rotated, transposed weight of the data gradient). */
int cfl_transpose_bf16_multi(const void* meta, int ntensors, int total_tiles, void* stream);

/* ---- client supervised step glue ------------------------------------------------
 * Replaces, per local batch (src/algorithms/ClientTrainer.py:344-361 with to_one_hot src/utils/Utils.py:6-13 and
 * accuracy ClientTrainer.py:114-129):
 *     fvec  = fvec - inter_distance * to_one_hot(labels, C)
 *     loss  = CE(fvec, labels);  center = CE(class_weight @ class_weight^T, arange(C))
 *     total = loss + center_weight * center          (center_weight = 0.5 in the reference)
 *     prec1, preck = accuracy(fvec, labels, topk=(1, topk))
 * fvec [B,C], labels int64 [B] (a label outside [0,C) makes the loss NaN), class_weight [C,Dw], all dense row-major.
 * out5 = {total, loss, center, prec@1 in %, prec@topk in %}.  Ties in the top-k count the lower class index first.
 * bwd: gout[0] = d/d total (device scalar); dfvec [B,C] and dclass_weight [C,Dw] may each be NULL.
 * ws: cfl_sup_ws_bytes(B, C), written by fwd and read by bwd.  C <= 4096. */
size_t cfl_sup_ws_bytes(int B, int C);
int cfl_sup_glue_fwd(const float* fvec, const long long* labels, const float* class_weight, int B, int C, int Dw,
                     float inter_distance, int topk, float center_weight, float* out5, void* ws, void* stream);
int cfl_sup_glue_bwd(const float* fvec, const long long* labels, const float* class_weight, int B, int C, int Dw,
                     float inter_distance, float center_weight, const float* gout, const void* ws, float* dfvec,
                     float* dclass_weight, void* stream);

/* ---- A2c: the text towers' GRU recurrence, last valid step only ------------------------------------------
 * Replaces nn.GRU(bidirectional) over packed captions + gather at lengths - 1
 * (src/networks/language_model.py:93-107 EncoderText.forward; src/networks/models/caption_encoder.py:87-101):
 * the reference keeps, per caption, the forward direction's final state and the backward direction's first step.
 *   cfl_gru_fwd:  xp [B, T, 3H] = words W_ih^T + b_ih (gate order r | z | n, torch.nn.GRU), w_hh [3H, H], b_hh [3H], lens int32 [B] ON
 *                 THE DEVICE (clamped to [0, T]); out [B, H] = state after lens[b] steps from a zero state.  hs [T + 1, B, H]
 *                 (time-major; hs[t] = state before step t) and gates [B, T, 4H] (r | z | n | hidden-side n pre-activation) are
 *                 written when non-NULL (training) -- positions beyond a row's length: hs carries the final state, gates are
 *                 not written (never read).
 *   cfl_gru_bwd:  dout [B, H] -> dxp [B, T, 3H] (gradient of xp, batch-major) and dg [T, B, 3H] (gradient of the hidden-side
 *                 pre-activations h W_hh^T + b_hh, time-major); zeros beyond a row's length.  The caller forms
 *                 dW_ih = dxp^T words, db_ih = sum dxp, dwords = dxp W_ih, dW_hh = dg^T hs[:T], db_hh = sum dg with library GEMMs.
 *   cfl_gru_cell0_fwd / _bwd: one GRU cell from a zero state (the backward direction at the last valid word): gx [B, 3H] =
 *                 x_last W_ih^T + b_ih -> out [B, H], saved [B, 3H] (r | z | n, may be NULL); bwd: dgx [B, 3H], dgh [B, 3H]
 *                 (per-row gradient of b_hh; the gradient of W_hh is identically zero: the state it multiplies is zero).
 * fp32 throughout, FMA chains in k order.  H in {32, 64, 128}: a row of W_hh per thread in registers for the whole launch; any
 * other H % 4 == 0 up to 512 (cfl_gru_streams_weights): W_hh streamed from L2 every step, and cfl_gru_fwd then needs
 * w_hh_t = W_hh^T [H, 3H] (k-major, coalesced; NULL otherwise).  Other widths stay on the library's GRU (cfl_gru_supported). */
int cfl_gru_supported(int H);
int cfl_gru_streams_weights(int H);
int cfl_gru_fwd(const float* xp, const float* w_hh, const float* w_hh_t, const float* b_hh, const int* lens, float* out, float* hs,
                float* gates, int B, int T, int H, void* stream);
int cfl_gru_bwd(const float* dout, const float* w_hh, const int* lens, const float* hs, const float* gates, float* dxp, float* dg,
                int B, int T, int H, void* stream);
int cfl_gru_cell0_fwd(const float* gx, const float* b_hh, float* out, float* saved, int B, int H, void* stream);
int cfl_gru_cell0_bwd(const float* dout, const float* saved, const float* b_hh, float* dgx, float* dgh, int B, int H, void* stream);

/* ---- A5: con_w aggregation ----------------------------------------------------------------
 * Replaces the closure `aggregation` in MMFL.distill (src/algorithms/MMFL.py:298-335).
 * logprob: out_l[n - row0] = V_n.G_n - log sum_m exp(V_n.G_m)  for n in [row0, row0+rows)
 *          (rows are independent => row-shardable across GPUs).
 * combine: W = softmax over the C clients of L[C, M]; out[n,:] = sum_c W[c,n] V_c[n,:].
 *          Vptrs_host: HOST array of C device pointers, C <= 64.  W_out [C, M] may be NULL.
 */
size_t cfl_conw_ws_bytes(int rows, int M, int D);
int cfl_conw_logprob(const float* V, const float* G, int M, int D, int row0, int rows,
                     float* out_l, void* ws, void* stream);
int cfl_conw_combine(const float* const* Vptrs_host, const float* L, int C, int M, int D,
                     float* out, float* W_out, void* stream);
/* Round 4: the same log-probabilities on the bank pass of rows A3/A4 (cfl_bank_image_build of G): the 8 waves of a workgroup
 * hold 256 rows of V in registers and stream the bank image once per 256 rows (32 x 32 x 16 MFMAs), online log-sum-exp per row, then one small launch
 * merges the splits and takes the positives as exact fp32 dot products.  Round 6: 256 < D <= 512 on the 4-wave form (one wave per
 * SIMD, 128 rows of V per workgroup), 512 < D <= 768 on 16-row steps (16 x 16 x 32 MFMAs, per-lane running log-sum-exp).
 * rows >= 512, D <= 768, D % 4 == 0, 16-byte aligned
 * operands; otherwise CFL_ELIMIT (cfl_conw_logprob takes every shape).  Same reference lines (MMFL.py:304-307).
 */
int cfl_conw_img_supported(int rows, int M, int D);
size_t cfl_conw_img_ws_bytes(int rows, int M, int D);
int cfl_conw_logprob_img(const float* V, const void* image_of_G, const float* G, int M, int D, int row0, int rows,
                         float* out_l, void* ws, void* stream);

/* ---- A2-head: PIE attention pooling + epilogue --------------------------------------------
 * Replaces MultiHeadSelfAttention.forward (n_head = 1) and the tail of PIENet.forward
 * (src/networks/models/pie_model.py:28-40,61-67), the avgpool of EncoderImage.forward
 * (image_encoder.py:55) and l2_normalize (src/utils/tensor_utils.py:25-27).
 * The two dense projections (w_1: Cd -> dh and fc: Cd -> D) stay library GEMMs on the host side.
 *
 * pool fwd: X [N,P,Cd], H = w_1(X) [N,P,dh] (pre-tanh), w2 [dh], mask [N,P] uint8 (1 = padded,
 *           may be NULL).  attn = softmax_P(w2 . tanh(H)), pooled = attn^T X [N,Cd],
 *           xmean = mean_P X [N,Cd] (may be NULL).
 * pool bwd: given d_pooled [N,Cd] and d_xmean [N,Cd] (may be NULL):
 *           dX = attn (x) d_pooled + d_xmean/P ; dH = ds (x) w2 * (1 - tanh^2 H) ; dw2 = sum ds tanh(H)
 *           with ds = softmax-backward of <d_pooled, X_p>.
 * epilogue fwd: r = sigmoid(res_pre); o = LayerNorm(out + r) * ln_w + ln_b; y = o / max(|o|, 1e-12)
 *           (flags & CFL_EPI_NO_L2NORM: y = o).  Saves stats[N,4] = {mean, rstd, 1/norm, 0}.
 * epilogue bwd: given dy (and optional direct gradients do_, dres on the o / r outputs), the
 *           forward inputs `out`, the saved r and stats -> d_out [N,D], d_res_pre [N,D], d_ln_w [D],
 *           d_ln_b [D].  ws: cfl_pie_ws_bytes(N, 1, D, 1).
 * l2norm: y = x / max(|x|_2, 1e-12) row-wise; inv_norm [N] saved for bwd.
 * Limits: P <= 1024, D <= 4096.
 */
#define CFL_EPI_NO_L2NORM 1
size_t cfl_pie_ws_bytes(int N, int P, int Cd, int dh);
int cfl_pie_pool_fwd(const float* X, const float* H, const float* w2, const unsigned char* mask,
                     int N, int P, int Cd, int dh, float* attn, float* pooled, float* xmean,
                     void* ws, void* stream);
int cfl_pie_pool_bwd(const float* X, const float* H, const float* w2, const unsigned char* mask,
                     const float* attn, const float* d_pooled, const float* d_xmean,
                     int N, int P, int Cd, int dh, float* dX, float* dH, float* dw2,
                     void* ws, void* stream);
/* Single-pass form of the two pool entry points (csrc/pie_fused.hip): X and H are read once per direction in the element
 * type the trunk / the w_1 GEMM produced (bf16 != 0: bfloat16 -- the autocast regime -- else fp32; fp32 arithmetic either
 * way) and dX / dH are written in that type.  Same math as cfl_pie_pool_fwd/bwd (pie_model.py:28-40,
 * image_encoder.py:54-57); tanh through exp2/rcp (absolute error <= 3e-7).
 *   cfl_pie_fused_supported: 1 when P <= 1024, Cd and dh multiples of the 16-byte vector (4 fp32 / 8 bf16), Cd <= 8192, dh <= 4096;
 *   otherwise use cfl_pie_pool_fwd/bwd on fp32 copies (the GRU text head, dh = 150).
 *   head_bwd ws: cfl_pie_ws_bytes(N, P, Cd, dh).  All tensor pointers 16-byte aligned. */
int cfl_pie_fused_supported(int N, int P, int Cd, int dh, int bf16);
int cfl_pie_head_fwd(const void* X, const void* H, int bf16, const float* w2, const unsigned char* mask,
                     int N, int P, int Cd, int dh, float* attn, float* pooled, float* xmean, void* stream);
int cfl_pie_head_bwd(const void* X, const void* H, int bf16, const float* w2, const float* attn,
                     const float* d_pooled, const float* d_xmean, int N, int P, int Cd, int dh,
                     void* dX, void* dH, float* dw2, void* ws, void* stream);
int cfl_pie_epilogue_fwd(const float* out, const float* res_pre, const float* ln_w, const float* ln_b,
                         int N, int D, float ln_eps, int flags, float* y, float* o, float* r,
                         float* stats, void* stream);
int cfl_pie_epilogue_bwd(const float* dy, const float* do_, const float* dres, const float* out,
                         const float* r, const float* ln_w, const float* ln_b, const float* stats,
                         int N, int D, int flags, float* d_out, float* d_res_pre, float* d_ln_w,
                         float* d_ln_b, void* ws, void* stream);
int cfl_l2norm_fwd(const float* x, int N, int D, float* y, float* inv_norm, void* stream);
int cfl_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, int N, int D,
                   float* dx, void* stream);

/* ---- A6: retrieval rank of the best positive ----------------------------------------------
 * Replaces ParallelMatMulModule.forward + the positive-rank loop of evaluate_recall
 * (src/algorithms/eval_coco.py:37-51,296-317): ranks[q] = #{g : <Q_q,G_g> > max_{lab_g == lab_q}
 * <Q_q,G_g>} with products and sums in fp64 on the fp32-valued inputs, like the reference's
 * float64 numpy buffers.  ranks [Nq] int32.  Queries with no positive get rank Ng.
 */
size_t cfl_rank_ws_bytes(int Nq, int Ng, int D);
int cfl_rank_count(const float* Q, const float* G, const long long* qlab, const long long* glab,
                   int Nq, int Ng, int D, int* ranks, void* ws, void* stream);

/* ---- A2 trunk glue: fused BatchNorm (+ residual add) (+ ReLU), training mode, NHWC bf16 ------
 * Replaces nn.BatchNorm2d -> (+= identity) -> nn.ReLU between the convolutions of the ResNet trunks
 * (torchvision Bottleneck/BasicBlock as used by src/networks/models/image_encoder.py:27; src/networks/
 * resnet_client.py:60-98).  x, residual, y, dy, dx, dres: bf16 [R, C] row-major (= channels_last [N,C,H,W] with
 * R = N*H*W); gamma/beta/statistics fp32 [C].  C % 8 == 0 and (C/8 divides 256 or C >= 2048 in multiples of 2048).
 * fwd:   batch statistics (biased variance), running stats updated in place when non-NULL
 *        (unbiased variance, momentum as in torch), y = relu?((x-mean)*invstd*gamma + beta (+ residual)).
 * apply: the same normalisation with given mean / invstd (evaluation mode).
 * bwd:   dy' = dy * (y > 0) if relu;  dgamma = sum dy' xhat, dbeta = sum dy';
 *        dx = gamma*invstd*(dy' - dbeta/R - xhat*dgamma/R);  dres = dy' (when has_residual).
 *        dy2 (may be NULL) is a second upstream gradient, added on the fly (the block output feeds the next
 *        convolution and the next residual add).  y may be NULL for relu without residual: the mask is then
 *        recomputed from x (needs beta), one activation read less in each of the two passes.
 *        relu_mask (fwd: optional output, R*C/8 bytes; bwd: optional input replacing y): bit k of byte i = (y > 0) of
 *        element 8i+k -- with a residual the backward then reads 1 byte instead of 16 per 8 outputs, twice.
 */
/* Channel-sliced block map of the BatchNorm kernels (round 5): for C = 256 ... 2048 (whole 64-channel slices) the statistics /
 * reduce pass and the apply pass of cfl_bn_fwd / cfl_bn_bwd run on workgroups that own a 64-channel slice of a row range, the apply
 * pass sums the partial rows of its own slice in its prologue and the `final` launch between the two passes is gone (torchvision
 * BatchNorm2d inside the trunks of src/networks/models/image_encoder.py:27-36: same results up to fp32 summation order).
 * cfl_bn_sliced(0 / 1) switches the map off / on, a negative argument only queries; returns the previous setting
 * (CFL_BN_NO_SLICE=1 in the environment starts with it off).  Measurement switch of tools/ab_step.py --knob bnslice. */
int cfl_bn_sliced(int on);
/* cfl_bn_bwd for a PRE-JOINED gradient (relu = 0, has_residual = 0, one upstream gradient: the bn3 backward of a bottleneck whose
 * gradient join ran in the data-gradient GEMM above, cfl_gemm_bf16_nt_join) whose apply pass ALSO produces the weight gradient of
 * the 1 x 1 convolution that made x (torchvision Bottleneck.conv3 inside src/networks/models/image_encoder.py:27-36; the reference
 * leaves it to cuDNN, this build's default is a library kernel on a side stream that re-reads dx from memory):
 *   dx[R, C] bf16 as cfl_bn_bwd,  dw[C, P] bf16 = dx^T a_in,  a_in [R, P] bf16 = the convolution's input (channels_last rows).
 * The dx tile of a stage goes to memory and, transposed through LDS, into the MFMAs; split-K partials + a fixed-order reduce
 * (deterministic).  P = 256, C a multiple of 128 (layer3 of ResNet-50 / -101); `supported` says whether a shape is taken.
 * ws: cfl_bn_bwd_wgrad_ws_bytes (the reduce pass's partials + parts x C x P floats). */
int cfl_bn_bwd_wgrad_supported(long long R, int C, int P);
size_t cfl_bn_bwd_wgrad_ws_bytes(long long R, int C, int P);
int cfl_bn_bwd_wgrad(const void* dy, const void* x, const void* a_in, int P, const float* gamma, const float* save_mean,
                     const float* save_invstd, long long R, int C, void* dx, float* dgamma, float* dbeta, void* dw, void* ws,
                     void* stream);
/* Weight gradient of a 3 x 3 / stride 1 / padding 1 convolution on channels_last bf16 activations (torchvision Bottleneck.conv2 inside
 * src/networks/models/image_encoder.py:27-36; the reference leaves it to cuDNN, this build's fallback is MIOpen's igemm_wrw kernel):
 *   dw[co][kh][kw][ci] (bf16, the channels_last weight's own memory order) = sum_{n,h,w} dy[n,h,w,co] x[n,h+kh-1,w+kw-1,ci]
 * dy [N,H,W,Co], x [N,H,W,Ci] bf16 rows.  One MFMA K step per zero-padded image row, both operands read transposed from LDS, the nine
 * taps share operands through a rolling window of rows and modular w-shifts; split-K over image ranges (one range per XCD at a time)
 * with fp32 partials and a fixed-order reduce: deterministic.  `supported`: square maps of 7, 14 or 28 with Ci % 64 == 0 and
 * Co % 128 == 0, or 56 x 56 with Co == 64 (every stride-1 conv2 of a ResNet-50 / -101).  ws: cfl_conv3x3_wgrad_ws_bytes
 * (splits x Co x 9 x Ci floats).
 * cfl_conv3x3_wgrad_splits(n): n > 0 forces the number of image ranges (rounded down to a multiple of 8; measurements: 16 runs the
 * layer3 shape on half of the chip), 0 restores the default (256 workgroups), negative only queries; returns the previous value. */
int cfl_conv3x3_wgrad_supported(int N, int H, int W, int Ci, int Co);
size_t cfl_conv3x3_wgrad_ws_bytes(int N, int H, int W, int Ci, int Co);
int cfl_conv3x3_wgrad(const void* dy, const void* x, int N, int H, int W, int Ci, int Co, void* dw, void* ws, void* stream);
int cfl_conv3x3_wgrad_splits(int splits);
/* measurement only (tools/kernel_bench.py --cases w3dbg): 1 = the layer3-shaped kernel without its staging, 2 = staging and barriers
 * only -- the RESULTS ARE WRONG in these modes; 0 restores the kernel, negative only queries; returns the previous mode */
int cfl_conv3x3_wgrad_debug(int mode);
/* Weight gradient of a 1 x 1 / stride 1 convolution on channels_last bf16 activations (torchvision Bottleneck.conv1 / conv3 /
 * downsample[0] inside src/networks/models/image_encoder.py:27-36; the reference leaves it to cuDNN, this build's fallback is the
 * library's batched GEMM with fp32 atomics):   dw[co][ci] (bf16) = sum_m dy[m, co] x[m, ci],   dy [M, Co], x [M, Ci] bf16 rows.
 * Row-major LDS-DMA staging, transposing LDS reads, up to 256 x 256 of dw per workgroup, ~128 workgroups, split-K over row ranges with
 * fp32 partials and a fixed-order reduce: deterministic.  `supported`: (Co, Ci) multiples of (256, 256), (128, 256), (256, 128), or
 * Co == 64 with Ci % 256 == 0, Ci == 64 with Co % 256 == 0, 64 x 64 (every stride-1 1 x 1 convolution of a ResNet-50 / -101).
 * ws: cfl_conv1x1_wgrad_ws_bytes (splits x Co x Ci floats).  cfl_conv1x1_wgrad_workgroups(n): n > 0 sets the number of workgroups
 * the split count aims at (default 128: the kernel is bound by bytes and lives on a side stream); returns the previous value. */
int cfl_conv1x1_wgrad_supported(long long M, int Ci, int Co);
size_t cfl_conv1x1_wgrad_ws_bytes(long long M, int Ci, int Co);
int cfl_conv1x1_wgrad(const void* dy, const void* x, long long M, int Ci, int Co, void* dw, void* ws, void* stream);
int cfl_conv1x1_wgrad_workgroups(int wgs);
/* cfl_bn_fwd / cfl_bn_apply / cfl_bn_bwd for FP32 activations (channels_last rows of C floats; every other argument as in the bf16
 * entries below, same kernels instantiated on 32-byte channel groups): BatchNorm2d (+ residual add) (+ ReLU) of the clients' fp32
 * encoders (src/networks/resnet_client.py:33-66,162-201 BasicBlock / stem; the reference runs them in fp32, ClientTrainer.py has no
 * mixed precision).  fp32 statistics and arithmetic: results equal torch's fp32 BatchNorm to summation order. */
int cfl_bn_fwd_f32(const void* x, const void* residual, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, long long R, int C, float eps, float momentum, int relu, void* y, float* save_mean,
                   float* save_invstd, unsigned char* relu_mask, void* ws, void* stream);
int cfl_bn_apply_f32(const void* x, const void* residual, const float* mean, const float* invstd, const float* gamma,
                     const float* beta, long long R, int C, int relu, void* y, void* stream);
int cfl_bn_bwd_f32(const void* dy, const void* dy2, const void* x, const void* y, const unsigned char* relu_mask, const float* gamma,
                   const float* beta, const float* save_mean, const float* save_invstd, long long R, int C, int relu,
                   int has_residual, void* dx, void* dres, float* dgamma, float* dbeta, void* ws, void* stream);
size_t cfl_bn_ws_bytes(long long R, int C);
int cfl_bn_fwd(const void* x, const void* residual, const float* gamma, const float* beta, float* running_mean,
               float* running_var, long long R, int C, float eps, float momentum, int relu, void* y,
               float* save_mean, float* save_invstd, unsigned char* relu_mask, void* ws, void* stream);
int cfl_bn_apply(const void* x, const void* residual, const float* mean, const float* invstd, const float* gamma,
                 const float* beta, long long R, int C, int relu, void* y, void* stream);
int cfl_bn_bwd(const void* dy, const void* dy2, const void* x, const void* y, const unsigned char* relu_mask, const float* gamma,
               const float* beta, const float* save_mean, const float* save_invstd, long long R, int C, int relu, int has_residual,
               void* dx, void* dres, float* dgamma, float* dbeta, void* ws, void* stream);
/* Stem tail (torchvision ResNet.bn1 -> relu -> maxpool(3, 2, 1) inside image_encoder.py:27-36) in one pass per direction:
 * the normalised activation and the scattered pooling gradient (411 MB each at batch 256) are never materialised.
 *   fwd: batch statistics of x (+ running statistics), then y_pool = maxpool(bf16(relu(bn(x)))) with the arg-max taps in idx --
 *        bit-identical to cfl_bn_fwd followed by cfl_maxpool3s2_fwd.   x [N,H,W,C] bf16, y_pool [N,Ho,Wo,C] bf16,
 *        idx [N*Ho*Wo*C] bytes, Ho = (H-1)/2+1;  ws: cfl_bn_ws_bytes(N*H*W, C).
 *   bwd: given g_pool (gradient w.r.t. y_pool): dx, dgamma, dbeta -- bit-identical to cfl_maxpool3s2_bwd followed by cfl_bn_bwd. */
int cfl_bn_pool_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, int N, int H,
                    int W, int C, float eps, float momentum, void* y_pool, void* idx, float* save_mean, float* save_invstd,
                    void* ws, void* stream);
int cfl_bn_pool_bwd(const void* g_pool, const void* idx, const void* x, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_invstd, int N, int H, int W, int C, void* dx, float* dgamma,
                    float* dbeta, void* ws, void* stream);
/* The same two passes for fp32 activations (the clients' encoders, src/networks/resnet_client.py:25-29,64: bn1 -> relu -> maxpool;
 * fp32 in the reference): x, y_pool, g_pool, dx are float tensors, nothing is rounded. */
int cfl_bn_pool_fwd_f32(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, int N, int H,
                        int W, int C, float eps, float momentum, void* y_pool, void* idx, float* save_mean, float* save_invstd,
                        void* ws, void* stream);
int cfl_bn_pool_bwd_f32(const void* g_pool, const void* idx, const void* x, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, int N, int H, int W, int C, void* dx, float* dgamma,
                        float* dbeta, void* ws, void* stream);

/* ---- S1 tail: fused multi-tensor gradient clip + AdamP step ------
 * Replaces nn.utils.clip_grad_norm_(model.parameters(), 2) + AdamP.step()
 * (src/algorithms/retrieval_trainer.py:211-214; optimizer = adamp.AdamP 0.3.0, optimizers.py:24).
 * Tensors are described by a device array of CflTensorMeta (p, g, m, v share one dense layout whose
 * outermost dimension is dim 0: `n0` rows of `inner` contiguous elements); work is described by a device
 * array of int triples {tensor, start, count} -- rows for CFL_OPT_MATRIX tensors (dim > 1: projection
 * applies), elements otherwise.  matrix_ids lists the CFL_OPT_MATRIX tensors.
 * cfl_grad_clip_coef: out2 = { ||g||_2 over CFL_OPT_CLIP tensors, min(1, max_norm / (norm + 1e-6)) };
 *                     partial_ws >= n_items floats.
 * cfl_adamp_step:     rowstats_ws >= 4 * (total rows of matrix tensors) floats, tstats_ws >= n_tensors floats;
 *                     clip_dev = out2 of cfl_grad_clip_coef or NULL; `step` is the 1-based step count.
 * cfl_adamp_step_counted: the same step with the count read ON THE DEVICE: a tensor's step = *gstep_dev + its
 *                     CflTensorMeta.step (then a signed offset, clamped to >= 1).  For steps replayed from a HIP graph (the
 *                     clients' contrast loops, MMClientTrainer.py:150-224; the server loop, retrieval_trainer.py:192-214):
 *                     a captured launch cannot carry the host's count as an argument; the caller increments the counter on
 *                     the same stream before each step.
 * Mixed precision (the reference trains with apex O2: fp16 model weights, fp32 master weights,
 * retrieval_trainer.py:107-111): p is the fp32 master, p16 the bf16 model weight rewritten by pass 3, and g may be
 * bf16 (CFL_OPT_GRAD_BF16); moments are always fp32.
 */
#define CFL_OPT_MATRIX    1
#define CFL_OPT_CLIP      2
#define CFL_OPT_GRAD_BF16 4      /* g points at bf16 gradients (mixed-precision trunks: bf16 weights, fp32 masters) */
typedef struct CflTensorMeta {
    void* p; void* g; void* m; void* v;
    void* p16;              /* optional bf16 shadow of p (the weight the model computes with), or NULL */
    long long numel;
    long long inner;
    long long row_base;     /* first row of this tensor in rowstats_ws */
    int n0;
    int flags;
    int step;               /* this tensor's own 1-based step count for the bias corrections (adamp.AdamP keeps `step` per
                             * parameter: tensors whose gradient was None in some steps lag behind); 0 = use the `step` argument.
                             * cfl_adamp_step_counted: the tensor's signed offset from the device counter */
    int reserved;
} CflTensorMeta;
int cfl_grad_clip_coef(const CflTensorMeta* meta_dev, const int* items_dev, int n_items, float max_norm,
                       float* partial_ws, float* out2, void* stream);
int cfl_adamp_step(const CflTensorMeta* meta_dev, int n_tensors, const int* items_dev, int n_items,
                   const int* matrix_ids_dev, int n_matrix, float* rowstats_ws, float* tstats_ws,
                   float lr, float beta1, float beta2, float eps, float weight_decay, float delta,
                   float wd_ratio, int nesterov, int step, const float* clip_dev, void* stream);
int cfl_adamp_step_counted(const CflTensorMeta* meta_dev, int n_tensors, const int* items_dev, int n_items,
                           const int* matrix_ids_dev, int n_matrix, float* rowstats_ws, float* tstats_ws,
                           float lr, float beta1, float beta2, float eps, float weight_decay, float delta,
                           float wd_ratio, int nesterov, const int* gstep_dev, const float* clip_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CREAMFL_HIP_H */
