/*
 * dclip_hip.h — C ABI of libdclip_hip.so: the MI355X (gfx950) kernels behind the DCLIP
 * distillation step.
 *
 * The reference (ChuckDanz/DCLIP) has no FFI of its own: every dense operation on its hot
 * path is a call into a third-party Python library (HF transformers CLIPModel, torch
 * nn.MultiheadAttention, torch.nn.functional).  Each entry point below therefore cites the
 * reference call site (path:line under /root/reference, or hf: for transformers 5.15.0
 * modeling_clip.py) whose arithmetic it replaces.  The Python binding a maintainer adds is
 * shown in INTEGRATION.md; the one this repo ships is dclip_amd/_lib.py (ctypes).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the caller (incl.
 *     `workspace`); the library never allocates, frees or synchronises.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     work is enqueued on it and the call returns immediately.
 *   - tensors are row-major, contiguous unless a leading dimension is given; activations are
 *     [rows = batch*seq, features]; weights are [out, in] exactly as stored in the checkpoints.
 *   - return 0 on success, negative DCLIP_E* otherwise; dclip_last_error() gives the text of the
 *     calling thread's last failure.  No exceptions cross the ABI.
 *   - all arithmetic is IEEE fp32 (exact-f32 MFMA, v_mfma_f32_32x32x2_f32 / 16x16x4_f32).
 */
#ifndef DCLIP_HIP_H
#define DCLIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCLIP_ABI_VERSION 1

enum {
  DCLIP_OK = 0,
  DCLIP_EINVAL = -1,      /* bad shape / null pointer / unsupported size */
  DCLIP_EWORKSPACE = -2,  /* workspace too small */
  DCLIP_ELAUNCH = -3      /* HIP launch error (text in dclip_last_error) */
};

int dclip_abi_version(void);
const char* dclip_last_error(void);

/* ------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = epilogue( sum_k A(m,k) * B(k,n) )           (fp32 MFMA, LDS-tiled)
 * Replaces every nn.Linear / Conv2d-as-matmul on the path: q/k/v/out_proj
 * (hf:modeling_clip.py:309-311,:333), fc1/fc2 (:347-349), patch_embedding (:209),
 * visual/text_projection (:751,:713), nn.MultiheadAttention in/out projections
 * (training/patch_text_aggregation.py:33,:42), the logits matmul
 * (training/CLIP_image_distillation.py:549) and all of their autograd transposes.
 *
 * layout bits say which index is contiguous in memory:
 *   DCLIP_A_KMAJOR : A stored [M][K] (lda >= K), else stored [K][M] (lda >= M)
 *   DCLIP_B_KMAJOR : B stored [N][K] (ldb >= K, the nn.Linear weight layout), else [K][N]
 * so  forward  y = x W^T      -> A_KMAJOR | B_KMAJOR
 *     dgrad    dx = dy W      -> A_KMAJOR            (B = W read as [K=N_out][N=in])
 *     wgrad    dW = dy^T x    -> 0                   (A = dy as [K=rows][M=out], B = x as [K=rows][N=in])
 * The contiguous dimension of each operand, and ldc, must be multiples of 4 (16-byte vectors).
 *
 * epilogue bits (applied in this order to v = acc * alpha):
 *   BIAS      v += bias[n]
 *   GELU      if (aux) aux[m,n] = v;  v = v * sigmoid(1.702 v)        (hf:activations.py:122)
 *   DGELU     v *= d/dx quick_gelu (aux[m,n])                          (its backward)
 *   RESIDUAL  v += residual[m,n]   (same ld as C)
 *   ACCUM     C[m,n] += v  instead of  C[m,n] = v
 *   A_ROWSUM  additionally aux[m] = sum_k A[m,k] (aux is float[M] here; [K][M]-major A only, not with GELU/DGELU):
 *             with A = dy in the wgrad layout this is the nn.Linear BIAS gradient, read off the operand
 *             fragments the weight-gradient GEMM feeds to the matrix cores anyway — dy is not read a second time.
 * split_k > 1 partitions K over split_k workgroups per tile; partial tiles go to `workspace`
 * (dclip_gemm_f32_workspace bytes) and are summed in fixed order by a second launch, so the
 * result is run-to-run deterministic.  split_k == 0 lets the library choose.
 */
#define DCLIP_A_KMAJOR 1
#define DCLIP_B_KMAJOR 2

#define DCLIP_EPI_BIAS 1
#define DCLIP_EPI_GELU 2
#define DCLIP_EPI_DGELU 4
#define DCLIP_EPI_RESIDUAL 8
#define DCLIP_EPI_ACCUM 16
#define DCLIP_EPI_A_ROWSUM 32

size_t dclip_gemm_f32_workspace(int M, int N, int K, int layout, int split_k);
int dclip_gemm_f32(const float* A, const float* B, float* C, const float* bias, const float* residual,
                   float* aux, int M, int N, int K, int lda, int ldb, int ldc, int layout, int epilogue,
                   float alpha, int split_k, void* workspace, size_t workspace_bytes, void* stream);

/* Column sums  out[n] (+)= sum_m X[m,n]  — bias gradients of every nn.Linear above. */
size_t dclip_colsum_f32_workspace(int M, int N);
int dclip_colsum_f32(const float* X, float* out, int M, int N, int ldx, int accumulate, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (biased variance, eps inside the sqrt) — nn.LayerNorm at
 * hf:modeling_clip.py:364-366 (layer_norm1/2), :642 (pre_layrnorm), :651 (post_layernorm),
 * :568 (final_layer_norm); training/patch_text_aggregation.py:35,:44 (norm_text/norm_image).
 * fwd saves mean and rstd per row.  bwd: dx = LN'(dy) (+ dresidual if non-null: the skip
 * connection's gradient is added in the same pass); dgamma/dbeta (+)= column reductions.
 */
int dclip_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                        float* rstd, int rows, int D, float eps, void* stream);
size_t dclip_layernorm_bwd_workspace(int rows, int D);
int dclip_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                        const float* rstd, const float* dresidual, float* dx, float* dgamma, float* dbeta,
                        int rows, int D, int accumulate_param_grads, void* workspace, size_t workspace_bytes,
                        void* stream);
/* The bf16 training path's form (configs c3 / c5; nothing in the reference, which is fp32-only —
 * training/CLIP_image_distill_training.py:40): the same pass also writes, when non-null,
 *   dx_bf16   [rows][D] bf16 copy of dx — the A operand of the data- and weight-gradient GEMMs that follow;
 *   dx_colsum [D]       column sums of dx — the bias gradient of the nn.Linear whose output gradient dx is
 *                       (out_proj.bias for LayerNorm2, the layer below's fc2.bias for LayerNorm1: hf:modeling_clip.py:346-383).
 * Same workspace as dclip_layernorm_bwd. */
int dclip_layernorm_bwd_ex(const float* dy, const float* x, const float* gamma, const float* mean,
                           const float* rstd, const float* dresidual, float* dx, void* dx_bf16, float* dgamma,
                           float* dbeta, float* dx_colsum, int rows, int D, int accumulate_param_grads,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-head self-attention core, head_dim = 64:  O = softmax(Q K^T / 8 [+ causal]) V
 * — eager_attention_forward, hf:modeling_clip.py:259-277 (vision: no mask; text: causal,
 * :546-551).  qkv is the fused projection output [B*S, 3*H*64] = [q | k | v] per token;
 * out is [B*S, H*64]; lse [B, H, S] keeps log-sum-exp per query row for the backward, which
 * recomputes P instead of storing it.
 */
int dclip_attention_fwd(const float* qkv, float* out, float* lse, int B, int S, int H, int causal,
                        void* stream);
int dclip_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                        float* dqkv, float* delta /* scratch [B*H*S] */, int B, int S, int H, int causal,
                        void* stream);
/* Workspace form of the backward (what dclip_amd.ops.attention_bwd calls).  `workspace` (16-byte aligned,
 * >= dclip_attention_bwd_workspace bytes) holds delta and — for long non-causal sequences (S > 80: the 197 tokens of
 * ViT-B/16, the 257 of ViT-L/14) — the dS^T blocks [B*H][Sp][Sp], Sp = S rounded up to 32 (S <= 512; longer sequences keep the two-kernel split): dS is formed ONCE in the dK/dV kernel
 * and read back by the dQ kernel (5 MFMA products instead of the 7 of the two-kernel recompute split); results equal
 * dclip_attention_bwd's up to the summation order of delta.  Same reference arithmetic: hf:modeling_clip.py eager
 * attention backward (autograd of :333-349). */
size_t dclip_attention_bwd_workspace(int B, int S, int H, int causal);
int dclip_attention_bwd_ws(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                           void* workspace, size_t workspace_bytes, int B, int S, int H, int causal, void* stream);
/* CLS-only form for the LAST vision layer: the model reads only row 0 of the final hidden state
 * (hf:modeling_clip.py:650), so that layer's attention output is needed for one query row per (image, head).
 * out [B, H*64], lse [B, H].  _bwd writes d k / d v for every row and d q for the CLS rows of dqkv [B*S, 3*H*64];
 * the caller zero-fills dqkv first (d q of the other rows is exactly zero).  delta: scratch [B*H]. */
int dclip_attention_cls_fwd(const float* qkv, float* out, float* lse, int B, int S, int H, void* stream);
/* Same idea for the frozen text tower's last layer: one query row per caption at position rows[b] (its first EOS,
 * hf:modeling_clip.py:574-581) against keys 0..rows[b] (causal).  Forward only.  out [B, H*64], lse [B, H]. */
int dclip_attention_row_fwd(const float* qkv, const int32_t* rows, float* out, float* lse, int B, int S, int H,
                            void* stream);
int dclip_attention_cls_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                            float* delta, int B, int S, int H, void* stream);

/* ------------------------------------------------------------------------------------------
 * Embedding plumbing.
 * im2col: pixel_values [B,C,Himg,Wimg] -> patch rows [B*g*g, C*p*p] so that
 *   patch_embedding (Conv2d stride=p, no bias; hf:modeling_clip.py:209-210) is one GEMM.
 * vision_assemble: x[b,0,:] = class_embedding + pos[0]; x[b,1+i,:] = patch[b,i,:] + pos[1+i]
 *   (hf:modeling_clip.py:212-217).  _bwd copies d x[:,1:,:] into compact d patch rows; d pos is the
 *   column sum of d x viewed as [B, S*D] (dclip_colsum_f32) and d class is its first D entries.
 * text_embed: x[b,t,:] = token_embedding[ids[b,t]] + position_embedding[t] (hf:modeling_clip.py:250-255);
 *   _bwd scatter-adds d x into d token_embedding (float atomics; caller zeroes or accumulates).
 * first_eos: index of the first EOS id per row, 0 if absent (hf:modeling_clip.py:574-581).
 * gather_rows: out[b,:] = x[b, idx[b], :] (idx == NULL: row 0, the CLS token, hf:modeling_clip.py:650);
 * scatter_rows: its transpose, writing the whole [B,S,D] gradient (zeros off the selected row).
 */
int dclip_im2col(const float* pixels, float* cols, int B, int C, int Himg, int Wimg, int patch, void* stream);
/* Same gather with a bf16 destination (row length ldc >= C*patch*patch, multiple of 4; patch % 4 == 0): the A operand
 * of the frozen towers' bf16 patch-embedding GEMM, written in one pass instead of im2col + cast. */
int dclip_im2col_bf16(const float* pixels, void* cols, int B, int C, int Himg, int Wimg, int patch, int ldc, void* stream);
int dclip_vision_assemble_fwd(const float* patch, const float* cls, const float* pos, float* x, int B, int S,
                              int D, void* stream);
int dclip_vision_assemble_bwd(const float* dx, float* dpatch, int B, int S, int D, void* stream);
int dclip_text_embed_fwd(const int64_t* ids, const float* tok, const float* pos, float* x, int B, int T, int D,
                         int vocab, void* stream);
int dclip_text_embed_bwd(const int64_t* ids, const float* dx, float* dtok, int B, int T, int D, int vocab,
                         void* stream);
int dclip_first_eos(const int64_t* ids, int32_t* idx, int B, int T, int64_t eos_id, void* stream);
int dclip_gather_rows(const float* x, const int32_t* idx, float* out, int B, int S, int D, void* stream);
int dclip_scatter_rows(const float* dout, const int32_t* idx, float* dx, int B, int S, int D, void* stream);

/* ------------------------------------------------------------------------------------------
 * Losses.
 * normalize_rows: xhat = x / max(||x||, eps), inv[b] = 1/max(||x||,eps)   (F.normalize,
 *   training/CLIP_image_distillation.py:545-546,:569-570; eps = 1e-12).
 *   _bwd: dx = inv * (dxhat - xhat <dxhat, xhat>); rows where the clamp was active get dxhat/eps.
 * contrastive_lse: for local rows a_i (i < Bl) against ALL columns b_j (j < Bg), both already
 *   normalised:  lse[i] = log sum_j exp(<a_i,b_j> * inv_temp),  diag[i] = <a_i, b_{i+offset}> * inv_temp.
 *   The [Bl,Bg] logits never reach HBM: MFMA tiles are reduced to per-row (max, sum-exp) partials with
 *   wave shuffles in the GEMM epilogue (training/CLIP_image_distillation.py:549,:556-557).
 * contrastive_grad: da_i = coef * sum_j ( exp(z_ij - lse_row[i]) + exp(z_ij - lse_col[j])
 *   - 2*[j == i+offset] ) * b_j   with z_ij = <a_i,b_j>*inv_temp — the gradient of
 *   (CE_rows + CE_cols) w.r.t. the normalised local rows; coef = inv_temp / (2*Bg).
 *   (This round the [Bl,Bg] weight tile is staged through `workspace`; see DESIGN.md.)
 * cosine_loss: loss_sum = sum_b (1 - cos_b), cos[b] kept for the backward, which returns
 *   ds = -coef * d cos / d s   (training/CLIP_image_distillation.py:564-576).
 * sub_reduce: out (+)= scale * sum_i (a[i] - b[i])  (b may be NULL) — fixed-order scalar reduction.
 */
int dclip_normalize_rows_fwd(const float* x, float* xhat, float* inv_norm, int B, int P, float eps, void* stream);
int dclip_normalize_rows_bwd(const float* dxhat, const float* xhat, const float* inv_norm, float* dx, int B,
                             int P, float eps, int accumulate, void* stream);
size_t dclip_contrastive_workspace(int Bl, int Bg, int P);
int dclip_contrastive_lse(const float* a_local, const float* b_global, float* lse, float* diag, int Bl, int Bg,
                          int P, int offset, float inv_temp, void* workspace, size_t workspace_bytes,
                          void* stream);
int dclip_contrastive_grad(const float* a_local, const float* b_global, const float* lse_row,
                           const float* lse_col, float* da_local, int Bl, int Bg, int P, int offset,
                           float inv_temp, float coef, void* workspace, size_t workspace_bytes, void* stream);
int dclip_cosine_loss_fwd(const float* s, const float* t, float* loss_sum, float* cos, int B, int P,
                          void* stream);
int dclip_cosine_loss_bwd(const float* s, const float* t, const float* cos, float* ds, int B, int P, float coef,
                          int accumulate, void* stream);
int dclip_sub_reduce(const float* a, const float* b, float* out, int n, float scale, int accumulate,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * Meta-teacher tail (training/patch_text_aggregation.py).
 * cross_attention: softmax(Q K^T / 8) V with separate query and key/value sequences — the core of
 *   nn.MultiheadAttention(E, heads) as called at :33 and :42 (head_dim 64, dropout 0, no masks: zero-padded
 *   rows ARE attended).  q [B*Lq, H*64]; kv [B*Lk, 2*H*64] = [k | v] (the packed in_proj rows
 *   E..3E applied in one GEMM); out [B*Lq, H*64]; lse [B,H,Lq].  Same MFMA kernels as dclip_attention_*.
 *   _bwd returns dq [B*Lq, E] and dkv [B*Lk, 2E]; delta is scratch [B*H*Lq].
 * aggregation: m = mean_l x_l; s_l = <x_l,m> / (max(|x_l|,1e-8) max(|m|,1e-8)); w = softmax(s / temperature);
 *   out[b,:] (+)= out_scale * sum_l w_l x_l   (:243-265; the 0.5/0.5 mix of :647 is out_scale + accumulate).
 *   weights [B,L] are kept for _bwd, which returns d x for d out (already including out_scale).  L <= 96.
 * pack_tokens: the token filter of training/text_tokenizer.py:195-213 plus the zero padding of
 *   patch_text_aggregation.py:606-620 on device: out[b,i,:] = tokens[b,1+i,:] for i < eos[b]-1, zero beyond;
 *   a caption with no word tokens contributes its sentence embedding as row 0.
 * mask_rows: x[b,r,:] = 0 for r >= count[b]  (zero padding of region embeddings, :555-581).
 */
int dclip_cross_attention_fwd(const float* q, const float* kv, float* out, float* lse, int B, int Lq, int Lk,
                              int H, void* stream);
int dclip_cross_attention_bwd(const float* q, const float* kv, const float* out, const float* dout,
                              const float* lse, float* dq, float* dkv, float* delta, int B, int Lq, int Lk,
                              int H, void* stream);
int dclip_aggregation_fwd(const float* x, float* out, float* weights, int B, int L, int E, float temperature,
                          float out_scale, int accumulate, void* stream);
int dclip_aggregation_bwd(const float* x, const float* weights, const float* dout, float* dx, int B, int L,
                          int E, float temperature, float out_scale, void* stream);
int dclip_pack_tokens(const float* tokens, const float* sentence, const int32_t* eos, float* out, int B, int T,
                      int Tmax, int P, void* stream);
int dclip_mask_rows(float* x, const int32_t* count, int B, int R, int E, void* stream);
/* NaN / Inf guards of the reference's teacher glue (training/patch_text_aggregation.py:497-499, :542, :649): x is
 * [groups][rows][E]; a group holding any non-finite value becomes zeros.  mode 0: detect, write flags[groups]
 * (1 = replaced) and zero in place; mode 1: zero the groups already flagged (the guard's backward). */
int dclip_sanitize_groups(float* x, int32_t* flags, int groups, int rows, int E, int mode, void* stream);

/* ------------------------------------------------------------------------------------------
 * Checkpoint consumers: retrieval and zero-shot evaluation on the similarity kernel.
 * The reference materialises caption x image similarities chunk by chunk and argsorts every row / column
 * (eval_scripts/flickr30k_eval.py:16-88, :249-266; eval_scripts/test_zero_shot_ImageNet.py:82-103).  The rank of a
 * ground truth is the number of candidates that score strictly higher, so it is computed without the matrix:
 *   rowdot_gather: out[i] = <a_i, b_{idx[i]}>            (the ground truth's own score; idx NULL = i)
 *   rank_count:    count[i] = #{ j < Bk, j != gt[i] : <q_i, c_j> > thresh[i] }   (MFMA tiles, per-row counts in the
 *                  epilogue; gt NULL = i.  The ground truth itself is excluded so that rounding differences between
 *                  its two evaluations cannot count it as "higher than itself".)
 * Inputs are L2-normalised rows (dclip_normalize_rows_fwd).  Exact ties count as not-higher (argsort's order
 * among equal scores is unspecified in the reference).
 */
size_t dclip_rank_count_workspace(int Bq, int Bk);
int dclip_rowdot_gather(const float* a, const float* b, const int32_t* idx, float* out, int Bq, int Bk, int P,
                        void* stream);
int dclip_rank_count(const float* queries, const float* candidates, const float* thresh, const int32_t* gt,
                     int32_t* count, int Bq, int Bk, int P, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Region-crop front end: what training/image_tokenizer.py:100-110 does per box on the
 * host with PIL — `image.crop(box)` (zero padding outside the image), `Resize((S,S))` (Pillow's antialiased
 * two-pass BILINEAR in 8-bit fixed point) and `ToTensor()` (uint8/255, CHW, no mean/std) — for all boxes of a
 * batch in three launches, bit-exact with Pillow 12.2.
 *   images [B, Hmax, Wmax, 3] uint8 (HWC, each image in the top-left corner), dims [B,2] = (h, w),
 *   boxes [NR,5] int32 = (image index, x1, y1, x2, y2) with x2 > x1, y2 > y1,  out [NR, 3, S, S] fp32.
 *   max_crop_h / max_crop_w: the largest (y2-y1) / (x2-x1) in `boxes` (sizes the workspace and the tap count).
 */
size_t dclip_crop_resize_workspace(int NR, int S, int max_crop_h, int max_crop_w);
int dclip_crop_resize_u8(const uint8_t* images, const int32_t* dims, const int32_t* boxes, float* out, int B,
                         int Hmax, int Wmax, int NR, int S, int max_crop_h, int max_crop_w, void* workspace,
                         size_t workspace_bytes, void* stream);

/* Student-side image preprocessing (the `clip_preprocess(images=...)` call of MultiModalDataset.__getitem__,
 * training/CLIP_image_distillation.py:349-350; arithmetic = HF CLIPImageProcessor, PIL backend): decoded RGB uint8
 * images -> shortest edge resized to S with Pillow's BICUBIC two-pass 8-bit resample (long edge = int(S*long/short)),
 * centred S x S window, float32(float64(v) * (1/255)), then (x - mean) / std in fp32, CHW.  Bit-exact with the host
 * library.  images [B,Hmax,Wmax,3] (each image in the top-left dims[b] = (h,w) corner); mean/stdv are HOST float[3].
 */
size_t dclip_clip_preprocess_workspace(int B, int Hmax, int Wmax, int S);
int dclip_clip_preprocess_u8(const uint8_t* images, const int32_t* dims, float* out, int B, int Hmax, int Wmax, int S,
                             const float* mean, const float* stdv, void* workspace, size_t workspace_bytes,
                             void* stream);

/* ------------------------------------------------------------------------------------------
 * bf16 forward path for FROZEN towers (BASELINE configs c3 / c5, "bf16 MFMA"): the teacher's region encoder
 * (training/image_tokenizer.py:119-120) and the frozen text tower never need gradients, so their GEMMs may run on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation (16x the fp32 MFMA rate).  The residual stream, LayerNorm
 * statistics, softmax and every epilogue stay fp32; only GEMM INPUTS are bf16.  Opt-in (precision="bf16"); the
 * default everywhere, and the benched config c2, is exact fp32.
 *   gemm_bf16: C[M,N] = epilogue( A[M,K] W[N,K]^T ), A and W bf16 K-major (lda, ldw multiples of 8), C fp32 or bf16;
 *              epilogue bits BIAS | GELU | RESIDUAL (fp32 residual, fp32 output only).
 *   cast_f32_bf16: row-wise fp32 -> bf16 (round to nearest even), destination rows zero padded to ldy.
 *   layernorm_fwd_bf16: nn.LayerNorm with fp32 statistics and a bf16 result (the next GEMM's A operand).
 */
int dclip_gemm_bf16(const void* A, const void* W, void* C, const float* bias, const float* residual, int M, int N,
                    int K, int lda, int ldw, int ldc, int epilogue, int out_bf16, void* stream);
int dclip_cast_f32_bf16(const float* x, void* y, int rows, int cols, int ldx, int ldy, void* stream);
/* bf16 TRAINING path (the student of configs c3 / c5; opt-in `student_precision="bf16"`, the default and the benched
 * config c2 stay exact fp32).  Forward, dgrad and wgrad GEMMs of the trainable vision tower — the autograd of
 * `self.student.get_image_features(...)` (training/CLIP_image_distillation.py:601) — run through dclip_gemm_bf16 with fp32
 * accumulation and fp32 master weights; residual stream, LayerNorm, softmax and the attention core stay fp32.
 *   gemm_bf16_ex         dclip_gemm_bf16 + `aux` (bf16 [M][ldc]): with GELU the pre-activation (bias included) is stored
 *                        there and the activation is taken of the STORED value; with DGELU (no GELU) the result is
 *                        multiplied by quick_gelu'(aux).
 *   layernorm_fwd_bf16_stats   layernorm_fwd_bf16 that also returns mean / rstd (fp32 [rows]) for the backward.
 *   transpose_to_bf16    x [rows][cols] (fp32, or bf16 when x_is_bf16) -> yT [cols][ldyT] bf16 (ldyT >= rows, multiple
 *                        of 8, zero padded) and, if y_copy != NULL, the untransposed bf16 copy [rows][ldy]: the
 *                        token-contiguous operands of dW = (dY^T)(X^T)^T and the A operand of the dgrad GEMM.
 *   rowsum_bf16          out[r] = sum of row r of a bf16 matrix [R][ld] (a bias gradient from dY^T). */
int dclip_gemm_bf16_ex(const void* A, const void* W, void* C, const float* bias, const float* residual, void* aux, int M,
                       int N, int K, int lda, int ldw, int ldc, int epilogue, int out_bf16, void* stream);
int dclip_layernorm_fwd_bf16_stats(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                   int rows, int D, float eps, void* stream);
int dclip_transpose_to_bf16(const void* x, int x_is_bf16, void* yT, void* y_copy, int rows, int cols, int ldx, int ldyT,
                            int ldy, void* stream);
/* Short-sequence self-attention with bf16 I/O for the bf16 training student (configs c3 / c5; eager_attention_forward,
 * hf:modeling_clip.py:259-277, arithmetic in fp32 as in dclip_attention_fwd / _bwd): qkv16 [B*S][3*H*64], out16 / dout16
 * [B*S][H*64] and dqkv16 [B*S][3*H*64] are bf16, lse [B*H][S] fp32.  Forward S <= 80, backward S <= 64. */
int dclip_attention_fwd_io16(const void* qkv16, void* out16, float* lse, int B, int S, int H, int causal, void* stream);
int dclip_attention_bwd_io16(const void* qkv16, const void* out16, const void* dout16, const float* lse, void* dqkv16, int B,
                             int S, int H, int causal, void* stream);
/* Every GEMM weight of a training tower converted in one launch (bf16 training path; no counterpart in the reference):
 * `refs` = device array of `ntensors` records {const float* src [rows][cols]; uint16* dst [rows][ld] (or null); uint16* dstT
 * [cols][ldT] (or null); int rows, cols, ld, ldT, tile0, tiles_c} (dclip_mt_weights_record_bytes() bytes each), 64x64
 * tiles: tiles_c = ceil(max(cols, ld) / 64), tile rows = ceil(max(rows, ldT) / 64), tile0 = running sum, ascending.
 * cols, ld % 4 == 0; ldT % 8 == 0; padding (ld > cols, ldT > rows) is written as zeros. */
int dclip_mt_weights_record_bytes(void);
int dclip_mt_weights_bf16(const void* refs, int ntensors, int total_tiles, void* stream);
int dclip_rowsum_bf16(const void* x, float* out, int R, int n, int ld, void* stream);
/* Split-K bf16 GEMM for the weight gradients (few output tiles, contraction over all tokens): fp32 C [M][ldc], no
 * epilogue; `splits` K-slices per 128x128 tile write fp32 partials into the caller's workspace
 * (dclip_gemm_bf16_splitk_workspace bytes), a second kernel adds them in fixed order (deterministic).
 * dclip_gemm_bf16_splitk_plan(M, N, K) = the split count to pass (1: few enough tiles already). */
int dclip_gemm_bf16_splitk_plan(int M, int N, int K);
size_t dclip_gemm_bf16_splitk_workspace(int M, int N, int splits);
int dclip_gemm_bf16_splitk(const void* A, const void* W, float* C, int M, int N, int K, int lda, int ldw, int ldc,
                           int splits, void* workspace, size_t workspace_bytes, void* stream);
/* The same weight gradient WITHOUT the transposes: dY [K = tokens][lddy >= M] and X [K][ldx >= N] bf16 as the backward has
 * them (token-major); C [M][ldc] fp32 = dY^T X.  Split-K over the tokens on the token-major form of the ping-pong kernel
 * (operands out of LDS through the transposing read), fixed-order reduce through `workspace`
 * (dclip_gemm_bf16_splitk_workspace(M, N, splits) bytes).  dclip_gemm_bf16_wgrad_tokmajor_plan(M, N, K) = the split count to
 * pass, or 0 when this form does not apply (K % 64, M or N % 8, too few work items): transpose and use
 * dclip_gemm_bf16_splitk then.  dclip_colsum_bf16: the bias gradient sum over tokens of a bf16 [M][ldx] matrix
 * (workspace as dclip_colsum_f32_workspace). */
int dclip_gemm_bf16_wgrad_tokmajor_plan(int M, int N, int K);
int dclip_gemm_bf16_wgrad_tokmajor(const void* dY, const void* X, float* C, int M, int N, int K, int lddy, int ldx, int ldc,
                                   int splits, void* workspace, size_t workspace_bytes, void* stream);
int dclip_colsum_bf16(const void* X, float* out, int M, int N, int ldx, int accumulate, void* workspace,
                      size_t workspace_bytes, void* stream);
/* Softmax attention of the frozen towers on bf16 q/k/v (the fused projection [B*S, 3*H*64] as written by
 * dclip_gemm_bf16 with out_bf16): fp32 scores / softmax, bf16 P and context [B*S, H*64].  Forward only. */
int dclip_attention_fwd_bf16(const void* qkv, void* out, int B, int S, int H, int causal, void* stream);
/* One attention output row per sequence for the LAST layer of a frozen bf16 tower: rows == NULL -> query row 0 against all
 * S keys (only the CLS row of the final hidden state is read, hf:modeling_clip.py:650); rows [B] int32 -> query row rows[b]
 * against keys 0..rows[b] (the pooled first-EOS row of the causal text tower, hf:modeling_clip.py:574-581).
 * qkv [B*S, 3*H*64] bf16, out [B, H*64] bf16, S <= 512; fp32 scores / softmax / accumulation. */
int dclip_attention_row_fwd_bf16(const void* qkv, const int32_t* rows, void* out, int B, int S, int H, void* stream);
/* Training forms of the bf16 attention (the bf16 student of configs c3 / c5; eager_attention_forward, hf:modeling_clip.py:259-277,
 * and its gradient): bf16 q/k/v, context, d(context) and dq|dk|dv; fp32 scores, softmax and dS; P and dS rounded to bf16 for
 * the products they feed.  The forward (S <= 288, not 257) also writes lse [B*H][S] fp32 = log-sum-exp of the scaled scores;
 * the backward (S <= 64) consumes it. */
int dclip_attention_fwd_bf16_lse(const void* qkv, void* out, float* lse, int B, int S, int H, int causal, void* stream);
int dclip_attention_bwd_bf16(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int S,
                             int H, int causal, void* stream);
int dclip_layernorm_fwd_bf16(const float* x, const float* gamma, const float* beta, void* y, int rows, int D,
                             float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser tail (pulled into the timed step).
 * sumsq / clip_coef: global-norm clipping as torch.nn.utils.clip_grad_norm_ does it, which is what Lightning's
 *   Trainer(gradient_clip_val=0.5) applies (training/CLIP_image_distill_training.py:41): each tensor writes
 *   dclip_sumsq_blocks(n) partial sums; clip_coef reduces ALL partials to coef = min(1, max_norm/(norm+1e-6))
 *   on the device (no host sync) and optionally returns the norm.
 * adamw: torch.optim.AdamW update (training/CLIP_image_distillation.py:680; decoupled weight decay, bias
 *   correction from the integer `step` >= 1); the gradient is multiplied by *grad_scale (device scalar, may be
 *   NULL) first, which is where the clip coefficient goes.
 */
int dclip_sumsq_blocks(size_t n);
int dclip_sumsq_f32(const float* x, size_t n, float* partial, void* stream);
int dclip_clip_coef(const float* partial, int n, float max_norm, float* coef, float* norm_out, void* stream);
int dclip_adamw_f32(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int step, const float* grad_scale, void* stream);
/* Multi-tensor forms: ONE launch for all parameters of a group.  `refs` is a DEVICE array of `ntensors` records of
 * dclip_mt_record_bytes() bytes each: { float* p; const float* g; float* m; float* v; uint64_t n; int32_t step;
 * int32_t chunk0; } where chunk0 is the running sum of ceil(n / dclip_mt_chunk_elems()) over the preceding
 * tensors and total_chunks the sum over all.  mt_sumsq writes one partial per chunk (feed them to dclip_clip_coef);
 * mt_adamw applies the update of dclip_adamw_f32 to every tensor. */
int dclip_mt_record_bytes(void);
int dclip_mt_chunk_elems(void);
int dclip_mt_sumsq_f32(const void* refs, int ntensors, int total_chunks, float* partial, void* stream);
int dclip_mt_adamw_f32(const void* refs, int ntensors, int total_chunks, float lr, float beta1, float beta2,
                       float eps, float weight_decay, const float* grad_scale, void* stream);
/* torch.optim.Adam (L2 coupled into the gradient: g += weight_decay * p) — replaces the teacher trainer's
 * `optim.Adam(trainable_params, lr=args.learning_rate)` (training/train_contrastive_teacher.py:245-248). */
int dclip_mt_adam_f32(const void* refs, int ntensors, int total_chunks, float lr, float beta1, float beta2,
                      float eps, float weight_decay, const float* grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Small elementwise helpers used between the ops above (all fp32, 16-byte vectorised).
 */
int dclip_axpby(const float* x, float* y, float a, float b, size_t n, void* stream); /* y = a*x + b*y */
int dclip_fill(float* y, float v, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCLIP_HIP_H */
