/*
 * equss_hip.h - C ABI of the MI355X (gfx950) EQUSS hot-path library (libequss_hip.so).
 *
 * The reference (pitlover/Expand-and-Quantize-for-Unsupervised-Semantic-Segmentation) is pure
 * Python: its "operator interface" for this path is the forward() of a handful of nn.Modules that
 * call stock ATen ops.  Each entry point below replaces the op sequence of one of those forward()s
 * and cites it.  The Python modules that mirror the reference classes bind these symbols with
 * ctypes (native.py of the package); nothing here depends on PyTorch.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer on the current HIP device unless noted; the caller owns all
 *    memory (inputs, outputs, workspaces); the library allocates nothing and frees nothing.
 *  - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and the call returns
 *    without synchronising the host.
 *  - Return value: 0 = success; > 0 = EQUSS_E_* argument/shape error (nothing was launched);
 *    < 0 = minus the hipError_t of a failed launch.  equss_last_error() gives a message for the
 *    calling thread.  No C++ exception and no abort crosses this boundary.
 *  - Token-major layout: activations are (rows, channels) row-major with rows = batch * tokens;
 *    this is the reference's (b, C, h, w) tensor viewed as (b, h, w, C).
 */
#ifndef EQUSS_HIP_H
#define EQUSS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EQUSS_ABI_VERSION 2

enum {
    EQUSS_OK = 0,
    EQUSS_E_NULL = 1,        /* a required pointer is NULL */
    EQUSS_E_SHAPE = 2,       /* a dimension is non-positive or violates a divisibility rule */
    EQUSS_E_UNSUPPORTED = 3, /* valid request that no kernel instance covers */
    EQUSS_E_ALIGN = 4,       /* pointer or leading dimension not aligned as required */
    EQUSS_E_WORKSPACE = 5    /* workspace smaller than the matching *_workspace_bytes() */
};

/* normalisation applied to z rows and codebook rows before the distance
 * (model/dino_pqgo.py:613-640, model/quantizer.py:120-138) */
enum { EQUSS_NORM_NONE = 0, EQUSS_NORM_L2 = 1, EQUSS_NORM_ZSCORE = 2 };

/* epilogues of equss_gemm_bf16 */
enum {
    EQUSS_EPI_BF16 = 0,       /* out_bf16 = acc + bias                                      */
    EQUSS_EPI_GELU_BF16 = 1,  /* out_bf16 = gelu_erf(acc + bias)   (vision_transformer.py:58-60) */
    EQUSS_EPI_RELU_BF16 = 2,  /* out_bf16 = relu(acc + bias)       (dino_pqgo.py:108-112)   */
    EQUSS_EPI_F32 = 3,        /* out_f32  = acc + bias                                      */
    EQUSS_EPI_RESID_F32 = 4   /* out_f32 += acc + bias  (residual, vision_transformer.py:110-111) */
};

int equss_abi_version(void);
const char* equss_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Per-kernel-class device timing (HIP events recorded on the caller's stream around every launch
 * the whole-path entry points make).  The reference's only instrumentation is host wall clock
 * around the forward (train.py:59-98) and two device synchronisations around its loss
 * (model/dino_pqgo.py:145-150); this replaces both without synchronising inside the path.
 *   equss_profile_enable(max_pairs): create an event pool and start recording; 0 disables and
 *     destroys the pool.  While the pool is exhausted launches simply go unrecorded.
 *   equss_profile_read(ms, launches, reset): synchronises the recorded events and ADDS the elapsed
 *     milliseconds / launch counts per class into ms[EQUSS_K_COUNT] / launches[EQUSS_K_COUNT]
 *     (HOST arrays); reset != 0 recycles the pool afterwards.
 * ------------------------------------------------------------------------------------------- */
enum {
    EQUSS_K_PATCH = 0,     /* patch gather + patch GEMM + token assembly */
    EQUSS_K_LAYERNORM = 1,
    EQUSS_K_GEMM_QKV = 2,
    EQUSS_K_ATTENTION = 3,
    EQUSS_K_GEMM_PROJ = 4,
    EQUSS_K_GEMM_FC1 = 5,
    EQUSS_K_GEMM_FC2 = 6,
    EQUSS_K_HEAD_GEMM1 = 7,
    EQUSS_K_HEAD_GEMM2 = 8,
    EQUSS_K_PQ_PREP = 9,
    EQUSS_K_PQ_ASSIGN = 10,
    EQUSS_K_PROBE_TABLE = 11,  /* low-resolution probe products + neighbour Gram entries */
    EQUSS_K_PROBE_PIXELS = 12, /* per-pixel interpolation, arg-max, losses */
    EQUSS_K_MLP_FUSED = 13,    /* fc1 + GELU + fc2 + residual + LayerNorm in one kernel */
    EQUSS_K_BLOCK_TAIL = 14,   /* proj + residual + LayerNorm2 + MLP + residual + LayerNorm in one kernel */
    EQUSS_K_PACK_WEIGHTS = 15, /* re-ordering of the block-tail weights when the caller has not prepared them */
    EQUSS_K_COUNT = 16
};
int equss_profile_enable(int max_pairs);
int equss_profile_read(double* ms, int64_t* launches, int reset);

/* ---------------------------------------------------------------------------------------------
 * Product-quantisation assign.
 * Replaces ProductQuantizerWrapper.forward + Codebook.forward (model/dino_pqgo.py:745-776,
 * :579-705) and the same core in model/quantizer.py:105-189, :383-542, :587-611:
 *   per sub-space p and row r:  zn = normalise(code[r, p*d:(p+1)*d]);  cn = normalise(codebook[p])
 *   dist[k] = (sum zn^2 + sum cn[k]^2) - 2 * <zn, cn[k]>;   idx = first arg-min over k
 *   emb = (gather_normalized ? cn : codebook[p])[idx];      zq = zn + (emb - zn)
 *   loss_sum[p] = sum over rows and d of (emb - zn)^2        (the mse numerator of :685-686)
 * All arithmetic is fp32; indices are int64 as torch.argmin returns them.
 *
 *   code       (rows, num_pq*d) fp32, leading dimension ld_code (elements, multiple of 4)
 *   codebooks  (num_pq, K, d) fp32 contiguous raw embedding weights
 *   idx        (num_pq, rows) int64 out
 *   zq         (rows, num_pq*d) fp32 out, leading dimension ld_zq; may be NULL
 *   loss_sum   (num_pq) fp32 out; may be NULL
 *   workspace  equss_pq_workspace_bytes() bytes, 256-byte aligned
 * Supported: d in {16,32,64,128}; K a multiple of 32, K <= 16*32*(128/d).
 * ------------------------------------------------------------------------------------------- */
size_t equss_pq_workspace_bytes(int rows, int num_pq, int K, int d);
int equss_pq_assign(const float* code, int64_t ld_code, const float* codebooks, int64_t* idx,
                    float* zq, int64_t ld_zq, float* loss_sum, void* workspace, size_t workspace_bytes,
                    int rows, int num_pq, int K, int d, int normalize, int gather_normalized,
                    void* stream);

/* equss_pq_assign for callers that cannot allocate or own device memory beyond inputs and outputs
 * (tight serving loops, harnesses that forbid hipMalloc, code that will run under HIP graph capture once one eager
 * call per GPU has been made: the first call on a device resolves the buffer's address and raises the kernels' LDS
 * limits, neither of which is a stream operation).  No counterpart
 * in the reference.  Same arguments, results and bits as equss_pq_assign minus the workspace: the
 * prepared codebooks and loss partials live in a buffer of equss_pq_static_scratch_bytes() (5 MiB) that
 * is part of the loaded library on every GPU.  Shapes whose equss_pq_workspace_bytes() exceeds it are
 * rejected with EQUSS_E_WORKSPACE (both shipped configurations fit: 8 x 256 x 128 needs 1.1 MiB,
 * 16 x 512 x 128 needs 4.1 MiB).  The call allocates and frees nothing.  All calls on one GPU share that
 * buffer: calls that may overlap on one GPU (different streams, different host threads) race and must be
 * serialised by the caller. */
size_t equss_pq_static_scratch_bytes(void);
int equss_pq_assign_static(const float* code, int64_t ld_code, const float* codebooks, int64_t* idx,
                           float* zq, int64_t ld_zq, float* loss_sum, int rows, int num_pq, int K, int d,
                           int normalize, int gather_normalized, void* stream);

/* softmax(-dist / tau) over the K codewords of every sub-space: the `distance_prob` tensor of
 * model/dino_pqgo.py:655,:699,:774 (and quantizer.py:151,:468 with tau = 1).
 *   prob (rows, num_pq*K) fp32 out. */
int equss_pq_prob(const float* code, int64_t ld_code, const float* codebooks, float* prob,
                  void* workspace, size_t workspace_bytes, int rows, int num_pq, int K, int d,
                  int normalize, float inv_tau, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Building blocks of the ViT / head forward (exported so each can be parity-tested alone).
 * ------------------------------------------------------------------------------------------- */

/* LayerNorm over the last dim, fp32 in -> bf16 out (nn.LayerNorm eps=1e-6,
 * model/dino/vision_transformer.py:98,:102,:160,:271).
 *   x (rows, C) fp32 with leading dimension ld_x; out (rows_out, C) bf16 with leading dimension
 *   ld_out.  If tokens_per_image > 0 the first token of every image (CLS) is dropped on output:
 *   input row b*tokens_per_image + 1 + t goes to output row b*(tokens_per_image-1) + t
 *   (model/dino/dino_featurizer.py:80-81).  out_f32 (same indexing, leading dimension C) may be
 *   NULL; when given it receives the un-rounded fp32 result as well. */
int equss_layernorm_bf16(const float* x, int64_t ld_x, const float* gamma, const float* beta,
                         void* out_bf16, int64_t ld_out, float* out_f32, int rows, int C,
                         int tokens_per_image, float eps, void* stream);

/* C = A * W^T + bias with bf16 operands, fp32 MFMA accumulation and a fused epilogue
 * (nn.Linear / 1x1 nn.Conv2d: vision_transformer.py:81,:89,:59,:62; dino_pqgo.py:104-112,:127-128).
 *   A (M, K) bf16, leading dimension lda;  W (N, K) bf16 (PyTorch [out, in] layout), ldw;
 *   bias (N) fp32 or NULL;  out (M, N) bf16 or fp32 by epilogue, leading dimension ldo.
 * Requires N % 128 == 0, K % 64 == 0, lda/ldw % 8 == 0, ldo % 4 == 0 (% 8 for a bf16 output: rows are
 * written in 16-byte pieces), 16-byte aligned A/W/out. */
int equss_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                    void* out, int64_t ldo, int M, int N, int K, int epilogue, void* stream);

/* Residual update fused with the LayerNorm that follows it:
 *   x += A * W^T + bias            (vision_transformer.py:110 / :111, fp32 residual stream, in place)
 *   ln_out = LayerNorm(x) * gamma + beta in bf16   (the norm2 of the same block, :111, or the norm1 of
 *            the next block, :107); ln_out may be NULL to skip it.
 *   A (M, K) bf16; W (N, K) bf16; x (M, N) fp32 with leading dimension ldx; ln_out (M, N) bf16.
 * N must be 384 (one workgroup tile then holds whole rows); other widths use equss_gemm_bf16 with
 * EQUSS_EPI_RESID_F32 followed by equss_layernorm_bf16. */
int equss_gemm_resid_ln_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                             float* x, int64_t ldx, const float* gamma, const float* beta, void* ln_out,
                             int64_t ld_ln, int M, int N, int K, float eps, void* stream);

/* The whole MLP of a transformer block fused with the residual update and the LayerNorm that follows
 * (vision_transformer.py:58-64 inside :106-114):
 *   x += gelu(a * W1^T + b1) * W2^T + b2 ;  ln_out = LayerNorm(x) * gamma + beta in bf16 (or NULL).
 *   a (M, C) bf16 = LayerNorm2 output; W1 (hidden, C) bf16; W2 (C, hidden) bf16; x (M, C) fp32 in place.
 * The (M, hidden) activation never reaches HBM.  C = 384 with hidden = 1536 only (ViT-S); other
 * widths use equss_gemm_bf16 (EQUSS_EPI_GELU_BF16) followed by equss_gemm_resid_ln_bf16. */
int equss_mlp_resid_ln_bf16(const void* a, int64_t lda, const void* w1, const float* b1, const void* w2,
                            const float* b2, float* x, int64_t ldx, const float* gamma, const float* beta,
                            void* ln_out, int64_t ld_ln, int M, int C, int hidden, float eps, void* stream);

/* Everything of a transformer block after its attention, in one kernel (vision_transformer.py:110-111
 * with :58-64):
 *   x += att * Wp^T + bp ;  ln_mid = LayerNorm2(x) ;  x += gelu(ln_mid * W1^T + b1) * W2^T + b2 ;
 *   ln_out = LayerNorm(x) * gamma + beta in bf16 (or NULL).
 * att (M, C) bf16 = attention output; ln_mid (M, C) bf16 is scratch that the kernel writes and reads back
 * (ln_out may be the same buffer).  Same restrictions and the same bits as equss_gemm_resid_ln_bf16
 * followed by equss_mlp_resid_ln_bf16. */
int equss_block_tail_bf16(const void* att, int64_t ld_att, const void* wp, const float* bp, const float* gamma2,
                          const float* beta2, void* ln_mid, int64_t ld_mid, const void* w1, const float* b1,
                          const void* w2, const float* b2, float* x, int64_t ldx, const float* gamma,
                          const float* beta, void* ln_out, int64_t ld_ln, int M, int C, int hidden, float eps,
                          void* stream);

/* Multi-head self-attention softmax(q k^T * scale) v without materialising the (N x N) matrix
 * (vision_transformer.py:81-88; the attention matrix it returns is discarded by
 * dino_featurizer.py:76-80).
 *   qkv (B*N, 3*H*64) bf16: the qkv Linear's output, columns ordered [3][H][64] (:81);
 *   out (B*N, H*64) bf16: heads re-interleaved as :88 does.  head_dim is fixed at 64. */
int equss_attention_bf16(const void* qkv, void* out, int B, int N, int H, float scale, void* stream);

/* Patch embedding + CLS + position embedding (vision_transformer.py:130-133,:200-211).
 *   img (B,3,H,W) fp32;  w_patch (C, 3*P*P) bf16;  b_patch (C) fp32;  cls (C) fp32;
 *   pos (1 + (H/P)*(W/P), C) fp32 already resized to this grid (:178-198 is input-independent,
 *   the host computes it once per image size);  x (B*(1+n), C) fp32 out.
 *   workspace: equss_patch_workspace_bytes() bytes. */
size_t equss_patch_workspace_bytes(int B, int H, int W, int P, int C);
int equss_patch_tokens(const float* img, const void* w_patch, const float* b_patch, const float* cls,
                       const float* pos, float* x, void* workspace, size_t workspace_bytes,
                       int B, int H, int W, int P, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-path entry points.
 * ------------------------------------------------------------------------------------------- */
typedef struct EqussVitBlock {
    const float* ln1_g; const float* ln1_b;
    const void* qkv_w;  const float* qkv_b;   /* (3C, C) bf16, (3C) fp32 */
    const void* proj_w; const float* proj_b;  /* (C, C) */
    const float* ln2_g; const float* ln2_b;
    const void* fc1_w;  const float* fc1_b;   /* (4C, C) */
    const void* fc2_w;  const float* fc2_b;   /* (C, 4C) */
} EqussVitBlock;

typedef struct EqussVit {
    int32_t embed_dim, depth, heads, patch;    /* C, 12, C/64, 8 or 16 */
    const void* patch_w; const float* patch_b; /* (C, 3*P*P) bf16, (C) fp32 */
    const float* cls;                          /* (C) fp32 */
    const float* norm_g; const float* norm_b;  /* final LayerNorm */
    const EqussVitBlock* blocks;               /* HOST array of `depth` entries */
    const void* tail_images;                   /* DEVICE, optional: equss_vit_pack_tail_images() of these weights, or NULL */
} EqussVit;

/* The fused kernel behind everything of a block after its attention (vision_transformer.py:110-111 with :58-64) streams
 * proj / fc1 / fc2 weights in the order its waves consume them.  A caller whose weights are frozen (the reference freezes
 * the backbone: dino_featurizer.py:20-21) writes that order once and hands it over in EqussVit.tail_images; with NULL
 * there, every forward re-orders the weights into its workspace first (one extra launch, ~32 MB of traffic).
 *   images: equss_vit_tail_image_bytes(vit) bytes, 256-byte aligned (0 = this embed dim has no such kernel); must be
 *   re-packed when proj_w / fc1_w / fc2_w of any block change. */
size_t equss_vit_tail_image_bytes(const EqussVit* vit);
int equss_vit_pack_tail_images(const EqussVit* vit, void* images, size_t images_bytes, void* stream);

/* DinoFeaturizer.forward (model/dino/dino_featurizer.py:64-83) = prepare_tokens + all blocks +
 * final norm + drop CLS (vision_transformer.py:227-239 with n=1).
 *   vit: HOST struct of device pointers;  pos: as equss_patch_tokens;
 *   feat_bf16 (B*n, C) bf16 out with leading dimension ld_feat (>= C, multiple of 8);
 *   feat_f32 (B*n, C) fp32 out or NULL.  n = (H/P)*(W/P). */
size_t equss_vit_workspace_bytes(const EqussVit* vit, int B, int H, int W);
int equss_vit_forward(const EqussVit* vit, const float* img, const float* pos, void* feat_bf16,
                      int64_t ld_feat, float* feat_f32, void* workspace, size_t workspace_bytes,
                      int B, int H, int W, void* stream);

/* VisionTransformer.get_intermediate_feat (model/dino/vision_transformer.py:227-239): the same forward,
 * returning what that method returns as `feat`: norm(x) after each of the last n_last blocks, CLS row
 * kept.  (Its other two results, the attention maps and qkv of those blocks, are never formed by the fused
 * attention kernel; the reference's only caller drops them, dino_featurizer.py:76-80.)
 *   feats  (n_last, B*(1+n), C) fp32 out, oldest block first (the order of the reference's list)
 *   x_taps (depth, B*(1+n), C) fp32 out or NULL: the residual stream after every block
 *   workspace: equss_vit_workspace_bytes() bytes. */
int equss_vit_intermediate_feat(const EqussVit* vit, const float* img, const float* pos, float* feats,
                                int n_last, float* x_taps, void* workspace, size_t workspace_bytes,
                                int B, int H, int W, void* stream);

/* fp32 parity mode of the same forward: every GEMM, the attention and the head evaluated in fp32
 * (exact fp32 products, fp32 accumulation; erff GELU; expf softmax), so that features, `code` and with
 * them the code indices can be compared with the reference's fp32 CPU run element by element.  Same
 * reference lines as equss_vit_forward / equss_expand_head; built for comparison, not for throughput.
 *   vit: an EqussVit whose *_w pointers address FP32 matrices in the same [out, in] layouts
 *   feat (B*n, C) fp32 out, or (B*(1+n), C) with the CLS rows when keep_cls != 0
 *   workspace: equss_vit_f32_workspace_bytes() bytes, 256-byte aligned. */
size_t equss_vit_f32_workspace_bytes(const EqussVit* vit, int B, int H, int W);
int equss_vit_forward_f32(const EqussVit* vit, const float* img, const float* pos, float* feat, int keep_cls,
                          void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);
/*   feat (rows, C) fp32; w1 (D, C), w2a (C, C), w2b (D, C) fp32 with their biases; hidden (rows, C) fp32
 *   scratch; code (rows, D) fp32 out = (f W1^T + b1) + (relu(f W2a^T + b2a) W2b^T + b2b), the reference's
 *   order of additions (dino_pqgo.py:127-128). */
int equss_expand_head_f32(const float* feat, const float* w1, const float* b1, const float* w2a, const float* b2a,
                          const float* w2b, const float* b2b, float* hidden, float* code, int rows, int C, int D,
                          void* stream);

/* Expansion head (model/dino_pqgo.py:104-112,:127-128):
 *   code = W1 f + b1 + W2b relu(W2a f + b2a) + b2b, evaluated as two GEMMs:
 *   h = relu(f W2a^T + b2a) written beside f, then code = [f | h] [W1 | W2b]^T + (b1 + b2b).
 *   fh (rows, 2C) bf16: columns [0,C) hold f on entry, [C,2C) are overwritten with h;
 *   w2a (C, C) bf16; b2a (C) fp32; w_cat (D, 2C) bf16 = [W1 | W2b]; b_cat (D) fp32 = b1 + b2b;
 *   code (rows, D) fp32 out. */
int equss_expand_head(void* fh, const void* w2a, const float* b2a, const void* w_cat,
                      const float* b_cat, float* code, int rows, int C, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Linear + cluster probes of the eval loop: UnSegEvaluator.forward with is_crf = False and
 * ClusterLookup.forward with alpha = None (model/evaluator.py:46-82, :93-111), called once per batch
 * by wrapper/PQGOWrapper.py:95.  The reference up-samples the (B, D, h, w) map bilinearly
 * (align_corners = False) to the label size and evaluates both probes per pixel; here the probes
 * are evaluated at (h, w) and their outputs interpolated, which is the same function of the inputs
 * (both probes are linear in the feature; the cosine's denominator comes from the neighbours' Gram
 * entries), so the up-sampled map is never written.
 *   feat     (B*h*w, D) fp32 token-major, leading dimension ld_feat (multiple of 4)
 *   probes   (64, D) fp32: rows [0, n_linear) = linear_probe.weight, rows [32, 32 + n_cluster) =
 *            cluster_probe.clusters normalised to unit length (:95), all other rows zero
 *   bias     (64) fp32: linear_probe.bias in [0, n_linear), zero elsewhere
 *   label    (B, H, W) int64; pixels with a label outside [0, n_linear) are left out of the
 *            cross-entropy (:72-80)
 *   linear_preds, cluster_preds (B, H, W) int64 out: first arg-max over classes (:68, :70)
 *   losses   (2) fp32 out: [0] mean cross-entropy over labelled pixels, [1] -mean max cosine (:106)
 *   feature_norm (B, H, W) fp32 out or NULL: max(|f_up|, 1e-12) per pixel, the denominator of :96
 *            (callers that train the probes need it to rebuild the cosine with autograd)
 *   workspace equss_probe_workspace_bytes() bytes, 256-byte aligned
 * D % 8 == 0; 1 <= n_linear, n_cluster <= 32.  All arithmetic is fp32.
 * ------------------------------------------------------------------------------------------- */
size_t equss_probe_workspace_bytes(int B, int h, int w, int H, int W);
int equss_probe_eval(const float* feat, int64_t ld_feat, const float* probes, const float* bias,
                     const int64_t* label, int64_t* linear_preds, int64_t* cluster_preds, float* losses,
                     float* feature_norm, void* workspace, size_t workspace_bytes, int B, int h, int w, int D,
                     int H, int W, int n_linear, int n_cluster, void* stream);

/* The same probes while they are trained (train.py:80-97 steps an optimiser over linear_probe.{weight,bias} and
 * cluster_probe.clusters through the losses of model/evaluator.py:46-82 on a detached feature map): predictions and
 * losses as equss_probe_eval, plus the gradients of the two loss SUMS with respect to the low-resolution table
 *   T[t][c] = <feat[t], probes[c]> + bias[c]:
 *   grad_table (B*h*w, 64) fp32 out, 256-byte aligned, cleared by the call:
 *     [t][c], c < n_linear:          d(sum over labelled pixels of the cross-entropy) / dT[t][c]
 *     [t][32 + c], c < n_cluster:    d(-sum over pixels of the maximal cosine) / dT[t][32 + c]  (arg-max and the pixel's
 *                                    feature norm held constant, as one_hot(argmax) and the detached features are, :101-106)
 *   losses (3) fp32 out: [0], [1] as equss_probe_eval, [2] the number of labelled pixels
 * so that d linear_loss / d weight = (grad_table[:, :n]^T feat) / losses[2], d linear_loss / d bias = column sums / losses[2],
 * d cluster_loss / d (normalised clusters) = (grad_table[:, 32:32+n']^T feat) / (B*H*W).  Atomic float adds: the gradient
 * table agrees between calls to rounding, not bit for bit. */
int equss_probe_train(const float* feat, int64_t ld_feat, const float* probes, const float* bias,
                      const int64_t* label, int64_t* linear_preds, int64_t* cluster_preds, float* losses,
                      float* grad_table, void* workspace, size_t workspace_bytes, int B, int h, int w, int D,
                      int H, int W, int n_linear, int n_cluster, void* stream);

/* Confusion-matrix accumulation of the eval loop: UnSegMetrics.update (model/metric.py:45-58).
 *   confusion (num_pred_classes, num_classes) int64, row-major, updated in place:
 *   confusion[pred, label] += 1 for every i < n with 0 <= label[i] < num_classes and
 *   0 <= preds[i] < num_classes (both int64; the reference applies the same upper bound to both, :49).
 * Integer atomics only, so the result is independent of scheduling. */
int equss_confusion_update(const int64_t* preds, const int64_t* label, int64_t* confusion, int64_t n,
                           int num_classes, int num_pred_classes, void* stream);

/* Code-index table for the reference's offline analyses (the NpyAppendArray dump sketched at
 * train.py:283-329, consumed by cross_entropy.py / edit_distance.py): one int16 row of num_pq indices per
 * label pixel and the int16 label beside it.
 *   z_index  (num_pq, B, h, w) int64: DIONPQGO.forward's stacked index maps (model/dino_pqgo.py:152)
 *   label    (B, H, W) int64
 *   upsample != 0: every label pixel takes the indices of the patch under it (nearest neighbour, source =
 *            floor(dst * in / out) as F.interpolate(mode="nearest")): idx_rows (B*H*W, num_pq), lab_rows (B*H*W)
 *   upsample == 0: the label is sampled at the patch positions instead: (B*h*w, num_pq) and (B*h*w)
 * idx_rows 16-byte aligned.  Values are narrowed to int16 (codebooks up to 32 767 entries, labels from -1). */
int equss_index_rows_i16(const int64_t* z_index, const int64_t* label, int16_t* idx_rows, int16_t* lab_rows,
                         int B, int h, int w, int H, int W, int num_pq, int upsample, void* stream);

/* Codeword usage counts of a quantisation pass: counts[p][k] += |{r : idx[p][r] == k}|, the one-hot sum of
 * model/dino_pqgo.py:668-672 (model/quantizer.py:159-160, :485-487) for all sub-spaces in one launch.
 *   idx (num_pq, rows) int64; counts (num_pq, K) fp32, accumulated in place (clear it first for the counts
 *   of one pass).  Integer counting, so the result does not depend on scheduling.  K <= 8192. */
int equss_code_usage(const int64_t* idx, float* counts, int64_t rows, int num_pq, int K, void* stream);

/* Point sampling of the correspondence loss and its backward (model/loss.py:647-648, `sample()`: F.grid_sample, bilinear, of
 * a (B, C, h, w) map at random coordinates; called by STEGOLoss.forward :701-739) on token-major rows.  The caller turns
 * every sample point into four (row index, weight) taps:
 *   equss_tap_gather:  out[p][:] = sum_k wgt[p][k] * rows[idx[p][k]][:]
 *   equss_tap_scatter: grad_rows[idx[p][k]][:] += wgt[p][k] * g[p][:], grad_rows cleared by the call (float atomics: points
 *                      that share a token are added in no fixed order)
 *   rows / grad_rows (n_rows, C) fp32, idx (points, 4) int64, wgt (points, 4) fp32, out / g (points, C) fp32
 * A tap whose index lies outside [0, n_rows) contributes nothing.  equss_tap_gather: C % 4 == 0, 16-byte aligned rows. */
int equss_tap_gather(const float* rows, int64_t ld_rows, const int64_t* idx, const float* wgt, float* out, int64_t ld_out,
                     int64_t points, int C, int64_t n_rows, void* stream);
int equss_tap_scatter(const float* g, int64_t ld_g, const int64_t* idx, const float* wgt, float* grad_rows, int64_t ld_rows,
                      int64_t points, int C, int64_t n_rows, void* stream);

/* Weight gradient of a 1x1 convolution / nn.Linear in the training pass (the backward of model/dino_pqgo.py:104-112:
 * dW = g^T f, summed over batch * tokens): C[m][n] = sum over r < rows of A[r][m] * B[r][n].
 *   A (rows, M) bf16, B (rows, N) bf16, row-major, leading dimensions multiples of 8, 16-byte aligned
 *   C (M, N) fp32 out, leading dimension ldc, cleared by the call
 * M and N multiples of 128.  bf16 MFMA, fp32 accumulation; the row range is split over workgroups that add their tiles
 * with float atomics, so C agrees between calls to rounding, not bit for bit. */
int equss_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                       int rows, int M, int N, void* stream);

/* Backward of the training-pass quantisation (model/dino_pqgo.py:613-690 under autograd: l2-normalised codes, raw
 * codeword gather, vq = mean_p(book * mse(emb, zn.detach()) + beta * mse(zn, emb.detach())), zq = zn + (emb - zn).detach()),
 * all sub-spaces in one pass over the rows:
 *   z      (rows, num_pq*d) fp32: the codes the forward quantised (before normalisation), leading dimension ld_z
 *   emb    (rows, num_pq*d) fp32: the gathered codewords the forward returned (equss_pq_assign's zq with
 *          gather_normalized = 0 and no straight-through arithmetic, i.e. books[idx])
 *   idx    (num_pq, rows) int64
 *   g_zq   (rows, num_pq*d) fp32 gradient of the quantised output, or NULL for zero
 *   g_vq   one fp32 on the device: gradient of the vq loss
 *   g_z    (rows, num_pq*d) fp32 out or NULL: gradient of z
 *   g_books (num_pq, K, d) fp32 out or NULL: gradient of the codebooks, cleared by the call, accumulated with float
 *          atomics (agrees between calls to rounding, not bit for bit)
 * d in {16, 32, 64, 128}; row buffers 16-byte aligned, leading dimensions multiples of 4. */
int equss_pq_st_backward(const float* z, int64_t ld_z, const float* emb, int64_t ld_emb, const int64_t* idx,
                         const float* g_zq, int64_t ld_g, const float* g_vq, float beta, float book,
                         float* g_z, int64_t ld_gz, float* g_books, int rows, int num_pq, int K, int d, void* stream);

/* Per-codeword sums of the rows assigned to it, for all sub-spaces in one launch: sums[p][k][:] += z[r, p*d:(p+1)*d] over
 * the rows r with idx[p][r] == k -- the one_hot^T @ z of the EMA codebook update (model/quantizer.py:485-491), without the
 * (rows x K) one-hot matrix.
 *   z (rows, num_pq*d) fp32 with leading dimension ld_z; idx (num_pq, rows) int64; sums (num_pq, K, d) fp32, accumulated in
 *   place (clear it first).  Float atomic adds: equal to the matrix product to rounding, not bit for bit. */
int equss_code_sums(const float* z, int64_t ld_z, const int64_t* idx, float* sums, int64_t rows, int num_pq, int K, int d,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EQUSS_HIP_H */
