// The memory-bound pieces of the ViT forward (LayerNorm, patch gather, token assembly) and the
// host-side sequencing of the whole DinoFeaturizer / expansion-head forward on one HIP stream.
#include "equss_common.h"

namespace equss {
int gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo,
              int M, int N, int K, int epilogue, hipStream_t s);
int attention_bf16(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t s);
bool gemm8_resid_ln_supports(int N, int K);
int gemm8_patch_ln(const void* cols, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* pos, float* x,
                   const float* gamma, const float* beta, void* ln, int M, int N, int K, int n, float eps, hipStream_t s);
bool mlp_fused_supports(int C, int hidden);
int block_fused(const void* att, int64_t lda0, const void* wp, const float* bp, const float* gamma2, const float* beta2,
                void* ln_mid, int64_t ld_mid, const void* w1, const float* b1, const void* w2, const float* b2, float* x,
                int64_t ldx, const float* gamma, const float* beta, void* ln, int64_t ld_ln, int M, float eps, hipStream_t s,
                const void* wp_image, const void* w1_image);
size_t tail_weight_image_bytes();
size_t tail_weight_wp_bytes();
size_t tail_weight_w1_bytes();
int pack_tail_weights(const void* const* wp, const void* const* w1, const void* const* w2, int depth, void* out, size_t stride, hipStream_t s);
bool block_ws_enabled();
int block_ws_mode();
size_t rs_weight_image_bytes();
int pack_rs_weights(const void* const* wp, const void* const* w1, const void* const* w2, int depth, void* out, size_t stride, hipStream_t s);
int block_rs(const void* att, int64_t lda0, const void* image, const float* bp, const float* gamma2, const float* beta2,
             const float* b1, const float* b2, float* x, int64_t ldx, const float* gamma, const float* beta, void* ln,
             int64_t ld_ln, int M, float eps, hipStream_t s);
int block_ws(const void* att, int64_t lda0, const void* wp_image, const void* w1_image, const void* w2_image, const float* bp,
             const float* gamma2, const float* beta2, const float* b1, const float* b2, float* x, int64_t ldx, const float* gamma,
             const float* beta, void* ln, int64_t ld_ln, int M, float eps, hipStream_t s);
int mlp_fused(const void* a, int64_t lda, const void* w1, const float* b1, const void* w2, const float* b2, float* x,
              int64_t ldx, const float* gamma, const float* beta, void* ln, int64_t ld_ln, int M, float eps, hipStream_t s);
int gemm8_resid_ln(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* x, int64_t ldx,
                   const float* gamma, const float* beta, void* ln, int64_t ld_ln, int M, int N, int K, float eps,
                   hipStream_t s);
}  // namespace equss

namespace {

using equss::group_sum;

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, the row lives in registers (C / 128 float2 per lane), two-pass
// mean / variance like nn.LayerNorm (biased variance, eps inside the sqrt).
// model/dino/vision_transformer.py:98,:102,:160 with eps = 1e-6 (:271).
// ---------------------------------------------------------------------------------------------
template <int VPL>  // float2 per lane: C = 128 * VPL
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ld_x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, u16* __restrict__ out,
                                                        int64_t ld_out, float* __restrict__ out_f32, int rows,
                                                        int tokens, float eps) {
    constexpr int C = 128 * VPL;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int orow = row;
    if (tokens > 0) {
        const int b = row / tokens, t = row % tokens;
        if (t == 0) return;  // CLS token is dropped (model/dino/dino_featurizer.py:81)
        orow = b * (tokens - 1) + t - 1;
    }
    const float2* xr = reinterpret_cast<const float2*>(x + (size_t)row * ld_x);
    float2 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = xr[i * 64 + lane];
        s += v[i].x + v[i].y;
    }
    const float mean = group_sum<64>(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i].x -= mean; v[i].y -= mean;
        q += v[i].x * v[i].x + v[i].y * v[i].y;
    }
    const float rstd = 1.0f / sqrtf(group_sum<64>(q) * (1.0f / C) + eps);
    const float2* g2 = reinterpret_cast<const float2*>(gamma);
    const float2* b2 = reinterpret_cast<const float2*>(beta);
    uint32_t* o = out ? reinterpret_cast<uint32_t*>(out + (size_t)orow * ld_out) : nullptr;
    float2* of = out_f32 ? reinterpret_cast<float2*>(out_f32 + (size_t)orow * C) : nullptr;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float2 g = g2[i * 64 + lane], bb = b2[i * 64 + lane];
        const float y0 = v[i].x * rstd * g.x + bb.x;
        const float y1 = v[i].y * rstd * g.y + bb.y;
        if (o) o[i * 64 + lane] = equss::pack_bf16(y0, y1);
        if (of) of[i * 64 + lane] = make_float2(y0, y1);
    }
}

// ---------------------------------------------------------------------------------------------
// Patch gather: img (B,3,H,W) fp32 -> rows of 3*P*P bf16 in the conv weight's (c, ph, pw) order,
// so the 8x8/8 convolution of PatchEmbed (vision_transformer.py:128-133) is a plain GEMM.
// One thread moves 8 horizontally adjacent pixels (32 B in, 16 B out).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ img, u16* __restrict__ cols,
                                                           int B, int H, int W, int P, long items) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= items) return;
    const int seg_per_row = P / 8;           // 8-pixel segments per patch row
    const int kk = 3 * P * seg_per_row;      // segments per patch
    const int gw = W / P, gh = H / P;
    const long patch = gid / kk;
    int rem = (int)(gid % kk);
    const int c = rem / (P * seg_per_row); rem %= P * seg_per_row;
    const int ph = rem / seg_per_row, sg = rem % seg_per_row;
    const int b = (int)(patch / ((long)gh * gw));
    const int pr = (int)(patch % ((long)gh * gw));
    const int py = pr / gw, px = pr % gw;
    const float* src = img + (((size_t)b * 3 + c) * H + (size_t)py * P + ph) * W + (size_t)px * P + sg * 8;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(src);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(src + 4);
    uint4 o;
    o.x = equss::pack_bf16(lo.x, lo.y); o.y = equss::pack_bf16(lo.z, lo.w);
    o.z = equss::pack_bf16(hi.x, hi.y); o.w = equss::pack_bf16(hi.z, hi.w);
    *reinterpret_cast<uint4*>(cols + (size_t)patch * (3 * P * P) + (size_t)c * P * P + ph * P + sg * 8) = o;
}

// x[b, 0] = cls + pos[0];  x[b, 1 + t] = patch[b, t] + pos[1 + t]   (vision_transformer.py:205-209)
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const float* __restrict__ patch,
                                                              const float* __restrict__ cls,
                                                              const float* __restrict__ pos, float* __restrict__ x,
                                                              int n, int C, long items) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= items) return;
    const int c4 = C / 4;
    const long row = gid / c4;
    const int col = (int)(gid % c4) * 4;
    const int b = (int)(row / (n + 1)), t = (int)(row % (n + 1));
    const f32x4 p = *reinterpret_cast<const f32x4*>(pos + (size_t)t * C + col);
    f32x4 v;
    if (t == 0) v = *reinterpret_cast<const f32x4*>(cls + col);
    else v = *reinterpret_cast<const f32x4*>(patch + ((size_t)b * n + t - 1) * C + col);
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    *reinterpret_cast<f32x4*>(x + (size_t)row * C + col) = v;
}

// x[b, 0] = cls + pos[0]: the CLS rows next to the patch rows that the patch GEMM's epilogue writes
__global__ __launch_bounds__(256) void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                       float* __restrict__ x, int n, int C, long items) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= items) return;
    const int c4 = C / 4;
    const long b = gid / c4;
    const int col = (int)(gid % c4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(cls + col);
    const f32x4 p = *reinterpret_cast<const f32x4*>(pos + col);
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    *reinterpret_cast<f32x4*>(x + (size_t)b * (n + 1) * C + col) = v;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct PatchWs { u16* cols; float* tmp; size_t bytes; };
PatchWs carve_patch(void* ws, int B, int H, int W, int P, int C) {
    const size_t n = (size_t)B * (H / P) * (W / P);
    PatchWs w;
    char* base = reinterpret_cast<char*>(ws);
    size_t off = 0;
    w.cols = reinterpret_cast<u16*>(base + off); off += align256(n * 3 * P * P * 2);
    w.tmp = reinterpret_cast<float*>(base + off); off += align256(n * C * 4);
    w.bytes = off;
    return w;
}

// bytes between the weight images of two blocks (either block-tail kernel's format fits)
size_t tail_image_stride() {
    const size_t a = equss::tail_weight_image_bytes(), b = equss::rs_weight_image_bytes();
    return align256(a > b ? a : b);
}

struct VitWs { float* x; u16* ln; u16* qkv; u16* att; u16* hid; void* patch; size_t patch_bytes; char* wimg; size_t bytes; };
// wimg: ring-stage images of every block's projection + fc1 weights for the fused block-tail kernel (C = 384, up to 16
// blocks: 1.4 MiB each, rewritten by every forward: the weights may have changed between two calls)
VitWs carve_vit(void* ws, int C, int B, int H, int W, int P, int depth) {
    const size_t n = (size_t)(H / P) * (W / P);
    const size_t rows = (size_t)B * (n + 1);
    VitWs w;
    char* base = reinterpret_cast<char*>(ws);
    size_t off = 0;
    w.x = reinterpret_cast<float*>(base + off); off += align256(rows * C * 4);
    w.ln = reinterpret_cast<u16*>(base + off); off += align256(rows * C * 2);
    w.qkv = reinterpret_cast<u16*>(base + off); off += align256(rows * 3 * C * 2);
    w.att = reinterpret_cast<u16*>(base + off); off += align256(rows * C * 2);
    w.hid = reinterpret_cast<u16*>(base + off); off += align256(rows * 4 * C * 2);
    w.patch = base + off;
    w.patch_bytes = carve_patch(nullptr, B, H, W, P, C).bytes;
    off += w.patch_bytes;
    w.wimg = nullptr;
    if (equss::mlp_fused_supports(C, 4 * C) && depth >= 1 && depth <= 16) {
        w.wimg = base + off;
        off += tail_image_stride() * depth;
    }
    w.bytes = off;
    return w;
}

int check_image(int B, int H, int W, int P, int C) {
    if (B <= 0 || H <= 0 || W <= 0) return equss::fail(EQUSS_E_SHAPE, "image batch %d x %d x %d must be positive", B, H, W);
    if (P != 8 && P != 16) return equss::fail(EQUSS_E_UNSUPPORTED, "patch size %d (8 or 16 supported)", P);
    if (H % P != 0 || W % P != 0)  // model/dino/dino_featurizer.py:70
        return equss::fail(EQUSS_E_SHAPE, "image %d x %d is not divisible by the patch size %d", H, W, P);
    if (C % 128 != 0 || C > 1024) return equss::fail(EQUSS_E_UNSUPPORTED, "embed dim %d (multiples of 128 up to 1024)", C);
    return EQUSS_OK;
}

int layernorm(const float* x, int64_t ld_x, const float* gamma, const float* beta, void* out, int64_t ld_out,
              float* out_f32, int rows, int C, int tokens, float eps, hipStream_t s) {
    const int grid = (rows + 3) / 4;
    u16* o = reinterpret_cast<u16*>(out);
#define EQUSS_LN_CASE(V)                                                                                     \
    case V:                                                                                                  \
        hipLaunchKernelGGL(layernorm_kernel<V>, dim3(grid), dim3(256), 0, s, x, ld_x, gamma, beta, o, ld_out, \
                           out_f32, rows, tokens, eps);                                                      \
        break;
    switch (C / 128) {
        EQUSS_LN_CASE(1) EQUSS_LN_CASE(2) EQUSS_LN_CASE(3) EQUSS_LN_CASE(4)
        EQUSS_LN_CASE(5) EQUSS_LN_CASE(6) EQUSS_LN_CASE(7) EQUSS_LN_CASE(8)
        default: return equss::fail(EQUSS_E_UNSUPPORTED, "layernorm: C=%d", C);
    }
#undef EQUSS_LN_CASE
    return equss::check_launch("layernorm_kernel");
}

// LayerNorm of rows of width 384 with exactly the arithmetic of the fused residual + LayerNorm GEMM
// epilogue (gemm8_bf16.hip): the same element-to-lane map (wave = 96-column group, lane = (row, 4-column
// quad)), the same order of partial sums and the same fma placement, so a row normalised here has the
// same bf16 bits as one normalised there.  Used for the trailing rows that leave the fused kernel.
__global__ __launch_bounds__(256) void layernorm_rowtile_kernel(const float* __restrict__ x, int64_t ld_x,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               u16* __restrict__ out, int64_t ld_out, int rows,
                                                               float eps) {
    __shared__ float2 red[16][4];
    const int lane = threadIdx.x & 63, wn = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int m = blockIdx.x * 16 + fr;
    f32x4 v[6];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) {
        const int n = wn * 96 + nt * 16 + fq * 4;
        v[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < rows) v[nt] = *reinterpret_cast<const f32x4*>(x + (size_t)m * ld_x + n);
        s1 += (v[nt].x + v[nt].y) + (v[nt].z + v[nt].w);
        s2 += fmaf(v[nt].x, v[nt].x, v[nt].y * v[nt].y) + fmaf(v[nt].z, v[nt].z, v[nt].w * v[nt].w);
    }
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    if (fq == 0) red[fr][wn] = make_float2(s1, s2);
    __syncthreads();
    const float2 p0 = red[fr][0], p1 = red[fr][1], p2 = red[fr][2], p3 = red[fr][3];
    const float mean = ((p0.x + p1.x) + (p2.x + p3.x)) * (1.0f / 384);
    const float ex2 = ((p0.y + p1.y) + (p2.y + p3.y)) * (1.0f / 384);
    const float rstd = 1.0f / sqrtf(fmaxf(fmaf(-mean, mean, ex2), 0.f) + eps);
    if (m >= rows) return;
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) {
        const int n = wn * 96 + nt * 16 + fq * 4;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + n);
        const f32x4 be = *reinterpret_cast<const f32x4*>(beta + n);
        uint2 pk;
        pk.x = equss::pack_bf16(fmaf((v[nt].x - mean) * rstd, ga.x, be.x), fmaf((v[nt].y - mean) * rstd, ga.y, be.y));
        pk.y = equss::pack_bf16(fmaf((v[nt].z - mean) * rstd, ga.z, be.z), fmaf((v[nt].w - mean) * rstd, ga.w, be.w));
        *reinterpret_cast<uint2*>(out + (size_t)m * ld_out + n) = pk;
    }
}

// x += A W^T + bias, ln = LayerNorm(x), with the workgroup count balanced against the chip.
// The fused kernel walks 128-row tiles with one persistent workgroup per CU, so a tile count just
// above a multiple of 256 leaves most CUs idle for a whole extra round (64 x 1601 tokens = 801 tiles
// = 3 rounds + 33 tiles).  Those trailing rows go through the 128 x 128-tile residual GEMM (three times
// as many, smaller tiles, two workgroups per CU) and a LayerNorm launch of their own instead; both
// repeat the fused kernel's fp32 operation order, so which route a row takes does not change its bits.
int resid_ln_balanced(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* x, int64_t ldx,
                      const float* gamma, const float* beta, void* ln, int64_t ld_ln, int M, int N, int K, float eps,
                      hipStream_t s) {
    const int tiles = (M + 127) / 128, rounds = tiles / 256, rem = tiles % 256;
    if (rounds < 1 || rem == 0 || rem > 128)
        return equss::gemm8_resid_ln(A, lda, W, ldw, bias, x, ldx, gamma, beta, ln, ld_ln, M, N, K, eps, s);
    const int m_main = rounds * 256 * 128, m_tail = M - m_main;
    int rc = equss::gemm8_resid_ln(A, lda, W, ldw, bias, x, ldx, gamma, beta, ln, ld_ln, m_main, N, K, eps, s);
    if (rc) return rc;
    const u16* a_tail = reinterpret_cast<const u16*>(A) + (size_t)m_main * lda;
    float* x_tail = x + (size_t)m_main * ldx;
    rc = equss::gemm_bf16(a_tail, lda, W, ldw, bias, x_tail, ldx, m_tail, N, K, EQUSS_EPI_RESID_F32, s);
    if (rc || !ln) return rc;
    hipLaunchKernelGGL(layernorm_rowtile_kernel, dim3((m_tail + 15) / 16), dim3(256), 0, s, x_tail, ldx, gamma, beta,
                       reinterpret_cast<u16*>(ln) + (size_t)m_main * ld_ln, ld_ln, m_tail, eps);
    return equss::check_launch("layernorm_rowtile_kernel");
}

// x += fc2(gelu(fc1(a))), ln = LayerNorm(x) for C = 384: the fused MLP kernel on as many whole rounds of
// 128-row tiles as the chip's 256 CUs take, and the trailing rows (a partly filled extra round would
// cost a full tile time, ~100 us) through the two-GEMM route with small tiles.  Both routes give a row
// the same bits.  `hid` is scratch for the trailing rows' hidden activation (m_tail x 1536 bf16).
int g_mlp_tail_max = 128;   // trailing tiles (of 256) up to which the split pays
int mlp_balanced(const void* a, int64_t lda, const void* w1, const float* b1, const void* w2, const float* b2, float* x,
                 int64_t ldx, const float* gamma, const float* beta, void* ln, int64_t ld_ln, void* hid, int M, float eps,
                 hipStream_t s) {
    const int C = 384, H = 1536;
    const int tiles = (M + 127) / 128, rounds = tiles / 256, rem = tiles % 256;
    if (rounds < 1 || rem == 0 || rem > g_mlp_tail_max || !hid)
        return equss::mlp_fused(a, lda, w1, b1, w2, b2, x, ldx, gamma, beta, ln, ld_ln, M, eps, s);
    const int m_main = rounds * 256 * 128, m_tail = M - m_main;
    int rc = equss::mlp_fused(a, lda, w1, b1, w2, b2, x, ldx, gamma, beta, ln, ld_ln, m_main, eps, s);
    if (rc) return rc;
    const u16* a_tail = reinterpret_cast<const u16*>(a) + (size_t)m_main * lda;
    float* x_tail = x + (size_t)m_main * ldx;
    rc = equss::gemm_bf16(a_tail, lda, w1, C, b1, hid, H, m_tail, H, C, EQUSS_EPI_GELU_BF16, s);
    if (rc) return rc;
    rc = equss::gemm_bf16(hid, H, w2, H, b2, x_tail, ldx, m_tail, C, H, EQUSS_EPI_RESID_F32, s);
    if (rc || !ln) return rc;
    hipLaunchKernelGGL(layernorm_rowtile_kernel, dim3((m_tail + 15) / 16), dim3(256), 0, s, x_tail, ldx, gamma, beta,
                       reinterpret_cast<u16*>(ln) + (size_t)m_main * ld_ln, ld_ln, m_tail, eps);
    return equss::check_launch("layernorm_rowtile_kernel");
}

// Everything after attention for C = 384: the fused block kernel on the whole rounds of tiles, the trailing
// rows through the small-tile routes of the projection and of the MLP (same bits either way).
int block_balanced(const void* att, const void* wp, const float* bp, const float* gamma2, const float* beta2, void* ln_mid,
                   const void* w1, const float* b1, const void* w2, const float* b2, float* x, const float* gamma,
                   const float* beta, void* ln, void* hid, int M, float eps, hipStream_t s, const char* wimg) {
    const int C = 384, H = 1536;
    const bool ws = wimg && equss::block_ws_enabled();   // the producer / consumer kernel works on 64-row tiles
    // (its 64-row tiles make a partly filled last round cheaper than the small-tile launches for the trailing rows: 10.35
    // against 10.62 ms per 64-image forward on one stream, 9.8 against 10.3 ms on two: it takes all rows)
    const int tiles = (M + 127) / 128, rounds = tiles / 256, rem = tiles % 256;
    const bool split = !ws && rounds >= 1 && rem != 0 && rem <= g_mlp_tail_max && hid;
    const int m_main = split ? rounds * 256 * 128 : M, m_tail = M - m_main;
    int rc;
    {
        equss::ProfScope ps(EQUSS_K_BLOCK_TAIL, s);   // the fused launch alone; the trailing rows count under the GEMM classes
        if (ws && equss::block_ws_mode() == 3)
            rc = equss::block_rs(att, C, wimg, bp, gamma2, beta2, b1, b2, x, C, gamma, beta, ln, C, m_main, eps, s);
        else if (ws)
            rc = equss::block_ws(att, C, wimg, wimg + equss::tail_weight_wp_bytes(),
                                 wimg + equss::tail_weight_wp_bytes() + equss::tail_weight_w1_bytes(), bp, gamma2, beta2, b1, b2, x, C,
                                 gamma, beta, ln, C, m_main, eps, s);
        else
            rc = equss::block_fused(att, C, wp, bp, gamma2, beta2, ln_mid, C, w1, b1, w2, b2, x, C, gamma, beta, ln, C, m_main, eps, s,
                                    wimg, wimg ? wimg + equss::tail_weight_wp_bytes() : nullptr);
    }
    if (rc || m_tail == 0) return rc;
    const u16* att_tail = reinterpret_cast<const u16*>(att) + (size_t)m_main * C;
    u16* mid_tail = reinterpret_cast<u16*>(ln_mid) + (size_t)m_main * C;
    float* x_tail = x + (size_t)m_main * C;
    {
        equss::ProfScope ps(EQUSS_K_GEMM_PROJ, s);
        rc = equss::gemm_bf16(att_tail, C, wp, C, bp, x_tail, C, m_tail, C, C, EQUSS_EPI_RESID_F32, s);
        if (rc) return rc;
        hipLaunchKernelGGL(layernorm_rowtile_kernel, dim3((m_tail + 15) / 16), dim3(256), 0, s, x_tail, C, gamma2, beta2, mid_tail,
                           C, m_tail, eps);
        rc = equss::check_launch("layernorm_rowtile_kernel");
        if (rc) return rc;
    }
    {
        equss::ProfScope ps(EQUSS_K_GEMM_FC1, s);
        rc = equss::gemm_bf16(mid_tail, C, w1, C, b1, hid, H, m_tail, H, C, EQUSS_EPI_GELU_BF16, s);
        if (rc) return rc;
    }
    equss::ProfScope ps(EQUSS_K_GEMM_FC2, s);
    rc = equss::gemm_bf16(hid, H, w2, H, b2, x_tail, C, m_tail, C, H, EQUSS_EPI_RESID_F32, s);
    if (rc || !ln) return rc;
    hipLaunchKernelGGL(layernorm_rowtile_kernel, dim3((m_tail + 15) / 16), dim3(256), 0, s, x_tail, C, gamma, beta,
                       reinterpret_cast<u16*>(ln) + (size_t)m_main * C, C, m_tail, eps);
    return equss::check_launch("layernorm_rowtile_kernel");
}

// ln_g / ln_b / ln: optional LayerNorm of the assembled tokens (block 0's norm1), bf16, same row layout as x.
int patch_tokens(const float* img, const void* w_patch, const float* b_patch, const float* cls, const float* pos,
                 float* x, void* ws, int B, int H, int W, int P, int C, hipStream_t s, const float* ln_g = nullptr,
                 const float* ln_b = nullptr, void* ln = nullptr, bool* ln_done = nullptr) {
    PatchWs w = carve_patch(ws, B, H, W, P, C);
    const int n = (H / P) * (W / P);
    const long items = (long)B * n * 3 * P * (P / 8);
    hipLaunchKernelGGL(patch_gather_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, img, w.cols, B, H, W,
                       P, items);
    int rc = equss::check_launch("patch_gather_kernel");
    if (rc) return rc;
    if (ln_done) *ln_done = false;
    if (equss::gemm8_resid_ln_supports(C, 3 * P * P)) {
        // C == 384: the GEMM's epilogue adds the position embedding, writes the token rows and their LayerNorm
        rc = equss::gemm8_patch_ln(w.cols, 3 * P * P, w_patch, 3 * P * P, b_patch, pos, x, ln_g, ln_b, ln, B * n, C, 3 * P * P, n,
                                   1e-6f, s);
        if (rc) return rc;
        const long items3 = (long)B * (C / 4);
        hipLaunchKernelGGL(cls_rows_kernel, dim3((unsigned)((items3 + 255) / 256)), dim3(256), 0, s, cls, pos, x, n, C, items3);
        rc = equss::check_launch("cls_rows_kernel");
        if (rc || !ln) return rc;
        if (ln_done) *ln_done = true;
        return layernorm(x, (int64_t)(n + 1) * C, ln_g, ln_b, ln, (int64_t)(n + 1) * C, nullptr, B, C, 0, 1e-6f, s);
    }
    rc = equss::gemm_bf16(w.cols, 3 * P * P, w_patch, 3 * P * P, b_patch, w.tmp, C, B * n, C, 3 * P * P, EQUSS_EPI_F32, s);
    if (rc) return rc;
    const long items2 = (long)B * (n + 1) * (C / 4);
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3((unsigned)((items2 + 255) / 256)), dim3(256), 0, s, w.tmp, cls, pos, x,
                       n, C, items2);
    return equss::check_launch("assemble_tokens_kernel");
}

}  // namespace

namespace equss {
// fp32-only entry points for the fp32 parity forward (parity_f32.hip)
int layernorm_f32(const float* x, int64_t ld_x, const float* gamma, const float* beta, float* out, int rows, int C,
                  int tokens, float eps, hipStream_t s) {
    return layernorm(x, ld_x, gamma, beta, nullptr, C, out, rows, C, tokens, eps, s);
}
int assemble_tokens_f32(const float* patch, const float* cls, const float* pos, float* x, int B, int n, int C, hipStream_t s) {
    const long items = (long)B * (n + 1) * (C / 4);
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, patch, cls, pos, x, n, C, items);
    return check_launch("assemble_tokens_kernel");
}
}  // namespace equss

extern "C" int equss_layernorm_bf16(const float* x, int64_t ld_x, const float* gamma, const float* beta,
                                    void* out_bf16, int64_t ld_out, float* out_f32, int rows, int C,
                                    int tokens_per_image, float eps, void* stream) {
    if (!x || !gamma || !beta || !out_bf16) return equss::fail(EQUSS_E_NULL, "layernorm: null pointer argument");
    if (rows <= 0 || C <= 0) return equss::fail(EQUSS_E_SHAPE, "layernorm: rows=%d C=%d must be positive", rows, C);
    if (C % 128 != 0 || C > 1024) return equss::fail(EQUSS_E_UNSUPPORTED, "layernorm: C=%d (multiples of 128 up to 1024)", C);
    if (tokens_per_image > 0 && rows % tokens_per_image != 0)
        return equss::fail(EQUSS_E_SHAPE, "layernorm: rows=%d is not a multiple of tokens_per_image=%d", rows, tokens_per_image);
    if (ld_x < C || ld_out < C || (ld_x % 2) || (ld_out % 2) || ((uintptr_t)x % 8) || ((uintptr_t)out_bf16 % 4) ||
        ((uintptr_t)gamma % 8) || ((uintptr_t)beta % 8) || (out_f32 && ((uintptr_t)out_f32 % 8)))
        return equss::fail(EQUSS_E_ALIGN, "layernorm: pointers must be 8-byte aligned and leading dimensions even");
    return layernorm(x, ld_x, gamma, beta, out_bf16, ld_out, out_f32, rows, C, tokens_per_image, eps,
                     reinterpret_cast<hipStream_t>(stream));
}

extern "C" size_t equss_patch_workspace_bytes(int B, int H, int W, int P, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || P <= 0 || C <= 0) return 0;
    return carve_patch(nullptr, B, H, W, P, C).bytes;
}

extern "C" int equss_patch_tokens(const float* img, const void* w_patch, const float* b_patch, const float* cls,
                                  const float* pos, float* x, void* workspace, size_t workspace_bytes, int B, int H,
                                  int W, int P, int C, void* stream) {
    if (!img || !w_patch || !b_patch || !cls || !pos || !x || !workspace)
        return equss::fail(EQUSS_E_NULL, "patch_tokens: null pointer argument");
    int rc = check_image(B, H, W, P, C);
    if (rc) return rc;
    if (((uintptr_t)img % 16) || ((uintptr_t)x % 16) || ((uintptr_t)pos % 16) || ((uintptr_t)cls % 16) ||
        ((uintptr_t)workspace % 256))
        return equss::fail(EQUSS_E_ALIGN, "patch_tokens: img/x/pos/cls must be 16-byte aligned, workspace 256-byte aligned");
    if (workspace_bytes < equss_patch_workspace_bytes(B, H, W, P, C))
        return equss::fail(EQUSS_E_WORKSPACE, "patch_tokens: workspace too small");
    return patch_tokens(img, w_patch, b_patch, cls, pos, x, workspace, B, H, W, P, C, reinterpret_cast<hipStream_t>(stream));
}

int g_fuse_min_tiles = 384;
int g_tail_images = 1;   // developer switch: ring-stage images of Wp / W1 for the fused block-tail kernel
int g_mlp_fused = 2;   // developer switch: 2 = proj + MLP in one kernel, 1 = MLP in one kernel, 0 = one launch per GEMM
extern "C" void equss_dev_mlp_fused(int on) { g_mlp_fused = on; }
extern "C" void equss_dev_mlp_tail_max(int n) { g_mlp_tail_max = n; }
extern "C" void equss_dev_fuse_min_tiles(int n) { g_fuse_min_tiles = n; }
extern "C" void equss_dev_tail_images(int on) { g_tail_images = on; }

extern "C" size_t equss_vit_workspace_bytes(const EqussVit* vit, int B, int H, int W) {
    if (!vit || B <= 0 || H <= 0 || W <= 0 || vit->patch <= 0) return 0;
    return carve_vit(nullptr, vit->embed_dim, B, H, W, vit->patch, vit->depth).bytes;
}

// prepare_tokens, then per block: x += proj(attn(LN1 x)); x += fc2(gelu(fc1(LN2 x))); final norm on
// the last block's output with the CLS row dropped (vision_transformer.py:106-114,:227-239;
// dino_featurizer.py:76-81).  ~8 launches per block, all on `stream`, no host synchronisation.
// Optional taps: inter (n_last, B*N, C) fp32 = norm(x) after each of the last n_last blocks with the CLS
// row kept (what get_intermediate_feat returns, :233-239); x_taps (depth, B*N, C) fp32 = the residual
// stream after every block.
namespace {
int vit_forward_impl(const EqussVit* vit, const float* img, const float* pos, void* feat_bf16, int64_t ld_feat,
                     float* feat_f32, void* workspace, int B, int H, int W, hipStream_t s, float* inter, int n_last,
                     float* x_taps) {
    const int C = vit->embed_dim, P = vit->patch, heads = vit->heads;
    int rc;
    VitWs w = carve_vit(workspace, C, B, H, W, P, vit->depth);
    const int n = (H / P) * (W / P), N = n + 1, rows = B * N;
    bool ln0_done = false;   // block 0's norm1 came out of the patch embedding
    {
        equss::ProfScope ps(EQUSS_K_PATCH, s);
        rc = patch_tokens(img, vit->patch_w, vit->patch_b, vit->cls, pos, w.x, w.patch, B, H, W, P, C, s,
                          vit->depth > 0 ? vit->blocks[0].ln1_g : nullptr, vit->depth > 0 ? vit->blocks[0].ln1_b : nullptr,
                          vit->depth > 0 ? w.ln : nullptr, &ln0_done);
    }
    if (rc) return rc;
    // The fused block-tail kernel reads the projection / fc1 weights as ring-stage images: written here, once per forward
    // for all blocks (one launch, ~17 MB), when that kernel is going to run.
    bool images = false;
    const char* wimg = w.wimg;
    if (g_tail_images && w.wimg && g_mlp_fused == 2 && (rows + 127) / 128 >= g_fuse_min_tiles && equss::gemm8_resid_ln_supports(C, C) &&
        equss::mlp_fused_supports(C, 4 * C)) {
        if (vit->tail_images && equss::block_ws_mode() == 3) {
            wimg = reinterpret_cast<const char*>(vit->tail_images);   // written once for these weights (equss_vit_pack_tail_images)
        } else {
            const void* wp[16]; const void* w1[16]; const void* w2[16];
            for (int i = 0; i < vit->depth; ++i) { wp[i] = vit->blocks[i].proj_w; w1[i] = vit->blocks[i].fc1_w; w2[i] = vit->blocks[i].fc2_w; }
            equss::ProfScope ps(EQUSS_K_PACK_WEIGHTS, s);
            rc = equss::block_ws_mode() == 3 ? equss::pack_rs_weights(wp, w1, w2, vit->depth, w.wimg, tail_image_stride(), s)
                                             : equss::pack_tail_weights(wp, w1, w2, vit->depth, w.wimg, tail_image_stride(), s);
            if (rc) return rc;
        }
        images = true;
    }
    const float scale = 0.125f;  // head_dim ** -0.5 with head_dim = 64 (vision_transformer.py:71-72)
#define EQUSS_STEP(cls, call)                 \
    {                                         \
        equss::ProfScope ps(cls, s);          \
        rc = (call);                          \
    }                                         \
    if (rc) return rc;
    auto run_block = [&](int i) -> int {
        const EqussVitBlock& k = vit->blocks[i];
        // With C == 384 the residual GEMMs also emit the LayerNorm their consumer needs (norm2 of this
        // block after proj, norm1 of the next block after fc2), so only block 0's norm1 and the final
        // norm run as standalone kernels.
        const bool fuse = equss::gemm8_resid_ln_supports(C, C);
        const bool last = i + 1 == vit->depth;
        if (!fuse || (i == 0 && !ln0_done)) {
            EQUSS_STEP(EQUSS_K_LAYERNORM, layernorm(w.x, C, k.ln1_g, k.ln1_b, w.ln, C, nullptr, rows, C, 0, 1e-6f, s))
        }
        EQUSS_STEP(EQUSS_K_GEMM_QKV, equss::gemm_bf16(w.ln, C, k.qkv_w, C, k.qkv_b, w.qkv, 3 * C, rows, 3 * C, C, EQUSS_EPI_BF16, s))
        EQUSS_STEP(EQUSS_K_ATTENTION, equss::attention_bf16(w.qkv, w.att, B, N, heads, scale, s))
        // (below ~1.5 rounds of 128-row tiles on the 256 CUs the one-launch-per-GEMM route spreads the work better)
        const bool big = (rows + 127) / 128 >= g_fuse_min_tiles;
        if (fuse && big && g_mlp_fused == 2 && equss::mlp_fused_supports(C, 4 * C)) {
            const EqussVitBlock* nb = last ? nullptr : &vit->blocks[i + 1];
            return block_balanced(w.att, k.proj_w, k.proj_b, k.ln2_g, k.ln2_b, w.ln, k.fc1_w, k.fc1_b, k.fc2_w, k.fc2_b, w.x, nb ? nb->ln1_g : nullptr, nb ? nb->ln1_b : nullptr, nb ? w.ln : nullptr, w.hid, rows, 1e-6f, s,
                                  images ? wimg + (size_t)i * tail_image_stride() : nullptr);
        }
        if (fuse) {
            EQUSS_STEP(EQUSS_K_GEMM_PROJ, resid_ln_balanced(w.att, C, k.proj_w, C, k.proj_b, w.x, C, k.ln2_g, k.ln2_b, w.ln, C, rows, C, C, 1e-6f, s))
        } else {
            EQUSS_STEP(EQUSS_K_GEMM_PROJ, equss::gemm_bf16(w.att, C, k.proj_w, C, k.proj_b, w.x, C, rows, C, C, EQUSS_EPI_RESID_F32, s))
            EQUSS_STEP(EQUSS_K_LAYERNORM, layernorm(w.x, C, k.ln2_g, k.ln2_b, w.ln, C, nullptr, rows, C, 0, 1e-6f, s))
        }
        if (fuse && big && g_mlp_fused && equss::mlp_fused_supports(C, 4 * C)) {
            const EqussVitBlock* nb = last ? nullptr : &vit->blocks[i + 1];
            EQUSS_STEP(EQUSS_K_MLP_FUSED, mlp_balanced(w.ln, C, k.fc1_w, k.fc1_b, k.fc2_w, k.fc2_b, w.x, C, nb ? nb->ln1_g : nullptr, nb ? nb->ln1_b : nullptr, nb ? w.ln : nullptr, C, w.hid, rows, 1e-6f, s))
            return EQUSS_OK;
        }
        EQUSS_STEP(EQUSS_K_GEMM_FC1, equss::gemm_bf16(w.ln, C, k.fc1_w, C, k.fc1_b, w.hid, 4 * C, rows, 4 * C, C, EQUSS_EPI_GELU_BF16, s))
        if (fuse) {
            const EqussVitBlock* nb = last ? nullptr : &vit->blocks[i + 1];
            EQUSS_STEP(EQUSS_K_GEMM_FC2, resid_ln_balanced(w.hid, 4 * C, k.fc2_w, 4 * C, k.fc2_b, w.x, C, nb ? nb->ln1_g : nullptr, nb ? nb->ln1_b : nullptr, nb ? w.ln : nullptr, C, rows, C, 4 * C, 1e-6f, s))
        } else {
            EQUSS_STEP(EQUSS_K_GEMM_FC2, equss::gemm_bf16(w.hid, 4 * C, k.fc2_w, 4 * C, k.fc2_b, w.x, C, rows, C, 4 * C, EQUSS_EPI_RESID_F32, s))
        }
        return EQUSS_OK;
    };
    const size_t tap = (size_t)rows * C;
    for (int i = 0; i < vit->depth; ++i) {
        rc = run_block(i);
        if (rc) return rc;
        if (x_taps) {
            hipError_t e = hipMemcpyAsync(x_taps + (size_t)i * tap, w.x, tap * sizeof(float), hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) { equss::fail(0, "vit_forward: copy of the residual tap: %s", hipGetErrorString(e)); return -(int)e; }
        }
        const int from_end = vit->depth - i;   // the reference keeps norm(x) of block i when depth - i <= n (:235-236)
        if (inter && from_end <= n_last) {
            EQUSS_STEP(EQUSS_K_LAYERNORM, layernorm(w.x, C, vit->norm_g, vit->norm_b, nullptr, C, inter + (size_t)(n_last - from_end) * tap, rows, C, 0, 1e-6f, s))
        }
    }
    if (feat_bf16 || feat_f32) {
        EQUSS_STEP(EQUSS_K_LAYERNORM, layernorm(w.x, C, vit->norm_g, vit->norm_b, feat_bf16, ld_feat, feat_f32, rows, C, N, 1e-6f, s))
    }
    return EQUSS_OK;
}

int check_vit(const EqussVit* vit, const float* img, const float* pos, void* workspace, size_t workspace_bytes, int B,
              int H, int W, const char* who) {
    if (!vit || !img || !pos || !workspace || !vit->blocks) return equss::fail(EQUSS_E_NULL, "%s: null pointer argument", who);
    const int C = vit->embed_dim;
    int rc = check_image(B, H, W, vit->patch, C);
    if (rc) return rc;
    if (vit->heads * 64 != C) return equss::fail(EQUSS_E_UNSUPPORTED, "%s: head dim must be 64 (C=%d, heads=%d)", who, C, vit->heads);
    if ((uintptr_t)workspace % 256) return equss::fail(EQUSS_E_ALIGN, "%s: workspace must be 256-byte aligned", who);
    if (workspace_bytes < equss_vit_workspace_bytes(vit, B, H, W))
        return equss::fail(EQUSS_E_WORKSPACE, "%s: workspace of %zu bytes is smaller than the %zu required", who,
                           workspace_bytes, equss_vit_workspace_bytes(vit, B, H, W));
    return EQUSS_OK;
}
}  // namespace

extern "C" size_t equss_vit_tail_image_bytes(const EqussVit* vit) {
    if (!vit || vit->depth < 1 || vit->depth > 16 || !equss::mlp_fused_supports(vit->embed_dim, 4 * vit->embed_dim)) return 0;
    return tail_image_stride() * (size_t)vit->depth;
}

extern "C" int equss_vit_pack_tail_images(const EqussVit* vit, void* images, size_t images_bytes, void* stream) {
    if (!vit || !vit->blocks || !images) return equss::fail(EQUSS_E_NULL, "vit_pack_tail_images: null pointer argument");
    const size_t need = equss_vit_tail_image_bytes(vit);
    if (need == 0) return equss::fail(EQUSS_E_UNSUPPORTED, "vit_pack_tail_images: embed dim %d has no fused block-tail kernel", vit->embed_dim);
    if (images_bytes < need) return equss::fail(EQUSS_E_WORKSPACE, "vit_pack_tail_images: %zu bytes given, %zu needed", images_bytes, need);
    if ((uintptr_t)images % 256) return equss::fail(EQUSS_E_ALIGN, "vit_pack_tail_images: images must be 256-byte aligned");
    const void* wp[16]; const void* w1[16]; const void* w2[16];
    for (int i = 0; i < vit->depth; ++i) { wp[i] = vit->blocks[i].proj_w; w1[i] = vit->blocks[i].fc1_w; w2[i] = vit->blocks[i].fc2_w; }
    return equss::pack_rs_weights(wp, w1, w2, vit->depth, images, tail_image_stride(), reinterpret_cast<hipStream_t>(stream));
}

extern "C" int equss_vit_forward(const EqussVit* vit, const float* img, const float* pos, void* feat_bf16,
                                 int64_t ld_feat, float* feat_f32, void* workspace, size_t workspace_bytes, int B,
                                 int H, int W, void* stream) {
    if (!feat_bf16) return equss::fail(EQUSS_E_NULL, "vit_forward: null pointer argument");
    int rc = check_vit(vit, img, pos, workspace, workspace_bytes, B, H, W, "vit_forward");
    if (rc) return rc;
    if (ld_feat < vit->embed_dim || (ld_feat % 8) || ((uintptr_t)feat_bf16 % 16) || (feat_f32 && ((uintptr_t)feat_f32 % 8)))
        return equss::fail(EQUSS_E_ALIGN, "vit_forward: feat must be 16-byte aligned with ld_feat a multiple of 8");
    return vit_forward_impl(vit, img, pos, feat_bf16, ld_feat, feat_f32, workspace, B, H, W,
                            reinterpret_cast<hipStream_t>(stream), nullptr, 0, nullptr);
}

extern "C" int equss_vit_intermediate_feat(const EqussVit* vit, const float* img, const float* pos, float* feats,
                                           int n_last, float* x_taps, void* workspace, size_t workspace_bytes, int B,
                                           int H, int W, void* stream) {
    if (!feats) return equss::fail(EQUSS_E_NULL, "vit_intermediate_feat: null pointer argument");
    int rc = check_vit(vit, img, pos, workspace, workspace_bytes, B, H, W, "vit_intermediate_feat");
    if (rc) return rc;
    if (n_last < 1 || n_last > vit->depth)
        return equss::fail(EQUSS_E_SHAPE, "vit_intermediate_feat: n_last=%d must lie in [1, depth=%d]", n_last, vit->depth);
    if (((uintptr_t)feats % 8) || (x_taps && ((uintptr_t)x_taps % 8)))
        return equss::fail(EQUSS_E_ALIGN, "vit_intermediate_feat: outputs must be 8-byte aligned");
    return vit_forward_impl(vit, img, pos, nullptr, vit->embed_dim, nullptr, workspace, B, H, W,
                            reinterpret_cast<hipStream_t>(stream), feats, n_last, x_taps);
}

extern "C" int equss_gemm_resid_ln_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                                        float* x, int64_t ldx, const float* gamma, const float* beta, void* ln_out,
                                        int64_t ld_ln, int M, int N, int K, float eps, void* stream) {
    if (!A || !W || !x || (ln_out && (!gamma || !beta))) return equss::fail(EQUSS_E_NULL, "gemm_resid_ln: null pointer argument");
    if (M <= 0 || K <= 0) return equss::fail(EQUSS_E_SHAPE, "gemm_resid_ln: M=%d K=%d must be positive", M, K);
    if (!equss::gemm8_resid_ln_supports(N, K))
        return equss::fail(EQUSS_E_UNSUPPORTED, "gemm_resid_ln: N=%d K=%d (N must be 384, K a multiple of 64)", N, K);
    if (lda < K || ldw < K || ldx < N || (ln_out && ld_ln < N) || (lda % 8) || (ldw % 8) || (ldx % 4) || (ld_ln % 8) ||
        ((uintptr_t)A % 16) || ((uintptr_t)W % 16) || ((uintptr_t)x % 16) || ((uintptr_t)ln_out % 16))
        return equss::fail(EQUSS_E_ALIGN, "gemm_resid_ln: operands must be 16-byte aligned with padded leading dimensions");
    return resid_ln_balanced(A, lda, W, ldw, bias, x, ldx, gamma, beta, ln_out, ld_ln, M, N, K, eps,
                             reinterpret_cast<hipStream_t>(stream));
}

// code = [f | relu(f W2a^T + b2a)] [W1 | W2b]^T + (b1 + b2b)   (model/dino_pqgo.py:104-112,:127-128)
extern "C" int equss_expand_head(void* fh, const void* w2a, const float* b2a, const void* w_cat, const float* b_cat,
                                 float* code, int rows, int C, int D, void* stream) {
    if (!fh || !w2a || !b2a || !w_cat || !b_cat || !code) return equss::fail(EQUSS_E_NULL, "expand_head: null pointer argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    u16* f = reinterpret_cast<u16*>(fh);
    int rc;
    EQUSS_STEP(EQUSS_K_HEAD_GEMM1, equss::gemm_bf16(f, 2 * C, w2a, C, b2a, f + C, 2 * C, rows, C, C, EQUSS_EPI_RELU_BF16, s))
    EQUSS_STEP(EQUSS_K_HEAD_GEMM2, equss::gemm_bf16(f, 2 * C, w_cat, 2 * C, b_cat, code, D, rows, D, 2 * C, EQUSS_EPI_F32, s))
    return EQUSS_OK;
}
