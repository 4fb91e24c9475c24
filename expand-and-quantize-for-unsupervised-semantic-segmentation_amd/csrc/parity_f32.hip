// fp32 "parity mode" of the ViT and the expansion head for gfx950.
//
// The throughput path rounds GEMM operands to bf16, which moves `code` by ~5e-3 relative and with it
// ~1 % of the code indices.  This file evaluates the same forward (model/dino/vision_transformer.py:
// 58-64, 79-91, 106-114, 130-133, 200-211, 227-239; model/dino_pqgo.py:104-112, 126-128) entirely in
// fp32, so that the indices can be compared with the reference's fp32 CPU run one by one: every product
// is an exact fp32 fma on the matrix pipe (v_mfma_f32_32x32x2_f32 is an fp32 fma chain), GELU uses erff,
// the softmax uses expf with a running maximum.  What is left against the reference is the summation
// order of fp32 dot products (its BLAS's versus an ascending-k chain), i.e. ~1e-6 relative on `code`.
// Built for comparison runs, not for speed: plain LDS-tiled kernels, one tile shape.
#include "equss_common.h"

namespace equss {
int layernorm_f32(const float* x, int64_t ld_x, const float* gamma, const float* beta, float* out, int rows, int C,
                  int tokens, float eps, hipStream_t s);
int assemble_tokens_f32(const float* patch, const float* cls, const float* pos, float* x, int B, int n, int C, hipStream_t s);
}  // namespace equss

namespace {

enum { F32_EPI_NONE = 0, F32_EPI_GELU = 1, F32_EPI_RELU = 2, F32_EPI_RESID = 3 };

// out = epilogue(A W^T + bias): A (M, K) fp32, W (N, K) fp32 (PyTorch [out, in]), out (M, N) fp32.
// 64 x 64 tile per workgroup, four waves of 32 x 32, K walked in slices of 32 through LDS.
// The A operand of the MFMA is the activation rows, the B operand the weight rows: a lane then owns one
// output column and 16 rows, and 32 consecutive lanes store 32 consecutive floats of a row.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ W, int64_t ldw,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int64_t ldo, int M, int N, int K) {
    constexpr int BM = 64, BN = 64, BK = 32, LD = BK + 1;
    __shared__ float As[BM * LD];
    __shared__ float Ws[BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = it * 256 + tid;
            const int r = item >> 3, c4 = item & 7;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (m0 + r < M) a = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + r) * lda + k0 + 4 * c4);
            const f32x4 w = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + r) * ldw + k0 + 4 * c4);
            float* ap = As + r * LD + 4 * c4;
            float* wp = Ws + r * LD + 4 * c4;
            ap[0] = a.x; ap[1] = a.y; ap[2] = a.z; ap[3] = a.w;
            wp[0] = w.x; wp[1] = w.y; wp[2] = w.z; wp[3] = w.w;
        }
        __syncthreads();
        const float* ar = As + (wm * 32 + li) * LD + lh;
        const float* wr = Ws + (wn * 32 + li) * LD + lh;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[kk], wr[kk], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + li;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        if (m >= M) continue;
        float v = acc[reg] + bv;
        if (EPI == F32_EPI_GELU) v = (v * 0.5f) * (1.0f + erff(v * 0.70710678118654752440f));
        if (EPI == F32_EPI_RELU) v = fmaxf(v, 0.f);
        float* o = out + (size_t)m * ldo + n;
        if (EPI == F32_EPI_RESID) v = *o + v;
        *o = v;
    }
}

int gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* out, int64_t ldo, int M,
             int N, int K, int epi, hipStream_t s) {
    if (N % 64 != 0 || K % 32 != 0 || (lda % 4) || (ldw % 4))
        return equss::fail(EQUSS_E_UNSUPPORTED, "gemm_f32: N=%d must be a multiple of 64, K=%d of 32, leading dimensions of 4", N, K);
    const dim3 grid(N / 64, (M + 63) / 64);
    switch (epi) {
        case F32_EPI_NONE: hipLaunchKernelGGL(gemm_f32_kernel<F32_EPI_NONE>, grid, dim3(256), 0, s, A, lda, W, ldw, bias, out, ldo, M, N, K); break;
        case F32_EPI_GELU: hipLaunchKernelGGL(gemm_f32_kernel<F32_EPI_GELU>, grid, dim3(256), 0, s, A, lda, W, ldw, bias, out, ldo, M, N, K); break;
        case F32_EPI_RELU: hipLaunchKernelGGL(gemm_f32_kernel<F32_EPI_RELU>, grid, dim3(256), 0, s, A, lda, W, ldw, bias, out, ldo, M, N, K); break;
        default: hipLaunchKernelGGL(gemm_f32_kernel<F32_EPI_RESID>, grid, dim3(256), 0, s, A, lda, W, ldw, bias, out, ldo, M, N, K); break;
    }
    return equss::check_launch("gemm_f32_kernel");
}

// softmax(q k^T * scale) v for head_dim 64 in fp32, flash style (vision_transformer.py:81-88).
// One wave per 32 queries, four waves per workgroup share the 32-key K / V tiles in LDS.
//   S^T tile = K Q^T on the matrix pipe: the lane owns one query and 16 of the tile's 32 keys, so the
//   running maximum / sum are an in-lane scan plus one exchange with the other half of the wave;
//   O^T += V^T P^T: the exponentials are already laid out as that MFMA's B operand (the key a lane holds
//   in register r is the key its half supplies in k-pair r), so P never leaves the registers.
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int N, int H, float scale) {
    constexpr int TK = 32, LDK = 65;
    __shared__ float Ks[TK * LDK];
    __shared__ __attribute__((aligned(16))) float Vs[TK * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qrow = blockIdx.x * 128 + wave * 32 + li;
    const bool valid = qrow < N;
    const int64_t ld = (int64_t)3 * H * 64;
    const float* base = qkv + (size_t)b * N * ld + (size_t)h * 64;
    float q[32];
#pragma unroll
    for (int t = 0; t < 32; ++t) q[t] = valid ? base[(size_t)qrow * ld + 2 * t + lh] * scale : 0.f;   // (q k) * 2^-3 == (q 2^-3) k
    float mrun = -__builtin_inff(), lrun = 0.f;
    f32x16 o0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 o1 = o0;
    const int ntiles = (N + TK - 1) / TK;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = it * 256 + tid;
            const int r = item >> 4, c4 = item & 15;
            const int key = kt * TK + r;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < N) {
                kv = *reinterpret_cast<const f32x4*>(base + (size_t)key * ld + (size_t)H * 64 + 4 * c4);
                vv = *reinterpret_cast<const f32x4*>(base + (size_t)key * ld + (size_t)2 * H * 64 + 4 * c4);
            }
            float* kp = Ks + r * LDK + 4 * c4;
            kp[0] = kv.x; kp[1] = kv.y; kp[2] = kv.z; kp[3] = kv.w;
            *reinterpret_cast<f32x4*>(Vs + r * 64 + 4 * c4) = vv;
        }
        __syncthreads();
        f32x16 sc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* kr = Ks + li * LDK + lh;
#pragma unroll
        for (int t = 0; t < 32; ++t) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * t], q[t], sc, 0, 0, 0);
        // sc[reg] = <k[key], q[query]> * scale with key = kt * 32 + (reg & 3) + 8 (reg >> 2) + 4 lh, query = this lane's
        float mx = -__builtin_inff();
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int key = kt * TK + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            if (key >= N) sc[reg] = -__builtin_inff();
            mx = fmaxf(mx, sc[reg]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);          // finite: every tile holds at least one key < N
        const float alpha = expf(mrun - mnew);       // 0 on the first tile (mrun = -inf)
        float psum = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            sc[reg] = expf(sc[reg] - mnew);          // masked keys: exp(-inf) = 0
            psum += sc[reg];
        }
        psum += __shfl_xor(psum, 32, 64);
        lrun = lrun * alpha + psum;
        mrun = mnew;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) { o0[reg] *= alpha; o1[reg] *= alpha; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kl = (r & 3) + 8 * (r >> 2) + 4 * lh;
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kl * 64 + li], sc[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[kl * 64 + 32 + li], sc[r], o1, 0, 0, 0);
        }
    }
    if (!valid) return;
    // o0[reg] = O[query][d] for d = (reg & 3) + 8 (reg >> 2) + 4 lh, o1 the same + 32
    float* orow = out + ((size_t)b * N + qrow) * ((size_t)H * 64) + (size_t)h * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 a = {o0[4 * g] / lrun, o0[4 * g + 1] / lrun, o0[4 * g + 2] / lrun, o0[4 * g + 3] / lrun};
        f32x4 c = {o1[4 * g] / lrun, o1[4 * g + 1] / lrun, o1[4 * g + 2] / lrun, o1[4 * g + 3] / lrun};
        *reinterpret_cast<f32x4*>(orow + 8 * g + 4 * lh) = a;
        *reinterpret_cast<f32x4*>(orow + 32 + 8 * g + 4 * lh) = c;
    }
}

// img (B,3,H,W) fp32 -> rows of 3*P*P fp32 in the conv weight's (c, ph, pw) order (vision_transformer.py:128-133).
__global__ __launch_bounds__(256) void patch_gather_f32_kernel(const float* __restrict__ img, float* __restrict__ cols,
                                                               int H, int W, int P, long items) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= items) return;
    const int seg = P / 4;                  // float4 segments per patch row
    const int per_patch = 3 * P * seg;
    const int gw = W / P, gh = H / P;
    const long patch = gid / per_patch;
    int rem = (int)(gid % per_patch);
    const int c = rem / (P * seg); rem %= P * seg;
    const int ph = rem / seg, sg = rem % seg;
    const int b = (int)(patch / ((long)gh * gw));
    const int pr = (int)(patch % ((long)gh * gw));
    const int py = pr / gw, px = pr % gw;
    const f32x4 v = *reinterpret_cast<const f32x4*>(img + (((size_t)b * 3 + c) * H + (size_t)py * P + ph) * W + (size_t)px * P + sg * 4);
    *reinterpret_cast<f32x4*>(cols + (size_t)patch * (3 * P * P) + (size_t)c * P * P + ph * P + sg * 4) = v;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct VitF32Ws { float *x, *ln, *qkv, *att, *hid, *cols, *tmp; size_t bytes; };
VitF32Ws carve(void* ws, int C, int B, int H, int W, int P) {
    const size_t n = (size_t)(H / P) * (W / P);
    const size_t rows = (size_t)B * (n + 1);
    VitF32Ws w;
    char* base = reinterpret_cast<char*>(ws);
    size_t off = 0;
    auto take = [&](size_t floats) { float* p = reinterpret_cast<float*>(base + off); off += align256(floats * 4); return p; };
    w.x = take(rows * C); w.ln = take(rows * C); w.qkv = take(rows * 3 * C); w.att = take(rows * C);
    w.hid = take(rows * 4 * C); w.cols = take((size_t)B * n * 3 * P * P); w.tmp = take((size_t)B * n * C);
    w.bytes = off;
    return w;
}

}  // namespace

extern "C" size_t equss_vit_f32_workspace_bytes(const EqussVit* vit, int B, int H, int W) {
    if (!vit || B <= 0 || H <= 0 || W <= 0 || vit->patch <= 0) return 0;
    return carve(nullptr, vit->embed_dim, B, H, W, vit->patch).bytes;
}

extern "C" int equss_vit_forward_f32(const EqussVit* vit, const float* img, const float* pos, float* feat, int keep_cls,
                                     void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream) {
    if (!vit || !img || !pos || !feat || !workspace || !vit->blocks)
        return equss::fail(EQUSS_E_NULL, "vit_forward_f32: null pointer argument");
    const int C = vit->embed_dim, P = vit->patch, heads = vit->heads;
    if (B <= 0 || H <= 0 || W <= 0) return equss::fail(EQUSS_E_SHAPE, "image batch %d x %d x %d must be positive", B, H, W);
    if (P != 8 && P != 16) return equss::fail(EQUSS_E_UNSUPPORTED, "patch size %d (8 or 16 supported)", P);
    if (H % P != 0 || W % P != 0) return equss::fail(EQUSS_E_SHAPE, "image %d x %d is not divisible by the patch size %d", H, W, P);
    if (C % 128 != 0 || C > 1024 || heads * 64 != C)
        return equss::fail(EQUSS_E_UNSUPPORTED, "vit_forward_f32: embed dim %d with %d heads (head dim 64, C a multiple of 128)", C, heads);
    if (((uintptr_t)img % 16) || ((uintptr_t)pos % 16) || ((uintptr_t)feat % 16) || ((uintptr_t)workspace % 256))
        return equss::fail(EQUSS_E_ALIGN, "vit_forward_f32: img/pos/feat must be 16-byte aligned, workspace 256-byte aligned");
    if (workspace_bytes < equss_vit_f32_workspace_bytes(vit, B, H, W))
        return equss::fail(EQUSS_E_WORKSPACE, "vit_forward_f32: workspace of %zu bytes is smaller than the %zu required",
                           workspace_bytes, equss_vit_f32_workspace_bytes(vit, B, H, W));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    VitF32Ws w = carve(workspace, C, B, H, W, P);
    const int n = (H / P) * (W / P), N = n + 1, rows = B * N;
    const long items = (long)B * n * 3 * P * (P / 4);
    hipLaunchKernelGGL(patch_gather_f32_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, img, w.cols, H, W, P, items);
    int rc = equss::check_launch("patch_gather_f32_kernel");
    if (rc) return rc;
    const float* patch_w = reinterpret_cast<const float*>(vit->patch_w);
    rc = gemm_f32(w.cols, 3 * P * P, patch_w, 3 * P * P, vit->patch_b, w.tmp, C, B * n, C, 3 * P * P, F32_EPI_NONE, s);
    if (rc) return rc;
    rc = equss::assemble_tokens_f32(w.tmp, vit->cls, pos, w.x, B, n, C, s);
    if (rc) return rc;
    for (int i = 0; i < vit->depth; ++i) {
        const EqussVitBlock& k = vit->blocks[i];
        const float* qkv_w = reinterpret_cast<const float*>(k.qkv_w);
        const float* proj_w = reinterpret_cast<const float*>(k.proj_w);
        const float* fc1_w = reinterpret_cast<const float*>(k.fc1_w);
        const float* fc2_w = reinterpret_cast<const float*>(k.fc2_w);
        if ((rc = equss::layernorm_f32(w.x, C, k.ln1_g, k.ln1_b, w.ln, rows, C, 0, 1e-6f, s))) return rc;
        if ((rc = gemm_f32(w.ln, C, qkv_w, C, k.qkv_b, w.qkv, 3 * C, rows, 3 * C, C, F32_EPI_NONE, s))) return rc;
        hipLaunchKernelGGL(attention_f32_kernel, dim3((N + 127) / 128, heads, B), dim3(256), 0, s, w.qkv, w.att, N, heads, 0.125f);
        if ((rc = equss::check_launch("attention_f32_kernel"))) return rc;
        if ((rc = gemm_f32(w.att, C, proj_w, C, k.proj_b, w.x, C, rows, C, C, F32_EPI_RESID, s))) return rc;
        if ((rc = equss::layernorm_f32(w.x, C, k.ln2_g, k.ln2_b, w.ln, rows, C, 0, 1e-6f, s))) return rc;
        if ((rc = gemm_f32(w.ln, C, fc1_w, C, k.fc1_b, w.hid, 4 * C, rows, 4 * C, C, F32_EPI_GELU, s))) return rc;
        if ((rc = gemm_f32(w.hid, 4 * C, fc2_w, 4 * C, k.fc2_b, w.x, C, rows, C, 4 * C, F32_EPI_RESID, s))) return rc;
    }
    return equss::layernorm_f32(w.x, C, vit->norm_g, vit->norm_b, feat, rows, C, keep_cls ? 0 : N, 1e-6f, s);
}

extern "C" int equss_expand_head_f32(const float* feat, const float* w1, const float* b1, const float* w2a, const float* b2a,
                                     const float* w2b, const float* b2b, float* hidden, float* code, int rows, int C,
                                     int D, void* stream) {
    if (!feat || !w1 || !b1 || !w2a || !b2a || !w2b || !b2b || !hidden || !code)
        return equss::fail(EQUSS_E_NULL, "expand_head_f32: null pointer argument");
    if (rows <= 0) return equss::fail(EQUSS_E_SHAPE, "expand_head_f32: rows=%d must be positive", rows);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc;
    // code = cluster1(f); code += cluster2(f)   (dino_pqgo.py:127-128)
    if ((rc = gemm_f32(feat, C, w1, C, b1, code, D, rows, D, C, F32_EPI_NONE, s))) return rc;
    if ((rc = gemm_f32(feat, C, w2a, C, b2a, hidden, C, rows, C, C, F32_EPI_RELU, s))) return rc;
    return gemm_f32(hidden, C, w2b, C, b2b, code, D, rows, D, C, F32_EPI_RESID, s);
}
