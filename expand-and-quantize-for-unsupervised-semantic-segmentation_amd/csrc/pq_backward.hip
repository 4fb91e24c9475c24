// Backward of the training-pass quantisation with l2-normalised codes and raw codeword gather
// (model/dino_pqgo.py:613-690 under autograd; the reference lets torch differentiate ~15 elementwise operations on
// (rows, D) tensors).  With zn = z / max(|z|, 1e-12) per sub-space, emb = books[idx] and
//     vq = mean_p( book * mse(emb, zn.detach()) + beta * mse(zn, emb.detach()) ),   zq = zn + (emb - zn).detach()
// the gradients are, with s = g_vq * 2 / (rows * d * num_pq):
//     d zn   = g_zq + beta * s * (zn - emb)          d z = (d zn - zn <zn, d zn>) / |z|
//     d books[p][k] = -book * s * sum over the rows with idx[p] = k of (zn - emb)
// One pass over the rows: z, emb and g_zq are read once, d z written once; the codebook gradient is accumulated with
// float atomic adds of whole sub-vectors (consecutive addresses, the shape the memory-side adders take at full rate).
// The additions of a codeword's rows come in no fixed order: d books agrees between calls to rounding, not bit for bit
// (as torch's index_add_, which this replaces).
#include "equss_common.h"

namespace {

struct PqBwdArgs {
    const float* z; int64_t ld_z;
    const float* emb; int64_t ld_emb;
    const int64_t* idx;          // (num_pq, rows)
    const float* g_zq; int64_t ld_g;   // or null
    const float* g_vq;           // one float on the device
    float* g_z; int64_t ld_gz;   // or null
    float* g_books;              // (num_pq, K, d), zero on entry, or null
    float beta, book, inv_count; // inv_count = 2 / (rows * d * num_pq)
    int rows, num_pq, K;
};

// D / 4 lanes hold one sub-vector as a float4 each; a wave covers 256 consecutive floats of a row.
template <int D>
__global__ __launch_bounds__(256) void pq_st_backward_kernel(const PqBwdArgs a) {
    constexpr int LPR = D / 4;
    __shared__ float stage_s[4][256];
    const int lane = threadIdx.x & 63;
    float* stage = stage_s[threadIdx.x >> 6];
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int cols = a.num_pq * D;
    const int wpr = (cols + 255) / 256;              // waves per row
    const int64_t units = (int64_t)a.rows * wpr;
    const float s = a.g_vq[0] * a.inv_count;
    const float sb = -a.book * s, sz = a.beta * s;
    for (int64_t u = wave; u < units; u += nwaves) {
        const int64_t r = u / wpr;
        const int col0 = (int)(u % wpr) * 256;
        const int col = col0 + lane * 4;
        const bool active = col < cols;              // (whole sub-vectors drop out together: 256 % D == 0)
        const int colc = active ? col : 0;
        const f32x4 z = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.z + r * a.ld_z + colc));
        const f32x4 e = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.emb + r * a.ld_emb + colc));
        f32x4 gq = {0.f, 0.f, 0.f, 0.f};
        if (a.g_zq) gq = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.g_zq + r * a.ld_g + colc));
        const float n2 = equss::group_sum<LPR>(z.x * z.x + z.y * z.y + z.z * z.z + z.w * z.w);
        const float norm = fmaxf(sqrtf(n2), 1e-12f);   // F.normalize's eps
        f32x4 zn, df, gn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            zn[i] = z[i] / norm;
            df[i] = zn[i] - e[i];
            gn[i] = fmaf(df[i], sz, gq[i]);
        }
        if (a.g_books) {
            // Through the wave's LDS row, so that each atomic instruction adds 64 consecutive floats (lane l of step i
            // holds float 64 i + l of the wave's 256): the memory-side adders take 256 consecutive bytes per
            // instruction at full rate, a lane's own four floats per instruction at a quarter of it.
            f32x4 sv;
#pragma unroll
            for (int i = 0; i < 4; ++i) sv[i] = df[i] * sb;
            if (active) *reinterpret_cast<f32x4*>(stage + lane * 4) = sv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = col0 + i * 64 + lane;
                if (c < cols) {
                    const int p = c / D;
                    const int64_t k = a.idx[(int64_t)p * a.rows + r];
                    const float v = stage[i * 64 + lane];
                    if (k >= 0 && k < a.K)   // (an index outside the codebook contributes nothing rather than faulting)
                        atomicAdd(a.g_books + ((int64_t)p * a.K + k) * D + (c - p * D), v);
                }
            }
        }
        if (a.g_z) {
            const float proj = equss::group_sum<LPR>(zn.x * gn.x + zn.y * gn.y + zn.z * gn.z + zn.w * gn.w);
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (gn[i] - zn[i] * proj) / norm;
            if (active) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(a.g_z + r * a.ld_gz + col));
        }
    }
}

}  // namespace

extern "C" int equss_pq_st_backward(const float* z, int64_t ld_z, const float* emb, int64_t ld_emb, const int64_t* idx,
                                    const float* g_zq, int64_t ld_g, const float* g_vq, float beta, float book,
                                    float* g_z, int64_t ld_gz, float* g_books, int rows, int num_pq, int K, int d,
                                    void* stream) {
    if (!z || !emb || !idx || !g_vq) return equss::fail(EQUSS_E_NULL, "equss_pq_st_backward: null pointer argument");
    if (rows <= 0 || num_pq <= 0 || K <= 0) return equss::fail(EQUSS_E_SHAPE, "equss_pq_st_backward: rows=%d num_pq=%d K=%d", rows, num_pq, K);
    if (d != 16 && d != 32 && d != 64 && d != 128)
        return equss::fail(EQUSS_E_UNSUPPORTED, "equss_pq_st_backward: sub-space dim %d (16, 32, 64 or 128)", d);
    const int64_t cols = (int64_t)num_pq * d;
    auto bad = [&](const float* p, int64_t ld) { return p && (ld < cols || (ld % 4) != 0 || ((uintptr_t)p % 16) != 0); };
    if (bad(z, ld_z) || bad(emb, ld_emb) || bad(g_zq, ld_g) || bad(g_z, ld_gz))
        return equss::fail(EQUSS_E_ALIGN, "equss_pq_st_backward: row buffers need 16-byte alignment and leading dimensions >= num_pq * d, multiples of 4");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (g_books && hipMemsetAsync(g_books, 0, (size_t)num_pq * K * d * sizeof(float), s) != hipSuccess)
        return equss::fail(EQUSS_E_UNSUPPORTED, "equss_pq_st_backward: cannot clear the codebook gradient");
    PqBwdArgs a;
    a.z = z; a.ld_z = ld_z; a.emb = emb; a.ld_emb = ld_emb; a.idx = idx; a.g_zq = g_zq; a.ld_g = ld_g; a.g_vq = g_vq;
    a.g_z = g_z; a.ld_gz = ld_gz; a.g_books = g_books; a.beta = beta; a.book = book;
    a.inv_count = (float)(2.0 / ((double)rows * d * num_pq));
    a.rows = rows; a.num_pq = num_pq; a.K = K;
    const int64_t units = (int64_t)rows * ((cols + 255) / 256);
    int64_t blocks = (units + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    switch (d) {
        case 16: hipLaunchKernelGGL(pq_st_backward_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
        case 32: hipLaunchKernelGGL(pq_st_backward_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
        case 64: hipLaunchKernelGGL(pq_st_backward_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(pq_st_backward_kernel<128>, dim3((unsigned)blocks), dim3(256), 0, s, a); break;
    }
    return equss::check_launch("pq_st_backward_kernel");
}
