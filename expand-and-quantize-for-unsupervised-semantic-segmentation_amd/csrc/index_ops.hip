// Code-index bookkeeping next to the PQ assign, HBM-bound integer kernels:
//  * index_rows_kernel: the (rows, num_pq) int16 index table + int16 labels that the reference's offline
//    analyses read (train.py:283-329: indices nearest-upsampled to the label size, flattened to
//    (b, H*W, num_pq), appended per batch), written directly from the (num_pq, b, h, w) int64 index maps.
//    The stock-op formulation materialises the up-sampled (num_pq, b, H, W) int64 tensor first
//    (419 MB per 64-image batch at 320 x 320); here every output pixel reads its num_pq source indices
//    (6.5 MB, L2 resident) and writes one 2*num_pq-byte row.
//  * code_usage_kernel: counts[p][k] += number of rows assigned to codeword k of sub-space p, the one-hot
//    sum of model/dino_pqgo.py:668-672 (and quantizer.py:159-160, :485-487) as an LDS histogram with
//    integer atomics (result independent of scheduling).
#include "equss_common.h"

namespace {

// one thread per output pixel; NPQ > 0: compile-time row width (row store in 16-byte pieces)
template <int NPQ>
__global__ __launch_bounds__(256) void index_rows_kernel(const int64_t* __restrict__ z_index,
                                                         const int64_t* __restrict__ label,
                                                         int16_t* __restrict__ idx_rows, int16_t* __restrict__ lab_rows,
                                                         int B, int h, int w, int H, int W, int num_pq, int upsample) {
    const int64_t n_out = upsample ? (int64_t)B * H * W : (int64_t)B * h * w;
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= n_out) return;
    const int oh = upsample ? H : h, ow = upsample ? W : w;
    const int b = (int)(pix / ((int64_t)oh * ow));
    const int rem = (int)(pix % ((int64_t)oh * ow));
    const int oy = rem / ow, ox = rem % ow;
    // nearest neighbour as F.interpolate(mode="nearest") picks it: source = floor(dst * in / out)
    int sy, sx, ly, lx;
    if (upsample) { sy = (int)(((int64_t)oy * h) / H); sx = (int)(((int64_t)ox * w) / W); ly = oy; lx = ox; }
    else { sy = oy; sx = ox; ly = (int)(((int64_t)oy * H) / h); lx = (int)(((int64_t)ox * W) / w); }
    const int64_t plane = (int64_t)B * h * w;
    const int64_t src = ((int64_t)b * h + sy) * w + sx;
    lab_rows[pix] = (int16_t)label[((int64_t)b * H + ly) * W + lx];
    if constexpr (NPQ > 0) {
        int16_t row[NPQ];
#pragma unroll
        for (int p = 0; p < NPQ; ++p) row[p] = (int16_t)z_index[p * plane + src];
        if constexpr (NPQ % 8 == 0) {
#pragma unroll
            for (int v = 0; v < NPQ / 8; ++v)
                reinterpret_cast<uint4*>(idx_rows + pix * NPQ)[v] = reinterpret_cast<const uint4*>(row)[v];
        } else {
#pragma unroll
            for (int p = 0; p < NPQ; ++p) idx_rows[pix * NPQ + p] = row[p];
        }
    } else {
        for (int p = 0; p < num_pq; ++p) idx_rows[pix * num_pq + p] = (int16_t)z_index[p * plane + src];
    }
}

__global__ __launch_bounds__(256) void code_usage_kernel(const int64_t* __restrict__ idx, float* __restrict__ counts,
                                                         int64_t rows, int K) {
    extern __shared__ unsigned int hist[];
    const int p = blockIdx.y;
    for (int k = threadIdx.x; k < K; k += blockDim.x) hist[k] = 0u;
    __syncthreads();
    const int64_t* src = idx + (int64_t)p * rows;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = src[r];
        if (k >= 0 && k < K) atomicAdd(&hist[k], 1u);
    }
    __syncthreads();
    // counts are whole numbers far below 2^24 per launch, so float atomic adds of them are exact and
    // their order does not matter
    for (int k = threadIdx.x; k < K; k += blockDim.x)
        if (hist[k]) atomicAdd(&counts[(int64_t)p * K + k], (float)hist[k]);
}

// sums[p][k][:] += z[r, p*d : (p+1)*d] for every row r with idx[p][r] == k: the one_hot^T @ z of the EMA codebook update
// (model/quantizer.py:485-491) without the (rows x K) one-hot matrix.  One wave per row and sub-space, a lane per column
// (two passes for d = 128), float atomic adds into the K x d table of the sub-space: the additions of a codeword's rows
// come in no fixed order, so the sums agree with the matrix product to rounding, not bit for bit.
__global__ __launch_bounds__(256) void code_sums_kernel(const float* __restrict__ z, int64_t ld_z, const int64_t* __restrict__ idx,
                                                        float* __restrict__ sums, int64_t rows, int K, int d) {
    const int p = blockIdx.y, lane = threadIdx.x & 63;
    const int64_t* src = idx + (int64_t)p * rows;
    float* table = sums + (int64_t)p * K * d;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const int64_t k = src[r];
        if (k < 0 || k >= K) continue;
        const float* zr = z + r * ld_z + (int64_t)p * d;
        for (int c = lane; c < d; c += 64) atomicAdd(&table[k * d + c], zr[c]);
    }
}

// Point sampling of the correspondence loss (model/loss.py:647-648: grid_sample, bilinear, of a (B, C, h, w) map at
// random coordinates) on token-major rows: out[p][:] = sum over the four taps k of wgt[p][k] * rows[idx[p][k]][:].
// One wave per point, a lane per four columns; a tap outside [0, n_rows) contributes nothing.
__global__ __launch_bounds__(256) void tap_gather_kernel(const float* __restrict__ rows, int64_t ld_rows, const int64_t* __restrict__ idx,
                                                        const float* __restrict__ wgt, float* __restrict__ out, int64_t ld_out,
                                                        int64_t points, int C, int64_t n_rows) {
    const int lane = threadIdx.x & 63;
    for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < points; p += (int64_t)gridDim.x * 4) {
        int64_t t[4];
        float w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t[k] = idx[p * 4 + k];
            w[k] = (t[k] >= 0 && t[k] < n_rows) ? wgt[p * 4 + k] : 0.f;
            t[k] = (t[k] >= 0 && t[k] < n_rows) ? t[k] : 0;
        }
        for (int c = lane * 4; c < C; c += 256) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(rows + t[k] * ld_rows + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = fmaf(w[k], v[i], acc[i]);
            }
            *reinterpret_cast<f32x4*>(out + p * ld_out + c) = acc;
        }
    }
}

// Its backward: grad_rows[idx[p][k]][:] += wgt[p][k] * g[p][:], float atomic adds of whole rows (a wave-instruction adds
// 64 consecutive floats).  Points that share a token come in no fixed order.
__global__ __launch_bounds__(256) void tap_scatter_kernel(const float* __restrict__ g, int64_t ld_g, const int64_t* __restrict__ idx,
                                                         const float* __restrict__ wgt, float* __restrict__ grad_rows, int64_t ld_rows,
                                                         int64_t points, int C, int64_t n_rows) {
    const int lane = threadIdx.x & 63;
    for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < points; p += (int64_t)gridDim.x * 4) {
        int64_t t[4];
        float w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t[k] = idx[p * 4 + k];
            w[k] = wgt[p * 4 + k];
        }
        for (int c = lane; c < C; c += 64) {
            const float v = g[p * ld_g + c];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t[k] >= 0 && t[k] < n_rows && w[k] != 0.f) atomicAdd(grad_rows + t[k] * ld_rows + c, w[k] * v);
        }
    }
}

}  // namespace

extern "C" int equss_tap_gather(const float* rows, int64_t ld_rows, const int64_t* idx, const float* wgt, float* out, int64_t ld_out,
                                int64_t points, int C, int64_t n_rows, void* stream) {
    if (!rows || !idx || !wgt || !out) return equss::fail(EQUSS_E_NULL, "tap_gather: null pointer argument");
    if (points < 0 || C <= 0 || (C % 4) != 0 || n_rows <= 0)
        return equss::fail(EQUSS_E_SHAPE, "tap_gather: points=%lld C=%d n_rows=%lld (C a multiple of 4)", (long long)points, C, (long long)n_rows);
    if (ld_rows < C || ld_out < C || (ld_rows % 4) != 0 || (ld_out % 4) != 0 || ((uintptr_t)rows % 16) != 0 || ((uintptr_t)out % 16) != 0)
        return equss::fail(EQUSS_E_ALIGN, "tap_gather: rows and out 16-byte aligned with leading dimensions multiples of 4");
    if (points == 0) return EQUSS_OK;
    int64_t gx = (points + 3) / 4;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(tap_gather_kernel, dim3((unsigned)gx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rows, ld_rows, idx, wgt,
                       out, ld_out, points, C, n_rows);
    return equss::check_launch("tap_gather_kernel");
}

extern "C" int equss_tap_scatter(const float* g, int64_t ld_g, const int64_t* idx, const float* wgt, float* grad_rows, int64_t ld_rows,
                                 int64_t points, int C, int64_t n_rows, void* stream) {
    if (!g || !idx || !wgt || !grad_rows) return equss::fail(EQUSS_E_NULL, "tap_scatter: null pointer argument");
    if (points < 0 || C <= 0 || n_rows <= 0 || ld_g < C || ld_rows < C)
        return equss::fail(EQUSS_E_SHAPE, "tap_scatter: points=%lld C=%d n_rows=%lld ld_g=%lld ld_rows=%lld", (long long)points, C,
                           (long long)n_rows, (long long)ld_g, (long long)ld_rows);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemset2DAsync(grad_rows, (size_t)ld_rows * sizeof(float), 0, (size_t)C * sizeof(float), (size_t)n_rows, s) != hipSuccess)
        return equss::fail(EQUSS_E_UNSUPPORTED, "tap_scatter: cannot clear the gradient rows");
    if (points == 0) return EQUSS_OK;
    int64_t gx = (points + 3) / 4;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(tap_scatter_kernel, dim3((unsigned)gx), dim3(256), 0, s, g, ld_g, idx, wgt, grad_rows, ld_rows, points, C, n_rows);
    return equss::check_launch("tap_scatter_kernel");
}

extern "C" int equss_code_sums(const float* z, int64_t ld_z, const int64_t* idx, float* sums, int64_t rows, int num_pq, int K,
                               int d, void* stream) {
    if (!z || !idx || !sums) return equss::fail(EQUSS_E_NULL, "code_sums: null pointer argument");
    if (rows < 0 || num_pq <= 0 || K <= 0 || d <= 0 || ld_z < (int64_t)num_pq * d)
        return equss::fail(EQUSS_E_SHAPE, "code_sums: rows=%lld num_pq=%d K=%d d=%d ld_z=%lld", (long long)rows, num_pq, K, d, (long long)ld_z);
    if (rows == 0) return EQUSS_OK;
    int64_t gx = (rows + 3) / 4;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(code_sums_kernel, dim3((unsigned)gx, num_pq), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), z, ld_z, idx,
                       sums, rows, K, d);
    return equss::check_launch("code_sums_kernel");
}

extern "C" int equss_index_rows_i16(const int64_t* z_index, const int64_t* label, int16_t* idx_rows, int16_t* lab_rows,
                                    int B, int h, int w, int H, int W, int num_pq, int upsample, void* stream) {
    if (!z_index || !label || !idx_rows || !lab_rows) return equss::fail(EQUSS_E_NULL, "index_rows: null pointer argument");
    if (B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || num_pq <= 0)
        return equss::fail(EQUSS_E_SHAPE, "index_rows: B=%d h=%d w=%d H=%d W=%d num_pq=%d must be positive", B, h, w, H, W, num_pq);
    if ((uintptr_t)idx_rows % 16) return equss::fail(EQUSS_E_ALIGN, "index_rows: idx_rows must be 16-byte aligned");
    const int64_t n_out = upsample ? (int64_t)B * H * W : (int64_t)B * h * w;
    const dim3 grid((unsigned)((n_out + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define EQUSS_IDX_CASE(N)                                                                                              \
    case N:                                                                                                            \
        hipLaunchKernelGGL(index_rows_kernel<N>, grid, dim3(256), 0, s, z_index, label, idx_rows, lab_rows, B, h, w, H, W, \
                           num_pq, upsample);                                                                          \
        break;
    switch (num_pq) {
        EQUSS_IDX_CASE(8) EQUSS_IDX_CASE(16) EQUSS_IDX_CASE(32)
        default:
            hipLaunchKernelGGL(index_rows_kernel<0>, grid, dim3(256), 0, s, z_index, label, idx_rows, lab_rows, B, h, w, H, W,
                               num_pq, upsample);
    }
#undef EQUSS_IDX_CASE
    return equss::check_launch("index_rows_kernel");
}

extern "C" int equss_code_usage(const int64_t* idx, float* counts, int64_t rows, int num_pq, int K, void* stream) {
    if (!idx || !counts) return equss::fail(EQUSS_E_NULL, "code_usage: null pointer argument");
    if (rows < 0 || num_pq <= 0 || K <= 0 || K > 8192)
        return equss::fail(EQUSS_E_SHAPE, "code_usage: rows=%lld num_pq=%d K=%d (K at most 8192)", (long long)rows, num_pq, K);
    if (rows == 0) return EQUSS_OK;
    int64_t gx = (rows + 256 * 16 - 1) / (256 * 16);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(code_usage_kernel, dim3((unsigned)gx, num_pq), dim3(256), K * sizeof(unsigned int),
                       reinterpret_cast<hipStream_t>(stream), idx, counts, rows, K);
    return equss::check_launch("code_usage_kernel");
}
