// C = A^T B with the long dimension summed: C[m][n] = sum over rows r of A[r][m] * B[r][n], A (rows, M) and B (rows, N) bf16
// row-major, C (M, N) fp32.  The weight gradients of the training pass have this shape (rows = batch * tokens, ~10^5; M and N
// a few hundred): dW = g^T f for a 1x1 convolution / nn.Linear (the backward of model/dino_pqgo.py:104-112).
//
// Both operands are stored with the summed index slowest, which is the opposite of what an MFMA fragment wants (8
// consecutive k per lane).  The tiles go to LDS as they are ([k][m], 256-byte rows, by LDS-DMA) and the fragments come
// out through ds_read_b64_tr_b16, the transposing LDS read: a 16-lane group reads a 4 (k) x 16 (m) block and every lane
// receives one column of it.  The four k rows of a block are 256 bytes apart, i.e. on the same banks; the 16-byte chunks
// of row k are therefore stored at position chunk ^ 4 (k & 3), applied on the source side of the DMA: conflict-free.
//
// Split K: a workgroup computes a 128 x 128 tile over a range of rows and adds it to C with float atomics (two 128-byte
// runs per instruction); C is cleared by the call.  The order of the additions is not fixed.
#include "equss_common.h"

namespace {

constexpr int TM = 128, TN = 128, BK = 32;
constexpr int kRowB = TM * 2;                 // bytes per k row of a tile
constexpr int kTileB = BK * kRowB;            // 8 KiB per operand tile
constexpr int kStageB = 2 * kTileB;           // A tile + B tile
constexpr int kTnThreads = 256;

struct TnArgs {
    const u16* A; int64_t lda;
    const u16* B; int64_t ldb;
    float* C; int64_t ldc;
    int rows, M, N, splits, rows_per_split;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t range_rsrc(const void* p, int64_t ld_bytes, int row0, int nrows) {
    // rows [row0, row0 + nrows) of a row-major matrix: loads past the end return zero (what a partial K step needs)
    const uint64_t base = reinterpret_cast<uint64_t>(p) + (uint64_t)row0 * (uint64_t)ld_bytes;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane((int)((int64_t)nrows * ld_bytes)), 0x00020000);
}

__global__ __launch_bounds__(kTnThreads, 2) void gemm_tn_kernel(const TnArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = g.N / TN, tiles = (g.M / TM) * tiles_n;
    const int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
    const int k0 = split * g.rows_per_split;
    const int kn = min(g.rows_per_split, g.rows - k0);
    if (kn <= 0) return;
    const int nsteps = (kn + BK - 1) / BK;

    // ---- staging: a piece = 4 k rows x 256 bytes = one wave-instruction; wave w moves pieces w and w + 4 of either tile
    const __amdgpu_buffer_rsrc_t a_rs = range_rsrc(g.A, g.lda * 2, k0, kn);
    const __amdgpu_buffer_rsrc_t b_rs = range_rsrc(g.B, g.ldb * 2, k0, kn);
    const int prow = lane >> 4, gch = (lane & 15) ^ (4 * (prow & 3));   // row within the piece, source chunk of this lane
    uint32_t a_lane[2], b_lane[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 4 * (wave + 4 * i) + prow;
        a_lane[i] = (uint32_t)(((int64_t)row * g.lda + m0 + 8 * gch) * 2);
        b_lane[i] = (uint32_t)(((int64_t)row * g.ldb + n0 + 8 * gch) * 2);
    }
    auto stage = [&](int step, int buf) {
        char* sb = smem + buf * kStageB;
        const uint32_t a_step = (uint32_t)((int64_t)step * BK * g.lda * 2), b_step = (uint32_t)((int64_t)step * BK * g.ldb * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = wave + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (__attribute__((address_space(3))) void*)(sb + p * 1024), 16, a_lane[i], a_step, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (__attribute__((address_space(3))) void*)(sb + kTileB + p * 1024), 16, b_lane[i], b_step, 0, 0);
        }
    };

    // ---- fragment reads: wave tile 64 x 64 = 2 x 2 MFMA tiles of 32 x 32; a fragment = 32 columns x 8 k per half-wave
    const int wm = wave >> 1, wn = wave & 1;
    const int q = (lane >> 2) & 3, pp = lane & 3, hg = (lane >> 4) & 1, kb = lane >> 5;
    // byte address of (k row 8 kb + q, columns c0 + 16 hg + 4 pp ...) in a tile: the row's 16-byte chunk c sits at c ^ 4 (k & 3)
    const uint32_t lds0 = equss::lds_addr(smem);
    auto frag_addr = [&](int c0) {
        const int chunk = c0 / 8 + 2 * hg + (pp >> 1);
        return lds0 + (uint32_t)((8 * kb + q) * kRowB + ((chunk ^ (4 * q)) * 16) + (pp & 1) * 8);
    };
    uint32_t a_addr[2], b_addr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        a_addr[t] = frag_addr(wm * 64 + t * 32);
        b_addr[t] = frag_addr(wn * 64 + t * 32) + kTileB;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // One 16-row k step of buffer BUF: 8 transposing reads, then 4 MFMAs.
    auto kstep = [&](auto buf_c, auto ks_c) {
        constexpr int OFF = decltype(buf_c)::value * kStageB + decltype(ks_c)::value * 16 * kRowB;
        equss::lds_u32x2 al[2][2], bl[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t aa = a_addr[t], bb = b_addr[t];
            al[t][0] = equss::lds_load_tr16_b64_at<OFF>(aa);
            al[t][1] = equss::lds_load_tr16_b64_at<OFF + 4 * kRowB>(aa);
            bl[t][0] = equss::lds_load_tr16_b64_at<OFF>(bb);
            bl[t][1] = equss::lds_load_tr16_b64_at<OFF + 4 * kRowB>(bb);
        }
        equss::lds_wait_for<0>(al[0][0], al[0][1], al[1][0], al[1][1], bl[0][0], bl[0][1], bl[1][0], bl[1][1]);
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const equss::lds_u32x4 av = {al[t][0][0], al[t][0][1], al[t][1][0], al[t][1][1]};
            const equss::lds_u32x4 bv = {bl[t][0][0], bl[t][0][1], bl[t][1][0], bl[t][1][1]};
            af[t] = __builtin_bit_cast(bf16x8, av);
            bf[t] = __builtin_bit_cast(bf16x8, bv);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    stage(0, 0);
    for (int step = 0; step < nsteps; step += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();        // stage `step` has landed for every wave; buffer 1 is no longer read
        asm volatile("" ::: "memory");
        if (step + 1 < nsteps) stage(step + 1, 1);
        kstep(I0(), I0());
        kstep(I0(), I1());
        if (step + 1 >= nsteps) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (step + 2 < nsteps) stage(step + 2, 0);
        kstep(I1(), I0());
        kstep(I1(), I1());
    }

    // ---- C += tile: register r of an accumulator is row (r & 3) + 8 (r >> 2) + 4 kb of the 32 x 32 tile, lane & 31 its column
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* cb = g.C + (int64_t)(m0 + wm * 64 + i * 32 + 4 * kb) * g.ldc + n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(cb + (int64_t)((r & 3) + 8 * (r >> 2)) * g.ldc, acc[i][j][r]);
        }
}

}  // namespace

extern "C" int equss_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                                  int rows, int M, int N, void* stream) {
    if (!A || !B || !C) return equss::fail(EQUSS_E_NULL, "equss_gemm_tn_bf16: null pointer argument");
    if (rows <= 0 || M <= 0 || N <= 0 || (M % TM) != 0 || (N % TN) != 0)
        return equss::fail(EQUSS_E_SHAPE, "equss_gemm_tn_bf16: rows=%d M=%d N=%d (M and N multiples of 128)", rows, M, N);
    if (lda < M || ldb < N || ldc < N || (lda % 8) != 0 || (ldb % 8) != 0 || ((uintptr_t)A % 16) != 0 || ((uintptr_t)B % 16) != 0 ||
        ((uintptr_t)C % 16) != 0)
        return equss::fail(EQUSS_E_ALIGN, "equss_gemm_tn_bf16: A and B 16-byte aligned with leading dimensions multiples of 8");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    TnArgs g;
    g.A = reinterpret_cast<const u16*>(A); g.lda = lda; g.B = reinterpret_cast<const u16*>(B); g.ldb = ldb;
    g.C = C; g.ldc = ldc; g.rows = rows; g.M = M; g.N = N;
    const int tiles = (M / TM) * (N / TN);
    int splits = (1024 + tiles - 1) / tiles;                      // about four workgroups per CU in all
    const int max_splits = (rows + 4 * BK - 1) / (4 * BK);        // at least four K steps per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    g.rows_per_split = ((rows + splits - 1) / splits + BK - 1) / BK * BK;
    g.splits = (rows + g.rows_per_split - 1) / g.rows_per_split;
    // 32-bit byte offsets inside a workgroup's row range (buffer addressing)
    if ((int64_t)(g.rows_per_split + BK) * (lda > ldb ? lda : ldb) * 2 >= ((int64_t)1 << 31))
        return equss::fail(EQUSS_E_UNSUPPORTED, "equss_gemm_tn_bf16: a row range of %d rows x %lld elements exceeds 2 GiB", g.rows_per_split,
                           (long long)(lda > ldb ? lda : ldb));
    if (hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)M, s) != hipSuccess)
        return equss::fail(EQUSS_E_UNSUPPORTED, "equss_gemm_tn_bf16: cannot clear C");
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(tiles * g.splits)), dim3(kTnThreads), 2 * kStageB, s, g);
    return equss::check_launch("gemm_tn_kernel");
}
