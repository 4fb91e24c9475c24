// C = A * W^T + bias on the gfx950 matrix cores: bf16 operands, fp32 accumulation, fused epilogue.
//
// Serves every nn.Linear / 1x1 nn.Conv2d of the path: qkv, proj, fc1, fc2
// (model/dino/vision_transformer.py:81,:89,:59,:62), the patch embedding seen as a GEMM (:128-133)
// and the expansion head (model/dino_pqgo.py:104-112,:127-128).  Both operands are K-contiguous
// (activations are token-major, PyTorch weights are [out, in]), which is exactly what the MFMA
// fragments want, so no transposition happens anywhere.
//
// Structure (one workgroup = 4 waves = one 128 x 128 output tile, K walked in steps of 64):
//  * global_load_lds_dwordx4 moves both operand tiles HBM/L2 -> LDS without touching registers;
//    two LDS stages, the loads of step k+1 are in flight while step k is on the matrix pipe.
//  * The LDS image keeps 128-byte rows; 16-byte chunks are XOR-swizzled with (row>>1)&7 through the
//    per-lane SOURCE address (the LDS side of an LDS-DMA is lane-linear), and the same XOR on the
//    ds_read_b128 side makes every fragment read conflict-free.
//  * v_mfma_f32_16x16x32_bf16 computes C^T tiles (W rows on the MFMA "A" side, activation rows on
//    the "B" side): a lane then owns 4 consecutive output columns of one row, so the epilogue
//    stores 8 B (bf16) or 16 B (fp32) per lane straight from registers.
//  * Workgroup ids are remapped so that the tiles sharing an activation row-panel run back to back
//    on one XCD and re-read it from that XCD's L2.
#include "equss_common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int kThreads = 256;

struct GemmArgs {
    const u16* A; int64_t lda;
    const u16* W; int64_t ldw;
    const float* bias;
    void* out; int64_t ldo;
    int M, N, K, ntn;  // ntn = N / BN
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const void __attribute__((address_space(1)))*)gsrc,
        (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

int g_variant = 3;  // 3: large-tile ring kernel where it applies, persistent 128x128 kernel otherwise

template <int EPI>
__device__ __forceinline__ void epilogue(const GemmArgs& g, f32x4 (&acc)[4][4], int m0, int n0, int wave_m, int wave_n,
                                         int fr, int fq) {
    // lane owns out[m][n .. n+3]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wave_n * 64 + nt * 16 + fq * 4;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) b4 = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + wave_m * 64 + mt * 16 + fr;
            if (m >= g.M) continue;
            f32x4 v = acc[nt][mt];
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            if constexpr (EPI == EQUSS_EPI_GELU_BF16) {
                const equss::gelu_f32x2 g0 = equss::gelu_erf2(equss::gelu_f32x2{v.x, v.y});
                            const equss::gelu_f32x2 g1 = equss::gelu_erf2(equss::gelu_f32x2{v.z, v.w});
                            v.x = g0.x; v.y = g0.y; v.z = g1.x; v.w = g1.y;
            }
            if constexpr (EPI == EQUSS_EPI_RELU_BF16) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if constexpr (EPI == EQUSS_EPI_F32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + (size_t)m * g.ldo + n) = v;
            } else if constexpr (EPI == EQUSS_EPI_RESID_F32) {
                // the accumulators started from out + bias (see the tile loop), so they are the result
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + (size_t)m * g.ldo + n) = acc[nt][mt];
            } else {
                uint2 pk;
                pk.x = equss::pack_bf16(v.x, v.y);
                pk.y = equss::pack_bf16(v.z, v.w);
                *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(g.out) + (size_t)m * g.ldo + n) = pk;
            }
        }
    }
}

// VAR 0: one tile per workgroup.  VAR 1: same, fragments of a whole K step preloaded ahead of the
// MFMA cluster.  VAR 2: VAR 1 + persistent walk over tiles with the next tile's first stage
// prefetched under the epilogue.
template <int EPI, int VAR>
__global__ __launch_bounds__(kThreads, 2) void gemm_bf16_kernel(const GemmArgs g) {
    constexpr int BK = 64;
    constexpr int kStageBytes = (BM + BN) * BK * 2;
    constexpr int PIECES = 4;                   // 1-KiB LDS-DMA pieces (8 rows x 128 B) per operand per wave
    auto swz = [](int row) { return (row >> 1) & 7; };  // chunk swizzle of a 128-byte row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;

    // XCD-aware tile order: workgroup ids that share (id % 8) share an XCD; each XCD owns a
    // contiguous run of logical tiles (n fastest), so tiles that run together share the activation
    // row-panel in that XCD's L2.
    const int ntiles = ((g.M + BM - 1) / BM) * g.ntn;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int x_start = xcd * q + (xcd < r ? xcd : r);
    const int x_len = q + (xcd < r ? 1 : 0);
    const int nslots = (gridDim.x + 7 - xcd) >> 3;  // workgroups resident on this XCD group
    int it = slot;
    if (it >= x_len) return;

    const int fr = lane & 15, fq = lane >> 4;
    const int sw = swz(fr);
    int a_off[2], w_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int off = fr * (BK * 2) + (((4 * ks + fq) ^ sw) * 16);
        a_off[ks] = (wave_m * 64) * (BK * 2) + off;
        w_off[ks] = BM * BK * 2 + (wave_n * 64) * (BK * 2) + off;
    }
    const int st_row = lane >> 3;                                // row inside an 8-row piece
    int st_chunk[PIECES], lds_row[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int row = (i * 4 + wave) * 8 + st_row;
        st_chunk[i] = ((lane & 7) ^ swz(row)) * 8;
        lds_row[i] = (i * 4 + wave) * 8 * (BK * 2);
    }

    const u16* a_src[PIECES];
    const u16* w_src[PIECES];
    auto set_tile = [&](int logical, int& m0, int& n0) {
        m0 = (logical / g.ntn) * BM;
        n0 = (logical % g.ntn) * BN;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int row = (i * 4 + wave) * 8 + st_row;
            int am = m0 + row;
            if (am > g.M - 1) am = g.M - 1;  // tail rows re-read the last row; their outputs are masked
            a_src[i] = g.A + (size_t)am * g.lda + st_chunk[i];
            w_src[i] = g.W + (size_t)(n0 + row) * g.ldw + st_chunk[i];
        }
    };
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * kStageBytes;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            glds16(a_src[i] + (size_t)kt * BK, base + lds_row[i]);
            glds16(w_src[i] + (size_t)kt * BK, base + BM * BK * 2 + lds_row[i]);
        }
    };

    const int nk = g.K / BK;
    int m0, n0;
    set_tile(x_start + it, m0, n0);
    stage(0, 0);
    int buf = 0;
    while (true) {
        f32x4 acc[4][4];  // [nt][mt], element j <-> output column n + j
        if constexpr (EPI == EQUSS_EPI_RESID_F32) {
            // residual epilogue: start from out + bias and accumulate onto it, the same order of fp32
            // operations as the large-tile residual + LayerNorm kernel, so a row gets the same bits
            // whichever of the two kernels handles it
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = n0 + wave_n * 64 + nt * 16 + fq * 4;
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if (g.bias) b4 = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int m = m0 + wave_m * 64 + mt * 16 + fr;
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
                    if (m < g.M) o = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.out) + (size_t)m * g.ldo + n);
                    o.x += b4.x; o.y += b4.y; o.z += b4.z; o.w += b4.w;
                    acc[nt][mt] = o;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int cm0 = m0, cn0 = n0;
        int nxt = it + nslots;
        if (VAR < 2 || nxt >= x_len) nxt = -1;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // this K step has landed for every wave; the other buffer is no longer read
            if (kt + 1 < nk) {
                stage(buf ^ 1, kt + 1);
            } else if (nxt >= 0) {
                set_tile(x_start + nxt, m0, n0);
                stage(buf ^ 1, 0);
            }
            const char* base = smem + buf * kStageBytes;
            if constexpr (VAR == 0) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 af[4], wf[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        af[t] = *reinterpret_cast<const bf16x8*>(base + a_off[ks] + t * 16 * (BK * 2));
                        wf[t] = *reinterpret_cast<const bf16x8*>(base + w_off[ks] + t * 16 * (BK * 2));
                    }
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
                }
            } else {
                bf16x8 af[2][4], wf[2][4];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        af[ks][t] = *reinterpret_cast<const bf16x8*>(base + a_off[ks] + t * 16 * (BK * 2));
                        wf[ks][t] = *reinterpret_cast<const bf16x8*>(base + w_off[ks] + t * 16 * (BK * 2));
                    }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][nt], af[ks][mt], acc[nt][mt], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
        }
        epilogue<EPI>(g, acc, cm0, cn0, wave_m, wave_n, fr, fq);
        if (nxt < 0) {
            if (VAR < 2) break;
            it += nslots;
            if (it >= x_len) break;
            // (only reached when the prefetch was skipped, which VAR 2 never does)
            break;
        }
        it = nxt;
    }
}

template <int EPI>
int launch(const GemmArgs& g, hipStream_t s) {
    const int ntiles = ((g.M + BM - 1) / BM) * g.ntn;
    const size_t lds = 2 * (BM + BN) * 64 * 2;
    if (g_variant >= 2) {
        const int grid = ntiles < 512 ? ntiles : 512;  // two resident workgroups per CU
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI, 2>), dim3(grid), dim3(kThreads), lds, s, g);
    } else if (g_variant == 1) {
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI, 1>), dim3(ntiles), dim3(kThreads), lds, s, g);
    } else {
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI, 0>), dim3(ntiles), dim3(kThreads), lds, s, g);
    }
    return equss::check_launch("gemm_bf16_kernel");
}

}  // namespace

namespace equss {

bool gemm8_supports(int N, int K, int epilogue);
int gemm8_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo,
               int M, int N, int K, int epilogue, hipStream_t s);

bool gemm_ws_supports(int M, int N, int K, int epilogue);
int gemm_ws_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo, int M, int N,
                 int K, int epilogue, hipStream_t s);

bool gemm_sw_supports(int M, int N, int K, int64_t ldo, int epilogue);
int gemm_sw_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo, int M, int N,
                 int epilogue, hipStream_t s);

int gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo,
              int M, int N, int K, int epilogue, hipStream_t s) {
    if (!A || !W || !out) return fail(EQUSS_E_NULL, "gemm_bf16: null pointer argument");
    if (M <= 0 || N <= 0 || K <= 0) return fail(EQUSS_E_SHAPE, "gemm_bf16: M=%d N=%d K=%d must be positive", M, N, K);
    if (N % BN != 0 || K % 64 != 0)
        return fail(EQUSS_E_SHAPE, "gemm_bf16: N=%d must be a multiple of %d and K=%d of 64", N, BN, K);
    if (lda < K || ldw < K || ldo < N || (lda % 8) || (ldw % 8) || (ldo % 4) || ((uintptr_t)A % 16) ||
        ((uintptr_t)W % 16) || ((uintptr_t)out % 16) || (bias && ((uintptr_t)bias % 16)))
        return fail(EQUSS_E_ALIGN, "gemm_bf16: operands must be 16-byte aligned, lda/ldw multiples of 8, ldo of 4");
    // (bf16 output rows go out in 16-byte pieces: eight elements)
    if ((epilogue == EQUSS_EPI_BF16 || epilogue == EQUSS_EPI_GELU_BF16 || epilogue == EQUSS_EPI_RELU_BF16) && (ldo % 8))
        return fail(EQUSS_E_ALIGN, "gemm_bf16: ldo=%lld must be a multiple of 8 for a bf16 output", (long long)ldo);
    // (the in-place residual update keeps the small tiles for short M: that is what the trailing-row
    // routes of the fused kernels want)
    // long M, K = 384, N a multiple of 384, bf16 output: the weights stay in registers (developer variants 4 and 6 keep it off)
    if (g_variant == 3 && gemm_sw_supports(M, N, K, ldo, epilogue))
        return gemm_sw_bf16(A, lda, W, ldw, bias, out, ldo, M, N, epilogue, s);
    // long M, N a multiple of 384, bf16 output: the producer / consumer kernel (developer variant 4 keeps it off)
    if ((g_variant == 3 || g_variant == 6) && gemm_ws_supports(M, N, K, epilogue))
        return gemm_ws_bf16(A, lda, W, ldw, bias, out, ldo, M, N, K, epilogue, s);
    // (the large-tile kernel addresses its operand pieces with 32-bit element offsets)
    const bool fits32 = (int64_t)M * lda < (1LL << 32) && (int64_t)N * ldw < (1LL << 32);
    if ((g_variant == 3 || g_variant == 4 || g_variant == 6) && fits32 && gemm8_supports(N, K, epilogue) && (epilogue != EQUSS_EPI_RESID_F32 || M > 16384))
        return gemm8_bf16(A, lda, W, ldw, bias, out, ldo, M, N, K, epilogue, s);
    GemmArgs g;
    g.A = reinterpret_cast<const u16*>(A); g.lda = lda;
    g.W = reinterpret_cast<const u16*>(W); g.ldw = ldw;
    g.bias = bias; g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K; g.ntn = N / BN;
    switch (epilogue) {
        case EQUSS_EPI_BF16: return launch<EQUSS_EPI_BF16>(g, s);
        case EQUSS_EPI_GELU_BF16: return launch<EQUSS_EPI_GELU_BF16>(g, s);
        case EQUSS_EPI_RELU_BF16: return launch<EQUSS_EPI_RELU_BF16>(g, s);
        case EQUSS_EPI_F32: return launch<EQUSS_EPI_F32>(g, s);
        case EQUSS_EPI_RESID_F32: return launch<EQUSS_EPI_RESID_F32>(g, s);
        default: return fail(EQUSS_E_UNSUPPORTED, "gemm_bf16: unknown epilogue %d", epilogue);
    }
}

}  // namespace equss

extern "C" void equss_dev_gemm_variant(int v) { g_variant = v; }  // tuning knob, not part of the ABI

extern "C" int equss_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                               void* out, int64_t ldo, int M, int N, int K, int epilogue, void* stream) {
    return equss::gemm_bf16(A, lda, W, ldw, bias, out, ldo, M, N, K, epilogue, reinterpret_cast<hipStream_t>(stream));
}
