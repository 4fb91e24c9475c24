// out = epi(A W^T + bias) for K = 384, N = 384 k and long M (the qkv projection and the head's first GEMM of ViT-S:
// nn.Linear, vision_transformer.py:83; dino_pqgo.py:104-112) with the weights stationary in registers.
//
// With K this short a tiled GEMM re-stages its weight tile for every row block, and the staging (L2 -> LDS at the rate
// the per-stage barrier protocol sustains) is what the K loops of the ring kernels wait for.  Here a workgroup keeps a
// 384-column strip of W for the whole launch: eight waves, two per SIMD, each with 48 columns x 384 k of W as 36 MFMA
// operand fragments (144 registers a lane) loaded once.  Only the activations move: a 32-row block of A (24 KiB) per
// tile, global -> registers -> LDS in fragment order (lane l of a 1-KiB fragment reads bytes 16 l: conflict free), two
// tiles ahead in registers and one ahead in LDS, one workgroup barrier per tile (LDS traffic only).  A fragment read feeds
// three MFMAs of each wave.  Everything is plain C++ (compiler-placed waits): no LDS-DMA, so no counted waits to get wrong.
//
// The k strips of a row block are taken by k workgroups at about the same time (any XCD: the block comes from HBM once and
// from the memory-side cache otherwise).
// Same operand rounding and fp32 operation order as gemm_ws_kernel / gemm8_kernel (accumulate from zero over ascending
// 32-wide k slices, then bias, then the activation): equal bits.
#include "equss_common.h"

namespace {

struct SwArgs {
    const u16* A; int64_t lda;
    const u16* W; int64_t ldw;
    const float* bias;
    u16* out; int64_t ldo;
    int M, ntn;
};

constexpr int K = 384, NS = K / 32, BM = 32, MT = BM / 16, NW = 8, WN = 48, NT = WN / 16, BN = NW * WN, kThreads = NW * 64;
constexpr int kTile = BM * K * 2;                 // 24 KiB: MT * NS fragments of 1 KiB
constexpr int kFrags = MT * NS, PF = kFrags / NW; // fragments a wave stages per tile
constexpr int ROWSTRIDE = WN * 2 + 16;            // padded row of a wave's output staging tile
constexpr int kStg = BM * ROWSTRIDE;
constexpr int OCH = WN * 2 / 16, PER_LANE = BM * OCH / 64;
constexpr int kLds = 2 * kTile + NW * kStg + BN * 4;
constexpr int kAhead = 3;                         // k slices whose fragments are requested ahead of the MFMAs
static_assert(kFrags % NW == 0 && (BM * OCH) % 64 == 0, "staging split");

typedef equss::lds_u32x4 L4;
__device__ __forceinline__ f32x4 MM(const L4& a, const L4& b, f32x4 c) {
#ifndef EQUSS_ABL_SWNOMFMA   // (EQUSS_ABL_*: developer builds that leave one cost out, tools/build_ablations.sh; wrong results by design)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#else
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#endif
}

template <int EPI>
__global__ __launch_bounds__(kThreads, 1) void gemm_sw_kernel(const SwArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    // Strip bid % ntn, and among the workgroups of that strip the row blocks dealt round robin: every workgroup of the launch
    // has work, a row block is read by its ntn workgroups at about the same time, and the launch sweeps the rows as one
    // window.  (Keeping the strips of a row block on one XCD, for its L2, with a range of rows per XCD: 10 % slower, and 2 of
    // an XCD's 32 workgroups idle when ntn = 3; a contiguous range of row blocks per workgroup: 18 % slower.)
    const int nrb = (g.M + BM - 1) / BM;
    const int bid = blockIdx.x;
    const int strip = bid % g.ntn, group = bid / g.ntn;
    const int groups = ((int)gridDim.x - strip + g.ntn - 1) / g.ntn;   // workgroups of this strip
    const int x_start = 0, x_len = nrb;
    if (group >= x_len) return;
    const int my_n = (x_len - group + groups - 1) / groups;

    // ---- this wave's 48 columns of W and of the bias, for the whole launch ------------------------------------------------
    const int n0 = strip * BN + wave * WN;
    L4 w[NT][NS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const u16* wr = g.W + (size_t)(n0 + nt * 16 + fr) * g.ldw + fq * 8;
#pragma unroll
        for (int s = 0; s < NS; ++s) w[nt][s] = *reinterpret_cast<const L4*>(wr + s * 32);
    }
    // (the strip's bias waits in LDS: twelve registers a lane are worth more as fragment read-ahead)
    float* const bias_s = reinterpret_cast<float*>(smem + 2 * kTile + NW * kStg);
    for (int i = tid; i < BN; i += kThreads) bias_s[i] = g.bias ? g.bias[strip * BN + i] : 0.f;

    // ---- staging of a row block: fragment f = wave + 8 i (row tile f / 12, k slice f % 12), this lane's 16 bytes of it ------
    auto fetch = [&](int i, L4 (&pre)[PF]) {   // row block number i of this workgroup (past the end: the last one again)
        const int rb = x_start + group + groups * (i < my_n ? i : my_n - 1);
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int f = wave + NW * p, mt = f / NS, s = f % NS;
            int row = rb * BM + mt * 16 + fr;
            if (row > g.M - 1) row = g.M - 1;   // rows past the end re-read the last row; their outputs are not stored
#ifndef EQUSS_ABL_SWNOFETCH
            pre[p] = *reinterpret_cast<const L4*>(g.A + (size_t)row * g.lda + s * 32 + fq * 8);
#else
            pre[p] = L4{(uint32_t)row, (uint32_t)s, 0u, 0u};
#endif
        }
    };
    auto stage = [&](const L4 (&pre)[PF], char* slot) {
#pragma unroll
        for (int p = 0; p < PF; ++p) *reinterpret_cast<L4*>(slot + (wave + NW * p) * 1024 + lane * 16) = pre[p];
    };
    char* const stg = smem + 2 * kTile + wave * kStg;

    auto tile = [&](int i, const char* slot) {
        const int m0 = (x_start + group + groups * i) * BM;
        f32x4 acc[NT][MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // fragment reads kAhead slices ahead of their MFMAs, pinned there: a read takes longer than the other wave of the
        // SIMD needs for a slice
        L4 a[NS][MT];
        auto request = [&](int s) {
#pragma unroll
#ifndef EQUSS_ABL_SWNOLDS
            for (int mt = 0; mt < MT; ++mt) a[s][mt] = *reinterpret_cast<const L4*>(slot + (mt * NS + s) * 1024 + lane * 16);
#else
            for (int mt = 0; mt < MT; ++mt) asm volatile("" : "=v"(a[s][mt]) : "v"(slot));
#endif
        };
#pragma unroll
        for (int s = 0; s < kAhead; ++s) request(s);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + kAhead < NS) request(s + kAhead);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = MM(w[nt][s], a[s][mt], acc[nt][mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef EQUSS_ABL_SWNOEPI
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(acc[nt][mt]));
        if (m0 >= 0) return;
#endif
        // ---- epilogue: bias (+ GELU / ReLU), bf16, re-shaped through the wave's own LDS rows into 16-byte row pieces ----
        f32x4 b4[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b4[nt] = *reinterpret_cast<const f32x4*>(bias_s + wave * WN + nt * 16 + fq * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 v = acc[nt][mt];
                v.x += b4[nt].x; v.y += b4[nt].y; v.z += b4[nt].z; v.w += b4[nt].w;
                if constexpr (EPI == EQUSS_EPI_GELU_BF16) {
                    const equss::gelu_f32x2 g0 = equss::gelu_erf2(equss::gelu_f32x2{v.x, v.y});
                    const equss::gelu_f32x2 g1 = equss::gelu_erf2(equss::gelu_f32x2{v.z, v.w});
                    v.x = g0.x; v.y = g0.y; v.z = g1.x; v.w = g1.y;
                }
                if constexpr (EPI == EQUSS_EPI_RELU_BF16) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                *reinterpret_cast<uint2*>(stg + (mt * 16 + fr) * ROWSTRIDE + (nt * 16 + fq * 4) * 2) =
                    uint2{equss::pack_bf16(v.x, v.y), equss::pack_bf16(v.z, v.w)};
            }
        // (the staging rows belong to this wave: its LDS operations execute in order, the compiler must keep them in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < PER_LANE; ++q) {
            const int idx = q * 64 + lane, row = idx / OCH, ch = idx % OCH;
            const L4 d = *reinterpret_cast<const L4*>(stg + row * ROWSTRIDE + ch * 16);
#ifndef EQUSS_ABL_SWNOSTORE
            if (m0 + row < g.M) *reinterpret_cast<L4*>(g.out + (size_t)(m0 + row) * g.ldo + n0 + ch * 8) = d;
#else
            asm volatile("" ::"v"(d));
#endif
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // The barrier between tiles orders LDS traffic only (the staged block becomes visible, the slot read last becomes
    // free): __syncthreads() would also wait for the row blocks requested ahead and for the tile's stores.
    auto tile_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // ---- row blocks two ahead in registers, one ahead in LDS, one barrier per tile -------------------------------------
    L4 pre_a[PF], pre_b[PF];
    char* const slot0 = smem;
    char* const slot1 = smem + kTile;
    fetch(0, pre_a);
    fetch(1, pre_b);
    stage(pre_a, slot0);
    fetch(2, pre_a);
    tile_barrier();
    for (int i = 0; i < my_n; i += 2) {
        stage(pre_b, slot1);        // row block i + 1 (every wave is past its reads of block i - 1)
        fetch(i + 3, pre_b);
        tile(i, slot0);
        tile_barrier();
        if (i + 1 >= my_n) break;
        stage(pre_a, slot0);        // row block i + 2
        fetch(i + 4, pre_a);
        tile(i + 1, slot1);
        tile_barrier();
    }
}

template <int EPI>
int launch(const SwArgs& g, hipStream_t s) {
    auto kern = gemm_sw_kernel<EPI>;
    if (!equss::lds_opt_in(reinterpret_cast<const void*>(kern), kLds))
        return equss::fail(EQUSS_E_UNSUPPORTED, "gemm_sw: cannot reserve %d bytes of LDS", kLds);
    hipLaunchKernelGGL(kern, dim3(256), dim3(kThreads), kLds, s, g);
    return equss::check_launch("gemm_sw_kernel");
}

}  // namespace

namespace equss {

bool gemm_sw_supports(int M, int N, int K_, int64_t ldo, int epilogue) {
    return K_ == K && N % BN == 0 && N / BN <= 4 && M >= 16384 && ldo % 8 == 0 &&
           (epilogue == EQUSS_EPI_BF16 || epilogue == EQUSS_EPI_GELU_BF16 || epilogue == EQUSS_EPI_RELU_BF16);
}

int gemm_sw_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo, int M, int N,
                 int epilogue, hipStream_t s) {
    SwArgs g;
    g.A = reinterpret_cast<const u16*>(A); g.lda = lda;
    g.W = reinterpret_cast<const u16*>(W); g.ldw = ldw;
    g.bias = bias; g.out = reinterpret_cast<u16*>(out); g.ldo = ldo; g.M = M; g.ntn = N / BN;
    switch (epilogue) {
        case EQUSS_EPI_BF16: return launch<EQUSS_EPI_BF16>(g, s);
        case EQUSS_EPI_GELU_BF16: return launch<EQUSS_EPI_GELU_BF16>(g, s);
        case EQUSS_EPI_RELU_BF16: return launch<EQUSS_EPI_RELU_BF16>(g, s);
        default: return fail(EQUSS_E_UNSUPPORTED, "gemm_sw: unsupported epilogue %d", epilogue);
    }
}

}  // namespace equss
