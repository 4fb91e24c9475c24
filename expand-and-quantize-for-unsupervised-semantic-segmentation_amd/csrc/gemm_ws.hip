// Wave-specialised C = A * W^T + bias (bf16 MFMA, fp32 accumulate, bf16 store epilogues): the producer /
// consumer form of gemm8_bf16.hip for N a multiple of 384.
//
// Why.  On this part the LDS-DMA requests of a wave and its MFMAs do not overlap: a wave that issues its pieces of
// the next ring stage sits at the issue of those vector-memory instructions while the address unit (64 B per clock
// per CU: 512 clocks for a 32 KiB stage, the same order as the stage's 768 clocks of MFMAs) works them off, and
// the MFMAs that follow in its instruction stream wait behind them.  With eight waves that all do both, in lock
// step, the two times add (tools/ubench/ring_real.hip: 35-50 GB/s of staging per CU next to MFMAs, against
// 130 GB/s alone).  Requests of OTHER waves do overlap: tools/ubench/dma_vs_mfma.hip runs eight MFMA + fragment
// read waves and two to four request-only waves on a CU at 108-111 GB/s of staging with the MFMA waves 6-12 %
// slower than alone.
//
// So: one workgroup per CU = 12 waves.  Waves 0-7 (2 x 4, two per SIMD) are consumers: fragment reads, MFMAs, the
// store epilogue; they issue no loads and never wait on vmcnt.  Waves 8-11 (one per SIMD) are producers: they walk
// the same (tile, K step) sequence S - 1 stages ahead and only request pieces and wait for them.  One raw s_barrier
// per K step, executed by all 12 waves, carries both hand-overs: "stage k has landed" (the producers waited for it
// before they arrived) and "the slot of stage k - 1 is free" (the consumers' reads of it returned before they
// arrived).  The consumers' epilogue has no barrier: it re-shapes through LDS rows of its own, and the producers
// keep the ring full meanwhile, across tiles.  Twelve waves per CU cap a wave at 168 registers: 96 accumulators +
// 40 fragment registers fit, the one-stage-ahead fragment pipelining of gemm8 does not and is not needed.
// Tile 128 x 384, BK = 32, S ring slots of 32 KiB; fragment layout, chunk swizzle, the C^T trick and the fp32
// operation order per output element are those of gemm8_bf16.hip (same bits).
#include "equss_common.h"

namespace {

struct WsArgs {
    const u16* A; int64_t lda;
    const u16* W; int64_t ldw;
    const float* bias;
    u16* out; int64_t ldo;
    int M, N, K, ntn;
};

constexpr int BM = 128, BN = 384;
constexpr int NC = 8, NP = 4, WN = 4;             // consumer waves (2 x 4), producer waves
constexpr int kThreads = (NC + NP) * 64;
constexpr int TM = 64, TN = 96, MT = 4, NT = 6;
// K step BK = 32: 64-byte tile rows, 1-KiB pieces of 16 rows, four ring slots of 32 KiB.
// K step BK = 64: 128-byte tile rows, pieces of 8 rows, two slots of 64 KiB.  A request for a 64-byte row piece costs the
// address path what one for a 128-byte piece costs (tools/ubench/ring_ws.hip: 66 against 96-115 GB/s per CU with idle
// consumers), and the wide step has half the barriers.
template <int BK> struct Geo {
    static constexpr int ROWB = BK * 2;                      // bytes per tile row
    static constexpr int RPP = 1024 / ROWB;                  // rows per 1-KiB piece
    static constexpr int kStage = (BM + BN) * ROWB;
    static constexpr int PIECES = kStage / 1024;
    static constexpr int PW = PIECES / NP;                   // pieces per producer wave
    static constexpr int A_PIECES = BM / RPP;
};
constexpr int ROWSTRIDE = TN * 2 + 16, RP = 16, OCH = TN * 2 / 16, PER_LANE = RP * OCH / 64;   // epilogue re-shaping
constexpr int kStg = RP * ROWSTRIDE;              // staging bytes per consumer wave

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const void __attribute__((address_space(1)))*)gsrc,
        (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}
// 16-byte chunk swizzle of a tile row (see gemm8_bf16.hip): 4 chunks per 64-byte row, 8 per 128-byte row
template <int BK> __device__ __forceinline__ int swz(int row) {
    if constexpr (BK == 32) return (0x78 >> (2 * ((row >> 2) & 3))) & 3;
    else return (row >> 1) & 7;
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EPI, int S, int BK>
__global__ __launch_bounds__(kThreads, 1) void gemm_ws_kernel(const WsArgs g) {
    using G = Geo<BK>;
    constexpr int ROWB = G::ROWB, RPP = G::RPP, kStage = G::kStage, PW = G::PW, A_PIECES = G::A_PIECES;
    constexpr int CPR = ROWB / 16;                    // chunks per row = lanes per row of a piece
    static_assert(S >= 2 && S <= 5, "ring depth");
    static_assert((S - 2) * PW <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int ntiles = ((g.M + BM - 1) / BM) * g.ntn;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int x_start = xcd * q + (xcd < r ? xcd : r);
    const int x_len = q + (xcd < r ? 1 : 0);
    const int nslots = (gridDim.x + 7 - xcd) >> 3;
    // A workgroup walks a contiguous piece of its XCD's tile range, i.e. the ntn column tiles of a row block one
    // after the other: the activation rows come from memory for the first of them and from L2 for the rest.  (Dealt
    // out one tile per turn, the ntn workgroups that share a row block all take its first-touch latency together.)
    const int t_lo = (int)((int64_t)x_len * slot / nslots), t_hi = (int)((int64_t)x_len * (slot + 1) / nslots);
    const int my_tiles = t_hi - t_lo;
    if (slot >= nslots || my_tiles <= 0) return;
    auto tile_of = [&](int t) { return x_start + t_lo + t; };   // t-th tile of this workgroup
    const int nk = g.K / BK;
    const int total = my_tiles * nk;

    float* bias_s = reinterpret_cast<float*>(smem + S * kStage + NC * kStg);
    for (int i = tid; i < g.N; i += kThreads) bias_s[i] = g.bias ? g.bias[i] : 0.f;
    __syncthreads();

    if (wave >= NC) {
        // ================= producers =================
        const int pw = wave - NC;
        const int st_row = lane / CPR, st_phys = lane % CPR;
        const u16* src[PW];
        int i_tile = 0, i_kt = 0, i_buf = 0, issued = 0;
        auto set_src = [&](int logical) {
            const int m0 = (logical / g.ntn) * BM;
            const int n0 = (logical % g.ntn) * BN;
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const int p = i * NP + pw;   // pieces [0, A_PIECES) are activation rows, the rest weight rows
                const int row = (p < A_PIECES ? p : p - A_PIECES) * RPP + st_row;
                const int chunk = (st_phys ^ swz<BK>(row)) * 8;
                if (p < A_PIECES) {
                    int am = m0 + row;
                    if (am > g.M - 1) am = g.M - 1;  // tail rows re-read the last row; outputs are masked
                    src[i] = g.A + (size_t)am * g.lda + chunk;
                } else {
                    src[i] = g.W + (size_t)(n0 + row) * g.ldw + chunk;
                }
            }
        };
        auto issue_one = [&]() {
            if (issued >= total) return;
            char* base = smem + i_buf * kStage;
#pragma unroll
            for (int i = 0; i < PW; ++i) glds16(src[i] + (size_t)i_kt * BK, base + (i * NP + pw) * 1024);
            ++issued;
            i_buf = i_buf + 1 == S ? 0 : i_buf + 1;
            if (++i_kt == nk) {
                i_kt = 0;
                ++i_tile;
                if (issued < total) set_src(tile_of(i_tile));
            }
        };
        // "stage j has landed": at most the pieces of the stages requested after it are outstanding
        auto wait_stage = [&](int j) {
            const int younger = issued - 1 - j;
            if constexpr (S == 2) wait_vm<0>();
            else {
                if (younger >= S - 2) wait_vm<(S - 2) * PW>();
                else if (younger == 2) wait_vm<(S > 3 ? 2 : 1) * PW>();
                else if (younger == 1) wait_vm<PW>();
                else wait_vm<0>();
            }
        };
        set_src(tile_of(0));
#pragma unroll
        for (int d = 0; d < S - 1; ++d) issue_one();
        wait_stage(0);
        __builtin_amdgcn_s_barrier();                       // B_0
        for (int step = 0; step < total; ++step) {
            asm volatile("" ::: "memory");
            issue_one();                                     // stage step + S - 1 into the slot of stage step - 1
            if (step + 1 < total) wait_stage(step + 1);
            __builtin_amdgcn_s_barrier();                   // B_{step + 1}
        }
        return;
    }

    // ================= consumers =================
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    // fragment read: 16-byte chunk fq (+ 4 per further 32 of K) of row fr (+ 16 per further fragment: same swizzle)
    int c_off[BK / 32];
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) c_off[ks] = ((4 * ks + fq) ^ swz<BK>(fr)) * 16;
    const int a_base = (wm * TM + fr) * ROWB;
    const int w_base = BM * ROWB + (wn * TN + fr) * ROWB;
    char* stg = smem + S * kStage + wave * kStg;
    int c_buf = 0;
    __builtin_amdgcn_s_barrier();                           // B_0
    for (int t = 0; t < my_tiles; ++t) {
        const int logical = tile_of(t);
        const int m0 = (logical / g.ntn) * BM;
        const int n0 = (logical % g.ntn) * BN;
        f32x4 acc[NT][MT];  // element j <-> output column n + j
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            const char* base = smem + c_buf * kStage;
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                bf16x8 af[MT], wf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + a_base + c_off[ks] + i * 16 * ROWB);
#pragma unroll
                for (int i = 0; i < NT; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(base + w_base + c_off[ks] + i * 16 * ROWB);
#ifndef EQUSS_ABL_WSNOMFMA
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
#else
#pragma unroll
                for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
                for (int i = 0; i < NT; ++i) asm volatile("" ::"v"(wf[i]));
#endif
            }
            c_buf = c_buf + 1 == S ? 0 : c_buf + 1;
            if (kt + 1 < nk) {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();               // (the fragment reads above have returned: the MFMAs used them)
                asm volatile("" ::: "memory");
            }
        }
        // ---- epilogue: bias (+ GELU / ReLU), bf16, re-shaped through the wave's own LDS rows into 16-byte row pieces.
        // Its LDS traffic is inline asm: see equss::lds_store_b64.
        const bool full = m0 + BM <= g.M;
#ifdef EQUSS_ABL_WSNOEPI   // (developer ablation builds, tools/build_ablations.sh: wrong results by design)
        if (g.M > 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(acc[nt][mt]));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            continue;
        }
#endif
        equss::lds_u32x4 b4u[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b4u[nt] = equss::lds_load_b128(bias_s + n0 + wn * TN + nt * 16 + fq * 4);
        equss::lds_wait(b4u);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 b4 = __builtin_bit_cast(f32x4, b4u[nt]);
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
                equss::lds_store_b64(stg + fr * ROWSTRIDE + (nt * 16 + fq * 4) * 2, equss::pack_bf16(v.x, v.y), equss::pack_bf16(v.z, v.w));
            }
            equss::lds_u32x4 d[PER_LANE];   // same-wave LDS accesses execute in order
#pragma unroll
            for (int j = 0; j < PER_LANE; ++j) {
                const int idx = j * 64 + lane;
                d[j] = equss::lds_load_b128(stg + (idx / OCH) * ROWSTRIDE + (idx % OCH) * 16);
            }
            equss::lds_wait(d);
#pragma unroll
            for (int j = 0; j < PER_LANE; ++j) {
                const int idx = j * 64 + lane;
                const int m = m0 + wm * TM + mt * 16 + idx / OCH;
#ifndef EQUSS_ABL_WSNOSTORE
                if (full || m < g.M)
                    *reinterpret_cast<equss::lds_u32x4*>(g.out + (size_t)m * g.ldo + n0 + wn * TN + (idx % OCH) * 8) = d[j];
#else
                asm volatile("" ::"v"(d[j]), "v"(m));
#endif
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // the barrier that ends the tile's last K step
        asm volatile("" ::: "memory");
    }
}

int g_ws_bk = 64;   // developer knob: K step of the ring (32 or 64)

template <int EPI, int S, int BK>
int launch_geo(const WsArgs& g, hipStream_t s) {
    constexpr int kStage = Geo<BK>::kStage;
    const int ntiles = ((g.M + BM - 1) / BM) * g.ntn;
    const int grid = ntiles < 256 ? ntiles : 256;
    if (g.N > 1152) return equss::fail(EQUSS_E_UNSUPPORTED, "gemm_ws: N=%d exceeds the bias staging area", g.N);
    constexpr int lds = S * kStage + NC * kStg + 1152 * 4;
    static_assert(lds <= 163840, "LDS budget");
    auto kern = gemm_ws_kernel<EPI, S, BK>;
    if (!equss::lds_opt_in(reinterpret_cast<const void*>(kern), lds))
        return equss::fail(EQUSS_E_UNSUPPORTED, "gemm_ws: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, s, g);
    return equss::check_launch("gemm_ws_kernel");
}
template <int EPI>
int launch(const WsArgs& g, hipStream_t s) {
    return g_ws_bk == 64 && g.K % 64 == 0 ? launch_geo<EPI, 2, 64>(g, s) : launch_geo<EPI, 4, 32>(g, s);
}

}  // namespace

extern "C" void equss_dev_ws_bk(int bk) { g_ws_bk = bk == 32 ? 32 : 64; }  // tuning knob, not part of the ABI

namespace equss {

bool gemm_ws_supports(int M, int N, int K, int epilogue) {
    return N % BN == 0 && N <= 1152 && K % 32 == 0 && M >= 16384 &&
           (epilogue == EQUSS_EPI_BF16 || epilogue == EQUSS_EPI_GELU_BF16 || epilogue == EQUSS_EPI_RELU_BF16);
}

int gemm_ws_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo, int M, int N,
                 int K, int epilogue, hipStream_t s) {
    WsArgs g;
    g.A = reinterpret_cast<const u16*>(A); g.lda = lda;
    g.W = reinterpret_cast<const u16*>(W); g.ldw = ldw;
    g.bias = bias; g.out = reinterpret_cast<u16*>(out); g.ldo = ldo; g.M = M; g.N = N; g.K = K; g.ntn = N / BN;
    switch (epilogue) {
        case EQUSS_EPI_BF16: return launch<EQUSS_EPI_BF16>(g, s);
        case EQUSS_EPI_GELU_BF16: return launch<EQUSS_EPI_GELU_BF16>(g, s);
        case EQUSS_EPI_RELU_BF16: return launch<EQUSS_EPI_RELU_BF16>(g, s);
        default: return fail(EQUSS_E_UNSUPPORTED, "gemm_ws: unsupported epilogue %d", epilogue);
    }
}

}  // namespace equss
