// Large-tile, persistent, deep-ring variant of C = A * W^T + bias (bf16 MFMA, fp32 accumulate,
// fused store-only epilogue).
//
// Why a second GEMM kernel.  PMC counters on the 128 x 128 two-stage kernel show its waves parked
// 57-59 % of the time at the vmcnt/barrier that ends every K step while the matrix pipe is busy
// 22 % of the time: a CU with one 32-64 KiB stage in flight sustains only ~26 GB/s of L2 -> LDS
// traffic (bytes in flight / latency), far below what the MFMAs can eat.  Two levers, both here:
//   * more flop per byte: tiles of 128 x 384 (N a multiple of 384: qkv, proj, fc1, fc2, patch
//     embed, head GEMM 1) or 256 x 256 (N a multiple of 256: head GEMM 2), 96-128 flop/B instead
//     of 64;
//   * more bytes in flight: an S-stage LDS ring filled by global_load_lds_dwordx4 S-1 K steps
//     ahead and retired with COUNTED s_waitcnt vmcnt(N) + a raw s_barrier, so loads stay in flight
//     across barriers and across tiles (the ring never drains: workgroups are persistent and the
//     issue cursor runs into the next tile while the compute cursor finishes the current one).
// One workgroup = 8 waves (two per SIMD) = one tile; one workgroup per CU; tiles are walked in an
// XCD-contiguous order so neighbours in time share operand panels in that XCD's L2.
// The epilogue only stores (the residual add lives in the LayerNorm that follows), which keeps the
// in-order vmcnt queue free of epilogue loads that would otherwise drain the ring.
// Fragment layout, chunk swizzle and the C^T trick are those of gemm_bf16.hip.
#include "equss_common.h"

namespace {

struct Gemm8Args {
    const u16* A; int64_t lda;
    const u16* W; int64_t ldw;
    const float* bias;
    void* out; int64_t ldo;
    int M, N, K, ntn;
    // residual + LayerNorm epilogue (EPI_RESID_LN): x (M, N) fp32 is updated in place, ln receives
    // LayerNorm(x) in bf16 (or is skipped when null)
    float* x; int64_t ldx;
    const float* gamma; const float* beta;
    u16* ln; int64_t ld_ln;
    float eps;
    const float* pos; int group;   // EPI_PATCH_LN
};

constexpr int EPI_RESID_LN = 100;  // internal epilogue id: x += acc + bias; ln = LayerNorm(x)
// internal epilogue id: token assembly of the patch embedding (vision_transformer.py:205-209) with the first
// LayerNorm: output row m (patch m % group of image m / group) goes to token row m + m / group + 1,
// x = pos[1 + m % group] + bias + acc, ln = LayerNorm(x)
constexpr int EPI_PATCH_LN = 101;
constexpr bool row_ln(int e) { return e == EPI_RESID_LN || e == EPI_PATCH_LN; }

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const void __attribute__((address_space(1)))*)gsrc,
        (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// Wait until at most `stages * PW (+ NST)` of this wave's vector-memory operations are outstanding.
// Operations retire in issue order, so "the loads of ring stage j have landed" is exactly "no more
// operations are outstanding than were issued after them": `stages` younger ring stages of PW loads
// each, plus the NST stores of an epilogue when that epilogue was issued after stage j's loads.
template <int PW, int NST>
__device__ __forceinline__ void wait_younger(int stages, bool stores_younger) {
#define EQUSS_W(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
    if (stores_younger) {
        switch (stages) {
            case 0: EQUSS_W(NST); break;
            case 1: EQUSS_W(PW + NST); break;
            case 2: EQUSS_W(2 * PW + NST); break;
            case 3: EQUSS_W(3 * PW + NST); break;
            default: EQUSS_W(4 * PW + NST); break;
        }
    } else {
        switch (stages) {
            case 0: EQUSS_W(0); break;
            case 1: EQUSS_W(PW); break;
            case 2: EQUSS_W(2 * PW); break;
            case 3: EQUSS_W(3 * PW); break;
            default: EQUSS_W(4 * PW); break;
        }
    }
#undef EQUSS_W
}

template <int EPI, int BM, int BN, int WM, int WN, int BK, int S>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && BM * BN >= 128 * 384) ? 1 : 2) void gemm8_kernel(const Gemm8Args g) {
    constexpr int NW = WM * WN;                    // waves per workgroup: 8 (one workgroup per CU) or 4 (two)
    constexpr int kThreads = NW * 64;
    static_assert(NW == 8 || NW == 4, "four or eight waves");
    static_assert(S >= 2 && S <= 6, "ring depth");
    constexpr int TM = BM / WM, TN = BN / WN;      // per-wave output tile
    constexpr int MT = TM / 16, NT = TN / 16;      // 16 x 16 MFMA tiles per wave
    constexpr int ROWB = BK * 2;                   // bytes per tile row
    constexpr int CPR = BK / 8;                    // 16-byte chunks per row
    constexpr int RPP = 1024 / ROWB;               // rows per 1-KiB LDS-DMA piece
    constexpr int KS = BK / 32;                    // MFMA k sub-steps per K step
    constexpr bool PIPE = KS == 1 && S >= 3;       // fragment reads pipelined one ring stage ahead
    constexpr int kStageBytes = (BM + BN) * ROWB;
    constexpr int A_PIECES = BM / RPP;
    constexpr int PW = (BM + BN) / RPP / NW;       // pieces per wave per stage
    // bf16 epilogue geometry (see the epilogue): rows per pass and 16-byte stores per lane per tile
    constexpr int ROWSTRIDE = TN * 2 + 16;
    constexpr int RP = (kStageBytes / NW >= 32 * ROWSTRIDE && TM >= 32) ? 32 : 16;
    constexpr int OCH = TN * 2 / 16;
    constexpr int PER_LANE = RP * OCH / 64;
    constexpr int NST = (EPI == EQUSS_EPI_F32 || row_ln(EPI) || EPI == EQUSS_EPI_RESID_F32) ? 0 : PER_LANE * (TM / RP);
    static_assert((S - 1) * PW + NST <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // 16-B chunk swizzle of a tile row.  128-B rows: (row >> 1) & 7.  64-B rows: a ds_read_b128 is served in the
    // lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32), so one group reads rows r, r + 12 at chunk fq
    // and rows r + 4, r + 8 at chunk fq ^ 1 of every bank quarter; the permutation 0,2,3,1 of (row >> 2) & 3
    // puts those four on four different 16-B columns (the identity left them two and two on the same banks).
    auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : ((0x78 >> (2 * ((row >> 2) & 3))) & 3); };

    const int ntiles = ((g.M + BM - 1) / BM) * g.ntn;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int x_start = xcd * q + (xcd < r ? xcd : r);
    const int x_len = q + (xcd < r ? 1 : 0);
    const int nslots = (gridDim.x + 7 - xcd) >> 3;
    if (slot >= x_len) return;
    const int my_tiles = (x_len - slot + nslots - 1) / nslots;
    const int nk = g.K / BK;
    const int total = my_tiles * nk;

    const int fr = lane & 15, fq = lane >> 4;
    int c_off[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) c_off[ks] = ((4 * ks + fq) ^ swz(fr)) * 16;
    const int a_base = (wm * TM + fr) * ROWB;
    const int w_base = BM * ROWB + (wn * TN + fr) * ROWB;

    // ---- issue cursor: which (tile, K step) is staged next, and into which ring slot -------------
    const int st_row = lane / CPR, st_phys = lane % CPR;
    // per-lane source of piece i as a 32-bit element offset from A or W (which of the two is a property of i:
    // A_PIECES is a multiple of NW): half the registers of 64-bit pointers
    static_assert(A_PIECES % NW == 0, "a wave's piece i is an activation piece for every wave or for none");
    uint32_t src[PW];
    int i_tile = slot, i_kt = 0, i_buf = 0, issued = 0;
    auto set_src = [&](int logical) {
        const int m0 = (logical / g.ntn) * BM;
        const int n0 = (logical % g.ntn) * BN;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int p = i * NW + wave;  // pieces [0, A_PIECES) are activation rows, the rest weight rows
            const int row = (i * NW < A_PIECES ? p : p - A_PIECES) * RPP + st_row;
            const int chunk = (st_phys ^ swz(row)) * 8;
            if (i * NW < A_PIECES) {
                int am = m0 + row;
                if (am > g.M - 1) am = g.M - 1;  // tail rows re-read the last row; outputs are masked
                src[i] = (uint32_t)am * (uint32_t)g.lda + chunk;
            } else {
                src[i] = (uint32_t)(n0 + row) * (uint32_t)g.ldw + chunk;
            }
        }
    };
    auto issue_one = [&]() {
        if (issued >= total) return;
        char* base = smem + i_buf * kStageBytes;
#pragma unroll
        for (int i = 0; i < PW; ++i)
            glds16((i * NW < A_PIECES ? g.A : g.W) + src[i] + (uint32_t)(i_kt * BK), base + (i * NW + wave) * 1024);
        ++issued;
        i_buf = i_buf + 1 == S ? 0 : i_buf + 1;
        if (++i_kt == nk) {
            i_kt = 0;
            i_tile += nslots;
            if (issued < total) set_src(x_start + i_tile);
        }
    };
    // bias -> LDS once, before the ring starts: an ordinary global load inside the loop would make the
    // compiler wait for it, and with it (in-order retirement) for every prefetch issued before it.
    float* bias_s = reinterpret_cast<float*>(smem + S * kStageBytes);
    for (int i = tid; i < g.N; i += kThreads) bias_s[i] = g.bias ? g.bias[i] : 0.f;
    float* gamma_s = bias_s + g.N;       // EPI_RESID_LN only (N == BN there)
    float* beta_s = gamma_s + BN;
    float2* red_s = reinterpret_cast<float2*>(beta_s + BN);  // BM rows x WN partial (sum, sum of squares)
    if constexpr (row_ln(EPI)) {
        for (int i = tid; i < BN; i += kThreads) {
            gamma_s[i] = g.ln ? g.gamma[i] : 1.f;
            beta_s[i] = g.ln ? g.beta[i] : 0.f;
        }
    }
    __syncthreads();
    set_src(x_start + i_tile);
#pragma unroll
    for (int d = 0; d < S - 1; ++d) issue_one();

    // ---- compute cursor ----------------------------------------------------------------------------
    int c_tile = slot, c_buf = 0, step = 0;
    int epi_mark = 0;  // ring stages [0, epi_mark) were issued before the latest exact-count epilogue
    typedef int frag_t __attribute__((ext_vector_type(4)));  // 8 bf16 kept as 4 dwords so copies stay whole
    frag_t af_c[MT], wf_c[NT], af_n[MT], wf_n[NT];  // PIPE: two alternating fragment sets
    if constexpr (PIPE) {
        wait_younger<PW, NST>(issued - 1, false);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < MT; ++i) af_c[i] = *reinterpret_cast<const frag_t*>(smem + a_base + i * 16 * ROWB + c_off[0]);
#pragma unroll
        for (int i = 0; i < NT; ++i) wf_c[i] = *reinterpret_cast<const frag_t*>(smem + w_base + i * 16 * ROWB + c_off[0]);
    }
    for (int t = 0; t < my_tiles; ++t, c_tile += nslots) {
        const int logical = x_start + c_tile;
        const int m0 = (logical / g.ntn) * BM;
        const int n0 = (logical % g.ntn) * BN;
        f32x4 acc[NT][MT];  // element j <-> output column n + j
        if constexpr (row_ln(EPI) || EPI == EQUSS_EPI_RESID_F32) {
            // Start the accumulators from x + bias: the residual tile is fetched while the first K step
            // is being waited for, straight into the accumulator registers, and the epilogue is left
            // with stores only.  (EPI_RESID_LN: x = g.x, one column tile; EPI_RESID_F32: x = out;
            // EPI_PATCH_LN: the position embedding of the row's token takes the place of x.)
            const float* xs = EPI == EPI_PATCH_LN ? g.pos : EPI == EPI_RESID_LN ? g.x : reinterpret_cast<const float*>(g.out);
            const int64_t ldxs = EPI == EPI_PATCH_LN ? g.N : EPI == EPI_RESID_LN ? g.ldx : g.ldo;
            size_t src_row[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = m0 + wm * TM + mt * 16 + fr;
                src_row[mt] = EPI == EPI_PATCH_LN ? (size_t)(1 + m % g.group) : (size_t)m;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + wn * TN + nt * 16 + fq * 4;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + n);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = m0 + wm * TM + mt * 16 + fr;
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
                    if (m < g.M) o = *reinterpret_cast<const f32x4*>(xs + src_row[mt] * ldxs + n);
                    o.x += b4.x; o.y += b4.y; o.z += b4.z; o.w += b4.w;
                    acc[nt][mt] = o;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (PIPE) {
            // Software-pipelined form (BK = 32, one MFMA sub-step per ring stage): the fragments of stage
            // step + 1 are read from LDS while the MFMAs of stage `step` run on fragments already in
            // registers, across tile boundaries too.  Ring slot (step - 1) % S, refilled below, held the
            // stage whose fragments every wave finished reading before it issued the MFMAs of its previous
            // iteration, i.e. before it arrived at this barrier.
            // One ring stage: multiply (ac, wc) while (an, wn_) is being read; the caller alternates the
            // two register sets, so no copy is needed (nk is even: K is a multiple of 64).
            auto stage = [&](frag_t (&ac)[MT], frag_t (&wc)[NT], frag_t (&an)[MT], frag_t (&wn_)[NT]) {
                if (step + 1 < total) {
                    wait_younger<PW, NST>(issued - step - 2, step + 1 < epi_mark);
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    issue_one();
                }
                // (after the very last stage this reads a slot nobody uses again; the values are dropped)
                const int nb = c_buf + 1 == S ? 0 : c_buf + 1;
                const char* base = smem + nb * kStageBytes;
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    an[i] = *reinterpret_cast<const frag_t*>(base + a_base + i * 16 * ROWB + c_off[0]);
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    wn_[i] = *reinterpret_cast<const frag_t*>(base + w_base + i * 16 * ROWB + c_off[0]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wc[nt]),
                                                                              __builtin_bit_cast(bf16x8, ac[mt]),
                                                                              acc[nt][mt], 0, 0, 0);
                asm volatile("" ::: "memory");
                c_buf = c_buf + 1 == S ? 0 : c_buf + 1;
                ++step;
            };
            for (int kt = 0; kt < nk; kt += 2) {
                stage(af_c, wf_c, af_n, wf_n);
                stage(af_n, wf_n, af_c, wf_c);
            }
        } else {
        for (int kt = 0; kt < nk; ++kt, ++step) {
            // this wave's pieces of stage `step` have landed once at most (issued - step - 1) younger
            // stages are outstanding (operations retire in order; epilogue stores in the queue only make
            // the wait stricter); the barrier then covers the other waves' pieces, and tells everyone
            // that ring slot (step - 1) % S is no longer being read.
            wait_younger<PW, NST>(issued - step - 1, step < epi_mark);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_one();
            const char* base = smem + c_buf * kStageBytes;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 af[MT], wf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(base + a_base + i * 16 * ROWB + c_off[ks]);
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    wf[i] = *reinterpret_cast<const bf16x8*>(base + w_base + i * 16 * ROWB + c_off[ks]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
            c_buf = c_buf + 1 == S ? 0 : c_buf + 1;
        }
        }
        // ---- epilogue (stores only) ---------------------------------------------------------------------
        if constexpr (row_ln(EPI)) {
            // The accumulators already hold x + bias + A W^T: write x back (16 bytes per lane), then LayerNorm
            // of the updated rows: a tile spans whole rows (BN == N), a row's 4 x NT x 4 values sit in the 4
            // waves of one wm group, so the statistics are an in-lane sum, two shuffles and one small LDS
            // exchange; the normalised rows leave as bf16 through the same wide-store path as below.
            static_assert(!row_ln(EPI) || WN == 4, "row statistics assume four waves per row");
            // row of the output buffers (EPI_PATCH_LN skips one CLS row per image)
            auto out_row = [&](int m) { return EPI == EPI_PATCH_LN ? (size_t)(m + m / g.group + 1) : (size_t)m; };
            epi_mark = 0;
            float s1[MT], s2[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                s1[mt] = 0.f; s2[mt] = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 v = acc[nt][mt];
                    // (explicit fma placement: layernorm_rowtile_kernel in vit_ops.hip repeats these sums
                    // in the same order so that both routes give the same bits)
                    s1[mt] += (v.x + v.y) + (v.z + v.w);
                    s2[mt] += fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);
                }
                s1[mt] += __shfl_xor(s1[mt], 16, 64); s2[mt] += __shfl_xor(s2[mt], 16, 64);
                s1[mt] += __shfl_xor(s1[mt], 32, 64); s2[mt] += __shfl_xor(s2[mt], 32, 64);
                if (fq == 0) equss::lds_store_f32x2(red_s + ((wm * TM + mt * 16 + fr) * WN + wn), s1[mt], s2[mt]);
            }
            if (g.ln) {
                const int free_slot = c_buf == 0 ? S - 1 : c_buf - 1;
                char* stg = smem + free_slot * kStageBytes + wave * (kStageBytes / NW);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();  // partial sums visible; ring slot no longer read
                asm volatile("" ::: "memory");
#pragma unroll
                for (int pass = 0; pass < TM / RP; ++pass) {
#pragma unroll
                    for (int ml = 0; ml < RP / 16; ++ml) {
                        const int mt = pass * (RP / 16) + ml;
                        const float2* pr = red_s + (wm * TM + mt * 16 + fr) * WN;
                        const float2 p0 = pr[0], p1 = pr[1], p2 = pr[2], p3 = pr[3];
                        const float mean = ((p0.x + p1.x) + (p2.x + p3.x)) * (1.0f / BN);
                        const float ex2 = ((p0.y + p1.y) + (p2.y + p3.y)) * (1.0f / BN);
                        const float rstd = 1.0f / sqrtf(fmaxf(fmaf(-mean, mean, ex2), 0.f) + g.eps);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const int n = wn * TN + nt * 16 + fq * 4;
                            const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma_s + n);
                            const f32x4 be = *reinterpret_cast<const f32x4*>(beta_s + n);
                            const f32x4 v = acc[nt][mt];
                            uint2 pk;
                            pk.x = equss::pack_bf16(fmaf((v.x - mean) * rstd, ga.x, be.x), fmaf((v.y - mean) * rstd, ga.y, be.y));
                            pk.y = equss::pack_bf16(fmaf((v.z - mean) * rstd, ga.z, be.z), fmaf((v.w - mean) * rstd, ga.w, be.w));
                            equss::lds_store_b64(stg + (ml * 16 + fr) * ROWSTRIDE + (nt * 16 + fq * 4) * 2, pk.x, pk.y);
                        }
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int j = 0; j < PER_LANE; ++j) {
                        const int idx = j * 64 + lane;
                        const int row = idx / OCH, ch = idx % OCH;
                        const uint4 d = *reinterpret_cast<const uint4*>(stg + row * ROWSTRIDE + ch * 16);
                        const int m = m0 + wm * TM + pass * RP + row;
                        if (m < g.M) *reinterpret_cast<uint4*>(g.ln + out_row(m) * g.ld_ln + wn * TN + ch * 8) = d;
                    }
                    asm volatile("" ::: "memory");
                }
            }
            // x goes back last: the first LDS load the compiler emits after a global_load_lds carries s_waitcnt vmcnt(0),
            // which would otherwise sit out the round trip of these stores
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = m0 + wm * TM + mt * 16 + fr;
                const size_t mo = out_row(m);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (m < g.M) *reinterpret_cast<f32x4*>(g.x + mo * g.ldx + wn * TN + nt * 16 + fq * 4) = acc[nt][mt];
            }
        } else if constexpr (EPI == EQUSS_EPI_F32 || EPI == EQUSS_EPI_RESID_F32) {
            // lane owns out[m][n .. n+3] as one 16-byte store (EPI_RESID_F32: bias and the old value are already in)
            // (bias through equss::lds_load_b128: an LDS load emitted by the compiler would wait for the ring stages in flight)
            equss::lds_u32x4 b4u[NT];
            if constexpr (EPI == EQUSS_EPI_F32) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b4u[nt] = equss::lds_load_b128(bias_s + n0 + wn * TN + nt * 16 + fq * 4);
                equss::lds_wait(b4u);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + wn * TN + nt * 16 + fq * 4;
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (EPI == EQUSS_EPI_F32) b4 = __builtin_bit_cast(f32x4, b4u[nt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = m0 + wm * TM + mt * 16 + fr;
                    if (m >= g.M) continue;
                    f32x4 v = acc[nt][mt];
                    if constexpr (EPI == EQUSS_EPI_F32) { v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
                    // (EPI_F32 is the head's `code` output, 420 MB written once: the non-temporal hint is worth 6 % of that launch)
                    if constexpr (EPI == EQUSS_EPI_F32)
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + (size_t)m * g.ldo + n));
                    else
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + (size_t)m * g.ldo + n) = v;
                }
            }
        } else {
            // bf16 output: a lane's fragment is only 8 bytes wide, and 32-byte row segments make the write
            // path crawl (measured ~9 GB/s per CU).  Re-shape each wave's sub-tile through the ring slot
            // that was just consumed (free until the next K step is staged into it): 8-byte fragment
            // writes in, 16-byte row-contiguous reads out, then 16-byte stores covering whole rows.
            // ROWSTRIDE = padded LDS row, RP = rows per pass, OCH = 16-byte chunks per output row.
            static_assert(RP * OCH % 64 == 0, "epilogue chunking");
            // A full tile issues exactly NST stores per wave, which lets the waits that follow leave them
            // in flight; the ragged last row-tile masks stores, so it falls back to the stricter wait.
            const bool full = m0 + BM <= g.M;
            epi_mark = full ? issued : 0;
            const int free_slot = c_buf == 0 ? S - 1 : c_buf - 1;
            char* stg = smem + free_slot * kStageBytes + wave * (kStageBytes / NW);
            __builtin_amdgcn_s_barrier();  // every wave is done reading that slot
            asm volatile("" ::: "memory");
            // (LDS traffic of this block goes through equss::lds_load_b128 / lds_store_b64: emitted by the compiler,
            // the first of them would carry s_waitcnt vmcnt(0) and drain the S - 1 ring stages in flight)
            equss::lds_u32x4 b4u[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b4u[nt] = equss::lds_load_b128(bias_s + n0 + wn * TN + nt * 16 + fq * 4);
            equss::lds_wait(b4u);
            f32x4 b4[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b4[nt] = __builtin_bit_cast(f32x4, b4u[nt]);
#pragma unroll
            for (int pass = 0; pass < TM / RP; ++pass) {
#pragma unroll
                for (int ml = 0; ml < RP / 16; ++ml) {
                    const int mt = pass * (RP / 16) + ml;
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
                        equss::lds_store_b64(stg + (ml * 16 + fr) * ROWSTRIDE + (nt * 16 + fq * 4) * 2, equss::pack_bf16(v.x, v.y),
                                             equss::pack_bf16(v.z, v.w));
                    }
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
                    const int row = idx / OCH, ch = idx % OCH;
                    const int m = m0 + wm * TM + pass * RP + row;
                    if (full || m < g.M)
                        *reinterpret_cast<equss::lds_u32x4*>(reinterpret_cast<u16*>(g.out) + (size_t)m * g.ldo + n0 + wn * TN + ch * 8) = d[j];
                }
                asm volatile("" ::: "memory");
            }
        }
    }
}

int g_stages = 2;
int g_prefer256 = 1;  // 256 x 256 tiles when N allows (131 flop/B instead of 96)
int g_cu_share = 256;   // CUs one launch may occupy (developer knob: two streams sharing the device)

template <int EPI, int BM, int BN, int WM, int WN, int BK, int S>
int launch_inst(const Gemm8Args& g, hipStream_t s) {
    const int ntiles = ((g.M + BM - 1) / BM) * g.ntn;
    constexpr int kThreads = WM * WN * 64;
    const int resident = (WM * WN == 8 ? 1 : 2) * g_cu_share;  // one 8-wave or two 4-wave workgroups per CU
    const int grid = ntiles < resident ? ntiles : resident;
    constexpr int ring = S * (BM + BN) * BK * 2;
    if (g.N > 4096) return equss::fail(EQUSS_E_UNSUPPORTED, "gemm8: N=%d exceeds the bias staging area", g.N);
    // (the kernel addresses its operand pieces by 32-bit element offsets: every route into it is checked here)
    if ((int64_t)g.M * g.lda >= (1LL << 32) || (int64_t)g.N * g.ldw >= (1LL << 32))
        return equss::fail(EQUSS_E_UNSUPPORTED, "gemm8: M * lda = %lld or N * ldw = %lld elements exceed 32-bit operand offsets",
                           (long long)g.M * g.lda, (long long)g.N * g.ldw);
    constexpr int extra = row_ln(EPI) ? (3 * BN * 4 + BM * WN * 8) : 16384;  // bias (+ gamma, beta, partial sums)
    static_assert(ring + extra <= 163840, "LDS budget");
    const int lds = ring + (row_ln(EPI) ? extra : g.N * 4);
    auto kern = gemm8_kernel<EPI, BM, BN, WM, WN, BK, S>;
    if (!equss::lds_opt_in(reinterpret_cast<const void*>(kern), ring + extra))
        return equss::fail(EQUSS_E_UNSUPPORTED, "gemm8: cannot reserve %d bytes of LDS", ring + extra);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, s, g);
    return equss::check_launch("gemm8_kernel");
}

template <int EPI, int BM, int BN, int WM, int WN>
int launch_shape(const Gemm8Args& g, hipStream_t s) {
    switch (g_stages) {
        case 2: return launch_inst<EPI, BM, BN, WM, WN, 64, 2>(g, s);
        case 3: return launch_inst<EPI, BM, BN, WM, WN, 32, 3>(g, s);
        default: return launch_inst<EPI, BM, BN, WM, WN, 32, 4>(g, s);
    }
}

template <int EPI>
int launch(Gemm8Args& g, hipStream_t s) {
    if (g_stages == 5 && g.N % 384 == 0) {
        g.ntn = g.N / 384;
        return launch_inst<EPI, 128, 384, 1, 4, 32, 4>(g, s);
    }
    if (g_stages == 6 && g.N % 192 == 0) {   // two 4-wave workgroups per CU, 128 x 192 tiles
        g.ntn = g.N / 192;
        return launch_inst<EPI, 128, 192, 2, 2, 32, 3>(g, s);
    }
    if (g_stages == 7 && g.N % 128 == 0) {   // two 4-wave workgroups per CU, 256 x 128 tiles
        g.ntn = g.N / 128;
        return launch_inst<EPI, 256, 128, 2, 2, 32, 3>(g, s);
    }
    if (g.N % 384 == 0 && !(g_prefer256 && g.N % 256 == 0)) {
        g.ntn = g.N / 384;
        // short-K, store-heavy shapes (qkv) run a little faster on the 4-deep ring of 32-wide stages
        // with fragment reads pipelined one stage ahead
        if (EPI == EQUSS_EPI_BF16 && g.K <= 512 && g.N > 384 && g_stages == 2)
            return launch_inst<EPI, 128, 384, 2, 4, 32, 4>(g, s);
        return launch_shape<EPI, 128, 384, 2, 4>(g, s);
    }
    g.ntn = g.N / 256;
    return launch_shape<EPI, 256, 256, 2, 4>(g, s);
}

}  // namespace

namespace equss {

bool gemm8_supports(int N, int K, int epilogue) {
    return (N % 384 == 0 || N % 256 == 0) && K % 64 == 0;
}

int gemm8_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo,
               int M, int N, int K, int epilogue, hipStream_t s) {
    Gemm8Args g;
    g.A = reinterpret_cast<const u16*>(A); g.lda = lda;
    g.W = reinterpret_cast<const u16*>(W); g.ldw = ldw;
    g.bias = bias; g.out = out; g.ldo = ldo; g.M = M; g.N = N; g.K = K; g.ntn = 0;
    g.x = nullptr; g.ldx = 0; g.gamma = nullptr; g.beta = nullptr; g.ln = nullptr; g.ld_ln = 0; g.eps = 0.f;
    g.pos = nullptr; g.group = 1;
    switch (epilogue) {
        case EQUSS_EPI_BF16: return launch<EQUSS_EPI_BF16>(g, s);
        case EQUSS_EPI_GELU_BF16: return launch<EQUSS_EPI_GELU_BF16>(g, s);
        case EQUSS_EPI_RELU_BF16: return launch<EQUSS_EPI_RELU_BF16>(g, s);
        case EQUSS_EPI_F32: return launch<EQUSS_EPI_F32>(g, s);
        case EQUSS_EPI_RESID_F32: return launch<EQUSS_EPI_RESID_F32>(g, s);
        default: return fail(EQUSS_E_UNSUPPORTED, "gemm8: unsupported epilogue %d", epilogue);
    }
}

// x (M, N) fp32 += A W^T + bias, then ln = LayerNorm(x) * gamma + beta in bf16 (ln may be null).
// Needs N == 384 so that one tile holds whole rows.
bool gemm8_resid_ln_supports(int N, int K) { return N == 384 && K % 64 == 0; }

int gemm8_resid_ln(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, float* x, int64_t ldx,
                   const float* gamma, const float* beta, void* ln, int64_t ld_ln, int M, int N, int K, float eps,
                   hipStream_t s) {
    Gemm8Args g;
    g.A = reinterpret_cast<const u16*>(A); g.lda = lda;
    g.W = reinterpret_cast<const u16*>(W); g.ldw = ldw;
    g.bias = bias; g.out = nullptr; g.ldo = 0; g.M = M; g.N = N; g.K = K; g.ntn = 1;
    g.x = x; g.ldx = ldx; g.gamma = gamma; g.beta = beta; g.ln = reinterpret_cast<u16*>(ln); g.ld_ln = ld_ln; g.eps = eps;
    g.pos = nullptr; g.group = 1;
    return launch_inst<EPI_RESID_LN, 128, 384, 2, 4, 64, 2>(g, s);
}

// Patch embedding as token rows (vision_transformer.py:128-133,:205-209) with the first LayerNorm:
//   x[b, 1 + t] = pos[1 + t] + bias + cols[b * n + t] W^T ,  ln[b, 1 + t] = LayerNorm(x[b, 1 + t])   (ln may be null)
// for the M = B * n patch rows; x and ln are (B * (n + 1), 384) token-major, the CLS rows are not touched.
int gemm8_patch_ln(const void* cols, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* pos, float* x,
                   const float* gamma, const float* beta, void* ln, int M, int N, int K, int n, float eps, hipStream_t s) {
    Gemm8Args g;
    g.A = reinterpret_cast<const u16*>(cols); g.lda = lda;
    g.W = reinterpret_cast<const u16*>(W); g.ldw = ldw;
    g.bias = bias; g.out = nullptr; g.ldo = 0; g.M = M; g.N = N; g.K = K; g.ntn = 1;
    g.x = x; g.ldx = N; g.gamma = gamma; g.beta = beta; g.ln = reinterpret_cast<u16*>(ln); g.ld_ln = N; g.eps = eps;
    g.pos = pos; g.group = n;
    return launch_inst<EPI_PATCH_LN, 128, 384, 2, 4, 64, 2>(g, s);
}

}  // namespace equss

extern "C" void equss_dev_gemm_cu_share(int n) { g_cu_share = n > 0 && n <= 256 ? n : 256; }
extern "C" void equss_dev_gemm_stages(int s) { g_stages = s & 7; g_prefer256 = (s >> 3) & 1; }  // tuning knob, not part of the ABI
