// The rest of a ViT-S block after attention (vision_transformer.py:110-111 with Mlp.forward :58-64) with the weights
// streamed straight into registers:   x += att Wp^T + bp ;  a = LayerNorm2(x) ;  x += fc2(gelu(fc1(a))) ;  ln = LayerNorm(x)
// Same arithmetic, operand rounding and fp32 operation order per element as block_ws_kernel / mlp_kernel<true, *>, so
// the same bits; what differs is where the operands live and who waits for whom.
//
//  * One workgroup per CU, eight waves, a tile of up to 112 rows (NG <= 7 groups of 16).  Every wave owns 48 of the
//    384 output columns (projection, fc2) and 32 of the 256 hidden units of a chunk (fc1) for ALL rows of the tile.
//  * Activations stay in LDS in MFMA-fragment order (1 KiB per 16 rows x 32 k, lane l at byte 16 l: conflict free): the
//    tile's attention rows, overwritten by its LayerNorm2 rows (84 KiB), and the hidden chunk (56 KiB).  A fragment
//    read feeds three (projection, fc2) or two (fc1) MFMAs of the reading wave.
//  * Weights never touch LDS.  pack_rs_weights lays every wave's operand fragments out in the order the wave consumes
//    them (Wp, then per hidden chunk W1 and W2), so a wave streams 324 KiB per tile as consecutive 1-KiB loads, four
//    groups of three fragments ahead of their use, with compiler-counted vmcnt waits.  No weight element is loaded by
//    two waves of a workgroup, no ring, no per-stage barrier: two workgroup barriers per hidden chunk (hidden tile
//    written / hidden tile free) and four around the two LayerNorms are all there is.
//  * The 16-row groups of the launch are dealt to the workgroups in contiguous, equally long runs, and a run is cut
//    into tiles of equal height (5, 6 or 7 groups: the tile body is instantiated per height), so no round of tiles is
//    left partly filled.
//  * LayerNorm statistics keep the summation order of the 96-column waves of the other kernels: the even wave of a pair
//    sums its three column groups and hands the running sums to the odd wave through LDS, which continues the chain.
#include "equss_common.h"
#include <type_traits>

namespace {

struct RsArgs {
    const u16* A0; int64_t lda0;                       // attention output (M, 384) bf16
    const u16* Wimg;                                   // per-wave weight streams (pack_rs_weights)
    const float* bp; const float* gamma2; const float* beta2;
    const float* b1; const float* b2;
    float* x; int64_t ldx;                             // residual stream, updated in place
    const float* gamma; const float* beta;
    u16* ln; int64_t ld_ln;                            // LayerNorm(x) for the next consumer (may be null)
    float eps;
    int M;
    int stagger;
};

constexpr int C = 384, HID = 1536, HC = 256, NCH = HID / HC;
constexpr int NW = 8, WN = C / NW, NT = WN / 16, NT1 = HC / NW / 16;   // 48 columns = 3 fragments; 32 hidden units = 2
constexpr int KS = C / 32, KS2 = HC / 32;                              // 12 k slices of C, 8 of a hidden chunk
constexpr int NGMAX = 7, RGMIN = 4;   // row groups of 16 in a tile: at most 7; a tile of fewer than 4 still pays for 4
constexpr int PD = 4, GW = 3;                                          // weight groups in flight, fragments per group
constexpr int kGroupsChunk = (KS * NT1 + KS2 * NT) / GW;               // 8 of W1 + 8 of W2
constexpr int kGroupsTile = KS * NT / GW + NCH * kGroupsChunk;         // 12 + 96
constexpr int kGroupsWave = kGroupsTile + PD;                          // + the first PD again: the stream wraps without a branch
constexpr int kWaveUnits = kGroupsWave * GW * 64;                      // 16-byte units of a wave's stream
constexpr int kThreads = NW * 64;
constexpr int kXs = NGMAX * KS * 1024, kHs = NGMAX * KS2 * 1024;
constexpr int kOffHs = kXs, kOffPar = kXs + kHs;
constexpr int kLds = kOffPar + (HID + 6 * C) * 4;
constexpr int kStatP = 0, kStatS = 4 * 4 * NGMAX * 16 * 8;             // statistics scratch inside the hidden tile
static_assert(kLds <= 163840 && kStatS + NGMAX * 16 * 4 * 8 <= kHs, "LDS budget");
static_assert((KS * NT) % GW == 0 && (KS * NT1) % GW == 0 && (KS * NT / GW) % PD == 0 && kGroupsChunk % PD == 0,
              "ring slots must be compile-time constants");

typedef equss::lds_u32x4 L4;
// Developer build (-DEQUSS_ABL_RSSTAMPS): the shader clock at every phase boundary of every wave of one workgroup, kept in the
// spare LDS and copied out at the end (tools/rs_stamps.py); the product build contains none of it.
#ifdef EQUSS_ABL_RSSTAMPS
constexpr int kStampMax = 160, kStampWg = 77;
__device__ uint32_t g_rs_stamps[NW * kStampMax];
#define RS_STAMP()                                                                                                        \
    do {                                                                                                                  \
        if (g.ln && blockIdx.x == kStampWg && (threadIdx.x & 63) == 0 && *stamp_n < kStampMax)                                \
            reinterpret_cast<uint32_t*>(smem + kLds)[(threadIdx.x >> 6) * kStampMax + (*stamp_n)++] = (uint32_t)__builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define RS_STAMP() do {} while (0)
#endif
__device__ __forceinline__ f32x4 MM(const L4& w, const L4& a, f32x4 c) {
#ifndef EQUSS_ABL_RSNOMFMA   // (EQUSS_ABL_*: developer builds that leave one cost out; wrong results by design)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
#else
    asm volatile("" ::"v"(w), "v"(a));
    return c;
#endif
}
__device__ __forceinline__ void add4(f32x4& v, const f32x4& b) { v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }

// LDS traffic only: the weight fragments requested ahead stay in flight across the barrier
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const void* p, int64_t ld_bytes, int row0, int nrows) {
    // rows [row0, row0 + nrows): loads past the end return zero, stores past it are dropped, with no per-lane predicate
    const uint64_t base = reinterpret_cast<uint64_t>(p) + (uint64_t)row0 * (uint64_t)ld_bytes;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(nrows * (int)ld_bytes), 0x00020000);
}

// Requests for the rows of a tile, shared by the kernel's prologue and by the tile body (which asks for the NEXT tile's
// rows before it stores its own: vector-memory operations retire in issue order, so loads issued behind a tile's stores
// would wait for those stores to reach memory).
constexpr int kAttPieces = (NGMAX * KS + NW - 1) / NW;   // 1-KiB fragments of the attention tile a wave stages
constexpr int kAttHalf = (kAttPieces + 1) / 2;              // ... in two halves: pieces [0, kAttHalf) and [kAttHalf, kAttPieces)
template <int RG_LO = 0, int RG_HI = NGMAX>
__device__ __forceinline__ void request_x(const RsArgs& g, f32x4 (&xr)[NGMAX][NT], int row0, int nrows, int l_, int wave) {
    const __amdgpu_buffer_rsrc_t x_rs = rows_rsrc(g.x, g.ldx * 4, row0, nrows);
    const uint32_t x_rg = 64u * (uint32_t)g.ldx;
    const uint32_t x_lane = (uint32_t)((l_ & 15) * (int)g.ldx + wave * WN + (l_ >> 4) * 4) * 4u;
#pragma unroll
    for (int rg = RG_LO; rg < RG_HI; ++rg)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#if !defined(EQUSS_ABL_RSNOX) && !defined(EQUSS_ABL_RSNOLD)
            // (read and written once per launch and not again before the next block's tail: non-temporal)
            xr[rg][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, x_lane + nt * 64, rg * x_rg, 2));
#else
            xr[rg][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
}
// attention rows global -> registers (rows past the end read as zeros; their outputs are not stored) ...
template <int HALF>
__device__ __forceinline__ void request_att(const RsArgs& g, L4 (&pre)[kAttHalf], int row0, int nrows, int l_, int wave) {
    const __amdgpu_buffer_rsrc_t a_rs = rows_rsrc(g.A0, g.lda0 * 2, row0, nrows);
    const uint32_t a_lane = (uint32_t)((l_ & 15) * (int)g.lda0 + (l_ >> 4) * 8) * 2u;
#pragma unroll
    for (int i = HALF * kAttHalf; i < (HALF ? kAttPieces : kAttHalf); ++i) {
        int f = wave + NW * i;
        f = f < NGMAX * KS ? f : NGMAX * KS - 1;   // (the last few waves repeat the last fragment: same bytes to the same place)
        const int rg = f / KS, s = f % KS;
        pre[i - HALF * kAttHalf] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_lane, (uint32_t)(rg * 16 * (int)g.lda0 + s * 32) * 2u, 0);
    }
}
// ... -> LDS in fragment order (lane l at byte 16 l of fragment (k slice s, row group rg))
template <int HALF>
__device__ __forceinline__ void stage_att(char* xs, const L4 (&pre)[kAttHalf], int l_, int wave) {
#pragma unroll
    for (int i = HALF * kAttHalf; i < (HALF ? kAttPieces : kAttHalf); ++i) {
        int f = wave + NW * i;
        f = f < NGMAX * KS ? f : NGMAX * KS - 1;
        const int rg = f / KS, s = f % KS;
        *reinterpret_cast<L4*>(xs + (s * NGMAX + rg) * 1024 + l_ * 16) = pre[i - HALF * kAttHalf];
    }
}

// the same rows global -> LDS without passing registers (LDS-DMA: lane l of a wave instruction lands at byte 16 l of its 1-KiB
// destination, which is the fragment order)
__device__ __forceinline__ void dma_att(const RsArgs& g, char* xs, int row0, int nrows, int l_, int wave) {
    const __amdgpu_buffer_rsrc_t a_rs = rows_rsrc(g.A0, g.lda0 * 2, row0, nrows);
    const uint32_t a_lane = (uint32_t)((l_ & 15) * (int)g.lda0 + (l_ >> 4) * 8) * 2u;
#pragma unroll
    for (int i = 0; i < kAttPieces; ++i) {
        int f = wave + NW * i;
        f = f < NGMAX * KS ? f : NGMAX * KS - 1;
        const int rg = f / KS, s = f % KS;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (__attribute__((address_space(3))) void*)(xs + (s * NGMAX + rg) * 1024), 16, a_lane,
                                                 (uint32_t)(rg * 16 * (int)g.lda0 + s * 32) * 2u, 0, 0);
    }
}

// One tile of ng <= NGMAX 16-row groups: rows [row0, row0 + nrows), nrows <= 16 ng.  wr is the wave's ring of weight groups
// (on entry groups 0 .. PD-1 of the stream are in it or on their way; the same holds on exit), w_rs the wave's stream.
// On entry xr holds (or is about to receive) the tile's residual rows and xs its attention rows, staged but not yet
// published by a barrier; on exit the same holds for the next tile, rows [next_row0, next_row0 + next_nrows).
__device__ __forceinline__ void rs_tile(const RsArgs& g, char* smem, L4 (&wr)[PD][GW], const __amdgpu_buffer_rsrc_t w_rs, f32x4 (&xr)[NGMAX][NT], int ng,
                                        int row0, int nrows, int next_row0, int next_nrows, int lane, int wave, int* stamp_n) {
    // fragment number `frag` of the wave's stream (plus a scalar byte offset): one lane-offset register, no 64-bit addresses
    // One body for every tile height: row groups RGMIN .. NGMAX - 1 are skipped group by group (wave-uniform branches) when
    // the tile is shorter.  (One instantiation per height in one kernel costs registers at every join: spilled weight
    // fragments; one kernel per height costs a launch per height.)
    constexpr int NG = NGMAX;
    // Lane numbers the compiler cannot see through: every phase recomputes its lane offsets from one (a few integer
    // instructions) instead of keeping them in registers across the tile, where they do not fit beside the accumulators: a
    // reload of a spilled register is a vector-memory load whose wait drains the weight requests in flight.
    // (v_mbcnt gives the lane number out of nothing, so not even `lane` has to stay in a register for them)
    // (inside the asm statement: as a builtin it is hoisted out of the tile loop like any other invariant, and spilled)
    auto olane = [&]() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    const int lane16 = olane() * 16;   // (per tile: nothing derived from the lane number lives across tiles)
    auto active = [&](int rg) { return rg < RGMIN || rg < ng; };
#ifndef EQUSS_ABL_RSNOW
    auto wload = [&](int frag, uint32_t sbase) { return __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane16, sbase + frag * 1024, 0); };
#else
    auto wload = [&](int frag, uint32_t sbase) { return L4{(uint32_t)frag, sbase, 0u, 0u}; };
#endif
    char* const xs = smem;
    char* const hs = smem + kOffHs;
    const float* const b1_s = reinterpret_cast<const float*>(smem + kOffPar);
    const float* const b2_s = b1_s + HID;
    const float* const gamma_s = b2_s + C;
    const float* const beta_s = gamma_s + C;
    const float* const bp_s = beta_s + C;
    const float* const gamma2_s = bp_s + C;
    const float* const beta2_s = gamma2_s + C;
    // (col0 = wave * WN + fq * 4 below: this lane's four columns of column group nt are col0 + 16 nt ..+ 3)

    RS_STAMP();   // tile begins
    const __amdgpu_buffer_rsrc_t x_rs = rows_rsrc(g.x, g.ldx * 4, row0, nrows);
    const uint32_t x_rg = 64u * (uint32_t)g.ldx;   // bytes from one 16-row group to the next
    f32x4 acc2[NG][NT];   // rows rg*16 + fr, columns col0 + nt*16 + 0..3
    {
        const int l_ = olane(), col0 = wave * WN + (l_ >> 4) * 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bp_s + col0 + nt * 16);
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) { acc2[rg][nt] = xr[rg][nt]; add4(acc2[rg][nt], b); }
        }
    }
    lds_barrier();
    RS_STAMP();   // rows staged

    // The MFMA loops are written in issue order and pinned there (sched_barrier): per row group the MFMAs of k slice s, then
    // that group's A fragment of slice s + 1 into the registers just read, so that a fragment read has the MFMAs of the
    // other NG - 1 groups to land; the weight group PD further on is requested into the ring slot the slice has used up.
    // (Left to itself the scheduler sinks the weight requests to just before their use and waits for every fragment read
    // right after issuing it.)
#define RS_PIN() __builtin_amdgcn_sched_barrier(0)
    L4 a[NG];
    // Fragment reads through three 32-bit LDS addresses (lane offset included) that the compiler cannot fold constants
    // into: k slices 0 .. 5 and 6 .. 11 of xs and the hidden tile, each with 16-bit immediate offsets.  (From one base it
    // materialises a register for every fragment beyond the first 64 KiB and keeps all of them across the tile.)
    typedef const __attribute__((address_space(3))) L4* LdsFrag;
    uint32_t xa_lo = equss::lds_addr(xs) + lane16, xa_hi = xa_lo + (KS / 2) * NGMAX * 1024, ha = equss::lds_addr(hs) + lane16;
    asm volatile("" : "+v"(xa_lo), "+v"(xa_hi), "+v"(ha));
#ifndef EQUSS_ABL_RSNOLDS
    auto rd = [](uint32_t base, int off) { return *reinterpret_cast<LdsFrag>(base + off); };
#else
    auto rd = [](uint32_t base, int off) { L4 v; asm volatile("" : "=v"(v) : "v"(base), "s"(off)); return v; };
#endif
    auto frag_x = [&](int s, int rg) { return s < KS / 2 ? rd(xa_lo, (s * NGMAX + rg) * 1024) : rd(xa_hi, ((s - KS / 2) * NGMAX + rg) * 1024); };
    auto frag_h = [&](int s, int rg) { return rd(ha, (s * NGMAX + rg) * 1024); };
    // ---- G0: acc2 = x + bp + att Wp^T (12 k slices; stream groups 0 .. 11) ---------------------------------------------
#pragma unroll
    for (int rg = 0; rg < NG; ++rg) a[rg] = frag_x(0, rg);
    RS_PIN();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int rg = 0; rg < NG; ++rg) {
            if (active(rg)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc2[rg][nt] = MM(wr[s % PD][nt], a[rg], acc2[rg][nt]);
                if (s + 1 < KS) a[rg] = frag_x(s + 1, rg);
            }
            RS_PIN();
        }
#pragma unroll
        for (int e = 0; e < GW; ++e) wr[s % PD][e] = wload((s + PD) * GW + e, 0);
        RS_PIN();
    }

    RS_STAMP();   // G0 done
    // LayerNorm of the rows in acc2 with scale / shift at gs / bs: fn(rg, nt, lo, hi) gets four bf16 as two dwords.
    // Per row the sums run over the column groups in the order of the 96-column layout (see the header).
    auto layer_norm = [&](const float* gs, const float* bs, auto all_first, auto&& fn, auto&& after_group) {
        // (lane numbers the compiler cannot see through: the scratch addresses are recomputed here, a few integer
        // instructions, instead of living in registers across the tile, where they would not fit beside the accumulators)
        const int lo_ = olane();
        const int fr = lo_ & 15, fq = lo_ >> 4, col0 = wave * WN + fq * 4;
        float2* const P = reinterpret_cast<float2*>(hs + kStatP);
        float2* const S = reinterpret_cast<float2*>(hs + kStatS);
        const int pair = wave >> 1;
        // The running sums of the row groups advance together, column group by column group (per row the order of the
        // additions is unchanged): one group after the other would be one long chain of dependent additions.
        auto chains = [&](float (&s1)[NG], float (&s2)[NG]) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int rg = 0; rg < NG; ++rg) {
                    if (!active(rg)) continue;
                    const f32x4 v = acc2[rg][nt];
                    s1[rg] += (v.x + v.y) + (v.z + v.w);
                    s2[rg] += fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);
                }
                RS_PIN();
            }
        };
        if (!(wave & 1)) {
            float s1[NG], s2[NG];
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) { s1[rg] = 0.f; s2[rg] = 0.f; }
            chains(s1, s2);
#pragma unroll
            for (int rg = 0; rg < NG; ++rg)
                if (active(rg)) P[(pair * 4 + fq) * (NGMAX * 16) + rg * 16 + fr] = make_float2(s1[rg], s2[rg]);
        }
        RS_STAMP();   // (LayerNorm: first half of the sums)
        lds_barrier();
        RS_STAMP();
        if (wave & 1) {
            float s1[NG], s2[NG];
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) {
                const float2 p = P[(pair * 4 + fq) * (NGMAX * 16) + rg * 16 + fr];
                s1[rg] = p.x; s2[rg] = p.y;
            }
            chains(s1, s2);
            // (__shfl_xor with its lane addresses built here: hoisted out of the tile loop they get spilled)
            auto other = [&](float v, int mask) {
                return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lo_ ^ mask) << 2, __builtin_bit_cast(int, v)));
            };
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) { s1[rg] += other(s1[rg], 16); s2[rg] += other(s2[rg], 16); }
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) { s1[rg] += other(s1[rg], 32); s2[rg] += other(s2[rg], 32); }
#pragma unroll
            for (int rg = 0; rg < NG; ++rg)
                if (active(rg) && fq == 0) S[(rg * 16 + fr) * 4 + pair] = make_float2(s1[rg], s2[rg]);
        }
        RS_STAMP();   // (LayerNorm: second half of the sums)
        lds_barrier();
        RS_STAMP();
        auto stats_of = [&](int rg, float& mean, float& rstd) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(S + (rg * 16 + fr) * 4);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(S + (rg * 16 + fr) * 4 + 2);
            mean = ((q0.x + q0.z) + (q1.x + q1.z)) * (1.0f / C);
            const float ex2 = ((q0.y + q0.w) + (q1.y + q1.w)) * (1.0f / C);
            rstd = 1.0f / sqrtf(fmaxf(fmaf(-mean, mean, ex2), 0.f) + g.eps);
        };
        // all_first: every group's mean and scale before the first row is normalised (the square roots and divisions of
        // the groups overlap); otherwise group by group (the epilogue, where seven means and scales kept across the loop
        // would not fit beside the accumulators and the next tile's rows).  Scale and shift are read again for each group.
        constexpr bool ALL = decltype(all_first)::value;
        float means[ALL ? NG : 1], rstds[ALL ? NG : 1];
        if constexpr (ALL) {
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) stats_of(rg, means[rg], rstds[rg]);
        }
#pragma unroll
        for (int rg = 0; rg < NG; ++rg) {
            if (!active(rg)) continue;
            float mean, rstd;
            if constexpr (ALL) { mean = means[rg]; rstd = rstds[rg]; }
            else stats_of(rg, mean, rstd);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 ga = *reinterpret_cast<const f32x4*>(gs + col0 + nt * 16);
                const f32x4 be = *reinterpret_cast<const f32x4*>(bs + col0 + nt * 16);
                const f32x4 v = acc2[rg][nt];
                fn(rg, nt, equss::pack_bf16(fmaf((v.x - mean) * rstd, ga.x, be.x), fmaf((v.y - mean) * rstd, ga.y, be.y)),
                   equss::pack_bf16(fmaf((v.z - mean) * rstd, ga.z, be.z), fmaf((v.w - mean) * rstd, ga.w, be.w)));
            }
            after_group(rg);
            RS_PIN();
        }
    };

    // ---- LayerNorm2 of the updated rows -> xs (bf16, fragment order), over the attention rows ----------------------------
    // (every wave passes the two barriers inside only after its last read of the attention rows)
    {
        const int l_ = olane(), fr = l_ & 15, col0 = wave * WN + (l_ >> 4) * 4;
        int ln2_off[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = col0 + nt * 16, kk = col & 31;   // slice col / 32, 16-byte piece kk / 8 of row fr, half (kk / 4) & 1
            ln2_off[nt] = (col >> 5) * (NGMAX * 1024) + ((kk >> 3) * 16 + fr) * 16 + ((kk >> 2) & 1) * 8;
        }
        layer_norm(gamma2_s, beta2_s, std::true_type(), [&](int rg, int nt, uint32_t lo, uint32_t hi) {
            *reinterpret_cast<uint2*>(xs + ln2_off[nt] + rg * 1024) = uint2{lo, hi};
        }, [](int) {});
    }
    lds_barrier();
    RS_STAMP();   // LayerNorm2 done

#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        const uint32_t wc = (uint32_t)(KS * NT / GW + c * kGroupsChunk) * GW * 1024;   // byte offset of this chunk's 16 groups
        // ---- G1: h = a W1_c^T, wave = all rows x 32 hidden units (12 k slices; 24 fragments = groups 0 .. 7 of the chunk) ----
        f32x4 acc1[NG][NT1];
#pragma unroll
        for (int rg = 0; rg < NG; ++rg)
#pragma unroll
            for (int n1 = 0; n1 < NT1; ++n1) acc1[rg][n1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rg = 0; rg < NG; ++rg) a[rg] = frag_x(0, rg);
        RS_PIN();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) {
                if (active(rg)) {
#pragma unroll
                    for (int n1 = 0; n1 < NT1; ++n1) {
                        const int f = s * NT1 + n1;   // fragment f of the chunk's W1 stream: group f / 3, member f % 3
                        acc1[rg][n1] = MM(wr[(f / GW) % PD][f % GW], a[rg], acc1[rg][n1]);
                    }
                    if (s + 1 < KS) a[rg] = frag_x(s + 1, rg);
                }
                RS_PIN();
            }
            // a group whose last fragment was used in this step makes room for the group PD further on
#pragma unroll
            for (int j = 0; j < KS * NT1 / GW; ++j)
                if (GW * j + GW - 1 >= s * NT1 && GW * j + GW - 1 < (s + 1) * NT1) {
#pragma unroll
                    for (int e = 0; e < GW; ++e) wr[j % PD][e] = wload((j + PD) * GW + e, wc);
                }
            RS_PIN();
        }
        RS_STAMP();   // G1 done
        // ---- gelu(h + b1) -> hs: this wave's 32 hidden units are k slice `wave` of the chunk ------------------------------
        lds_barrier();   // every wave is past its reads of the previous chunk's hidden tile (and of the statistics scratch)
        RS_STAMP();   // hidden tile free
        const int l_ = olane(), fr = l_ & 15, fq = l_ >> 4, col0 = wave * WN + fq * 4;
        // The next tile's attention rows: requested here in the last chunk (every wave is past its last read of xs), they land
        // under that chunk's GELU and G2.
        if (c == NCH - 1) dma_att(g, xs, next_row0, next_nrows, l_, wave);
#pragma unroll
        for (int n1 = 0; n1 < NT1; ++n1) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(b1_s + c * HC + wave * (HC / NW) + n1 * 16 + fq * 4);
            char* const dst = hs + wave * (NGMAX * 1024) + ((n1 * 2 + (fq >> 1)) * 16 + fr) * 16 + (fq & 1) * 8;
            // (two row groups = four polynomial chains at a time, written step by step: one chain after the other leaves
            // the vector pipe waiting for its own results)
#pragma unroll
            for (int rg = 0; rg < NG; rg += 2) {
                if (!active(rg)) continue;
                const bool two = rg + 1 < NG && active(rg + 1);
                f32x4 v0 = acc1[rg][n1], v1 = acc1[rg + 1 < NG ? rg + 1 : rg][n1];
                add4(v0, b);
                add4(v1, b);
#ifndef EQUSS_ABL_RSNOGELU
                equss::gelu_f32x2 o[4] = {{v0.x, v0.y}, {v0.z, v0.w}, {v1.x, v1.y}, {v1.z, v1.w}};
                equss::gelu_erf2x4(o);
#else
                const equss::gelu_f32x2 o[4] = {{v0.x, v0.y}, {v0.z, v0.w}, {v1.x, v1.y}, {v1.z, v1.w}};
#endif
                *reinterpret_cast<uint2*>(dst + rg * 1024) = uint2{equss::pack_bf16(o[0].x, o[0].y), equss::pack_bf16(o[1].x, o[1].y)};
                if (two) *reinterpret_cast<uint2*>(dst + (rg + 1) * 1024) = uint2{equss::pack_bf16(o[2].x, o[2].y), equss::pack_bf16(o[3].x, o[3].y)};
            }
        }
        if (c == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(b2_s + col0 + nt * 16);
#pragma unroll
                for (int rg = 0; rg < NG; ++rg) add4(acc2[rg][nt], b);
            }
        }
        RS_STAMP();   // GELU done
        lds_barrier();   // the hidden tile is complete
        RS_STAMP();   // hidden tile complete
        // ---- G2: acc2 += h W2[:, chunk]^T (8 k slices; groups 8 .. 15 of the chunk) --------------------------------------
#pragma unroll
        for (int rg = 0; rg < NG; ++rg) a[rg] = frag_h(0, rg);
        RS_PIN();
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) {
                if (active(rg)) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc2[rg][nt] = MM(wr[ks % PD][nt], a[rg], acc2[rg][nt]);
                    if (ks + 1 < KS2) a[rg] = frag_h(ks + 1, rg);
                }
                RS_PIN();
            }
#pragma unroll
            for (int e = 0; e < GW; ++e) wr[ks % PD][e] = wload((KS * NT1 / GW + ks + PD) * GW + e, wc);
            RS_PIN();
        }
        RS_STAMP();   // G2 done
    }
    RS_STAMP();   // last G2 done
    lds_barrier();   // the hidden tile is free: statistics scratch
    RS_STAMP();

    // ---- epilogue: LayerNorm of the final rows as bf16, x back -------------------------------------------------------------
    // A lane holds four columns of one row, so a store instruction straight from the accumulators writes sixteen pieces of
    // 64 bytes (x) or 32 bytes (LayerNorm rows) in sixteen rows, and such stores leave a CU at a fraction of its store
    // rate.  Instead each 16-row group goes through a staging tile in the free hidden-tile space: the eight waves write
    // their columns, a barrier, then every wave instruction stores 1 KiB of consecutive memory.
    constexpr int kXFirst = 4;   // row groups of the next tile's residual rows requested before the first store
    constexpr int kStgX = kStatS + NGMAX * 16 * 4 * 8, kRowX = C * 4 + 16, kStgL = kStgX + 16 * kRowX, kRowL = C * 2 + 16;
    static_assert(kStgL + 16 * kRowL <= kHs, "staging tiles must fit the hidden tile");
    const int l_ = olane(), fr = l_ & 15, fq = l_ >> 4, col0 = wave * WN + fq * 4;
    const __amdgpu_buffer_rsrc_t ln_rs = rows_rsrc(g.ln, g.ld_ln * 2, row0, g.ln ? nrows : 0);
    auto group_out = [&](int rg) {
        // this wave's columns of the group's rows -> staging (padded rows: conflict free)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(hs + kStgX + fr * kRowX + (col0 + nt * 16) * 4) = acc2[rg][nt];
        RS_STAMP();   // (group normalised and written to the staging tiles)
        lds_barrier();
        RS_STAMP();
        // 24 KiB of x = 24 pieces of 1 KiB, three per wave; 12 KiB of LayerNorm rows = 12 pieces
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int flat = (wave * 3 + j) * 1024 + l_ * 16, r = flat / (C * 4), cb = flat % (C * 4);
            const L4 d = *reinterpret_cast<const L4*>(hs + kStgX + r * kRowX + cb);
#if !defined(EQUSS_ABL_RSNOX) && !defined(EQUSS_ABL_RSNOST)
            __builtin_amdgcn_raw_buffer_store_b128(d, x_rs, (uint32_t)(r * (int)g.ldx * 4 + cb), rg * x_rg, 2);
#else
            asm volatile("" ::"v"(d));
#endif
        }
        if (g.ln) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = wave + NW * j;
                if (q < 12) {
                    const int flat = q * 1024 + l_ * 16, r = flat / (C * 2), cb = flat % (C * 2);
                    const L4 d = *reinterpret_cast<const L4*>(hs + kStgL + r * kRowL + cb);
#ifndef EQUSS_ABL_RSNOST
                    __builtin_amdgcn_raw_buffer_store_b128(d, ln_rs, (uint32_t)(r * (int)g.ld_ln * 2 + cb), rg * 32u * (uint32_t)g.ld_ln, 0);
#else
                    asm volatile("" ::"v"(d));
#endif
                }
            }
        }
        RS_STAMP();   // (stores issued)
        lds_barrier();   // the staging tiles are free again
        RS_STAMP();
        // (the last groups of the next tile's residual rows go into the accumulator registers that as many finished groups
        // have left: all seven at once would not fit beside the accumulators)
        if (rg == NGMAX - kXFirst - 1) request_x<kXFirst, NGMAX>(g, xr, next_row0, next_nrows, l_, wave);
    };
    // (the next tile's residual rows are requested before the sums and the stores, into the registers of the G1
    // accumulators and fragments)
    request_x<0, kXFirst>(g, xr, next_row0, next_nrows, l_, wave);
    if (g.ln) {
        layer_norm(gamma_s, beta_s, std::false_type(), [&](int rg, int nt, uint32_t lo, uint32_t hi) {
            *reinterpret_cast<uint2*>(hs + kStgL + fr * kRowL + (col0 + nt * 16) * 2) = uint2{lo, hi};
        }, group_out);
    } else {
#pragma unroll
        for (int rg = 0; rg < NG; ++rg)
            if (active(rg)) group_out(rg);
    }
    RS_STAMP();   // epilogue issued
#undef RS_PIN
    // (the next tile's first writes go to xs, whose last readers passed several barriers ago; its first write to the
    // statistics scratch comes after its own first barrier, which every wave reaches only after the reads above)
}

__global__ __launch_bounds__(kThreads, 1) void block_rs_kernel(const RsArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // this workgroup's run of 16-row groups, cut into tiles of equal height
    const int G = (g.M + 15) / 16;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int gq = G / nwg, gr = G % nwg;
    const int g_begin = bid * gq + (bid < gr ? bid : gr), g_len = gq + (bid < gr ? 1 : 0);
    if (g_len == 0) return;
    const int n_tiles = (g_len + NGMAX - 1) / NGMAX;
    // Tile heights.  Every tile boundary is a burst of row traffic (the finished tile's rows out, the next tile's in) during
    // which the workgroup computes nothing; with equal heights everywhere all 256 workgroups burst at the same moments and
    // share the HBM rate.  So a workgroup's first tile is 0 .. 3 groups shorter than the balanced height, by class
    // (bid / 8) % 4 (workgroups b and b + 8 share an XCD: every XCD has all classes), and the rest of its run is cut
    // evenly: boundaries of different classes fall about a group's time apart.
    int h0 = (g_len + n_tiles - 1) / n_tiles;
    if (n_tiles > 1 && g.stagger >= 10) {
        int cut = ((bid >> 3) % (g.stagger / 10)) * (g.stagger % 10);   // (stagger = 10 classes + groups per class step)
        while (cut > 0 && (h0 - cut < 1 || g_len - (h0 - cut) > NGMAX * (n_tiles - 1))) --cut;
        h0 -= cut;
    }
    const int rest = g_len - h0, nrest = n_tiles > 1 ? n_tiles - 1 : 1;
    const int tq = rest / nrest, tr = rest % nrest;   // after the first tile: tr tiles of tq + 1 groups, then tiles of tq

    float* const par = reinterpret_cast<float*>(smem + kOffPar);
    for (int i = tid; i < HID; i += kThreads) par[i] = g.b1[i];
    for (int i = tid; i < C; i += kThreads) {
        par[HID + i] = g.b2[i];
        par[HID + C + i] = g.ln ? g.gamma[i] : 1.f;
        par[HID + 2 * C + i] = g.ln ? g.beta[i] : 0.f;
        par[HID + 3 * C + i] = g.bp[i];
        par[HID + 4 * C + i] = g.gamma2[i];
        par[HID + 5 * C + i] = g.beta2[i];
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t w_rs = rows_rsrc(g.Wimg, (int64_t)kWaveUnits * 16, wave, 1);   // this wave's stream
    L4 wr[PD][GW];
#pragma unroll
    for (int p = 0; p < PD; ++p)
#pragma unroll
        for (int e = 0; e < GW; ++e) wr[p][e] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane * 16, (p * GW + e) * 1024, 0);

    int stamp_n = 0;
    int gpos = g_begin;
    // the first tile's rows (every later tile's are requested by the tile before it)
    f32x4 xr[NGMAX][NT];
    {
        const int ng0 = h0, nrows0 = g.M - gpos * 16 < ng0 * 16 ? g.M - gpos * 16 : ng0 * 16;
        L4 pre0[kAttHalf], pre1[kAttHalf];
        request_x(g, xr, gpos * 16, nrows0, lane, wave);
        request_att<0>(g, pre0, gpos * 16, nrows0, lane, wave);
        request_att<1>(g, pre1, gpos * 16, nrows0, lane, wave);
        stage_att<0>(smem, pre0, lane, wave);
        stage_att<1>(smem, pre1, lane, wave);
    }
#pragma unroll 1
    for (int t = 0; t < n_tiles; ++t) {
        const int ng = t == 0 ? h0 : tq + (t - 1 < tr ? 1 : 0);
        const int row0 = gpos * 16;
        const int nrows = g.M - row0 < ng * 16 ? g.M - row0 : ng * 16;
        const int ng_next = t + 1 < n_tiles ? tq + (t < tr ? 1 : 0) : 0;
        const int next_row0 = (gpos + ng) * 16;
        int next_nrows = g.M - next_row0 < ng_next * 16 ? g.M - next_row0 : ng_next * 16;
        next_nrows = next_nrows < 0 ? 0 : next_nrows;
        rs_tile(g, smem, wr, w_rs, xr, ng, row0, nrows, next_row0, next_nrows, lane, wave, &stamp_n);
        gpos += ng;
    }
#ifdef EQUSS_ABL_RSSTAMPS
    __syncthreads();
    if (g.ln && blockIdx.x == kStampWg)   // (launches that write LayerNorm rows: every block's but the last)
        for (int i = tid; i < NW * kStampMax; i += kThreads) g_rs_stamps[i] = reinterpret_cast<const uint32_t*>(smem + kLds)[i];
#endif
}

// One thread per 16 bytes of a block's weight streams:
//   [wave 0..7][group 0..111][member 0..2][64 lanes][8 bf16]; lane l of a fragment holds k = 8 (l / 16) .. + 7 of weight
//   row (l % 16) of the fragment.  Groups 0..11: Wp rows wave*48 + member*16, k slice = group.  Then per hidden chunk c
//   (256 units) 16 groups: fragments f = 0..23 of W1 (k slice f / 2, rows c*256 + wave*32 + (f % 2)*16), then 8 groups of
//   W2 (k slice c*256 + 32 ks, rows wave*48 + member*16).  Groups 108..111 repeat groups 0..3.
struct RsPackArgs { const u16* wp[16]; const u16* w1[16]; const u16* w2[16]; u16* out; size_t stride; };   // stride: elements between blocks
__global__ __launch_bounds__(256) void pack_rs_weights_kernel(const RsPackArgs a) {
    const int blk = blockIdx.y;
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= NW * kWaveUnits) return;
    const int w = u / kWaveUnits, r = u % kWaveUnits;
    const int l = r & 63, f = r >> 6, e = f % GW;
    int grp = f / GW;
    if (grp >= kGroupsTile) grp -= kGroupsTile;
    const int fr = l & 15, fq = l >> 4;
    const u16* src;
    if (grp < KS * NT / GW) {
        src = a.wp[blk] + (size_t)(w * WN + e * 16 + fr) * C + grp * 32 + fq * 8;
    } else {
        const int cg = grp - KS * NT / GW, c = cg / kGroupsChunk, j = cg % kGroupsChunk;
        if (j < KS * NT1 / GW) {
            const int fi = j * GW + e, s = fi / NT1, n1 = fi % NT1;
            src = a.w1[blk] + (size_t)(c * HC + w * (HC / NW) + n1 * 16 + fr) * C + s * 32 + fq * 8;
        } else {
            const int ks = j - KS * NT1 / GW;
            src = a.w2[blk] + (size_t)(w * WN + e * 16 + fr) * HID + c * HC + ks * 32 + fq * 8;
        }
    }
    u16* dst = a.out + (size_t)blk * a.stride + (size_t)u * 8;
    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
}

int g_rs_stagger = 41;   // developer switch: 10 * classes + step (in 16-row groups) between the first tiles of two classes

}  // namespace

extern "C" void equss_dev_rs_stagger(int on) { g_rs_stagger = on >= 10 ? on : 0; }   // tuning knob, not part of the ABI

namespace equss {

size_t rs_weight_image_bytes() { return (size_t)NW * kWaveUnits * 16; }

// Weight streams of `depth` blocks (depth <= 16), block i at out + i * stride bytes (stride >= rs_weight_image_bytes(), a multiple of 16).
int pack_rs_weights(const void* const* wp, const void* const* w1, const void* const* w2, int depth, void* out, size_t stride, hipStream_t s) {
    if (depth < 1 || depth > 16) return fail(EQUSS_E_UNSUPPORTED, "pack_rs_weights: depth=%d (1..16)", depth);
    RsPackArgs a;
    for (int i = 0; i < depth; ++i) {
        if (!wp[i] || !w1[i] || !w2[i] || ((uintptr_t)wp[i] % 16) || ((uintptr_t)w1[i] % 16) || ((uintptr_t)w2[i] % 16))
            return fail(EQUSS_E_ALIGN, "pack_rs_weights: weights must be non-null and 16-byte aligned");
        a.wp[i] = reinterpret_cast<const u16*>(wp[i]);
        a.w1[i] = reinterpret_cast<const u16*>(w1[i]);
        a.w2[i] = reinterpret_cast<const u16*>(w2[i]);
    }
    a.out = reinterpret_cast<u16*>(out); a.stride = stride / 2;
    hipLaunchKernelGGL(pack_rs_weights_kernel, dim3((NW * kWaveUnits + 255) / 256, depth), dim3(256), 0, s, a);
    return check_launch("pack_rs_weights_kernel");
}

int block_rs(const void* att, int64_t lda0, const void* image, const float* bp, const float* gamma2, const float* beta2,
             const float* b1, const float* b2, float* x, int64_t ldx, const float* gamma, const float* beta, void* ln,
             int64_t ld_ln, int M, float eps, hipStream_t s) {
    if (M <= 0) return fail(EQUSS_E_SHAPE, "block_rs: M=%d must be positive", M);
    if ((lda0 % 8) || (ldx % 4) || (ld_ln % 4) || ((uintptr_t)att % 16) || ((uintptr_t)x % 16) || ((uintptr_t)ln % 8) ||
        ((uintptr_t)image % 16))
        return fail(EQUSS_E_ALIGN, "block_rs: operands must be aligned with padded leading dimensions");
    // (row offsets inside a tile are 32-bit byte offsets of buffer instructions)
    if (lda0 > (1 << 22) || ldx > (1 << 22) || ld_ln > (1 << 22))
        return fail(EQUSS_E_UNSUPPORTED, "block_rs: leading dimensions above 2^22 elements are not supported");
    RsArgs g;
    g.A0 = reinterpret_cast<const u16*>(att); g.lda0 = lda0;
    g.Wimg = reinterpret_cast<const u16*>(image);
    g.bp = bp; g.gamma2 = gamma2; g.beta2 = beta2; g.b1 = b1; g.b2 = b2;
    g.x = x; g.ldx = ldx; g.gamma = gamma; g.beta = beta; g.ln = reinterpret_cast<u16*>(ln); g.ld_ln = ld_ln;
    g.eps = eps; g.M = M; g.stagger = g_rs_stagger;
    const int groups = (M + 15) / 16;
    const int grid = groups < 256 ? groups : 256;
    if (!lds_opt_in(reinterpret_cast<const void*>(block_rs_kernel), 163840))
        return fail(EQUSS_E_UNSUPPORTED, "block_rs: cannot reserve %d bytes of LDS", kLds);
#ifdef EQUSS_ABL_RSSTAMPS
    constexpr int kLdsLaunch = kLds + NW * kStampMax * 4;
    static_assert(kLdsLaunch <= 163840, "stamps must fit the spare LDS");
#else
    constexpr int kLdsLaunch = kLds;
#endif
    hipLaunchKernelGGL(block_rs_kernel, dim3(grid), dim3(kThreads), kLdsLaunch, s, g);
    return check_launch("block_rs_kernel");
}

}  // namespace equss

#ifdef EQUSS_ABL_RSSTAMPS
extern "C" int equss_dev_rs_stamps(uint32_t* host_out) {   // (developer build only) NW x kStampMax stamps of the last launch
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rs_stamps), sizeof(uint32_t) * NW * kStampMax);
}
#endif
