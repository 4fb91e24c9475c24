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
};

constexpr int C = 384, HID = 1536, HC = 256, NCH = HID / HC;
constexpr int NW = 8, WN = C / NW, NT = WN / 16, NT1 = HC / NW / 16;   // 48 columns = 3 fragments; 32 hidden units = 2
constexpr int KS = C / 32, KS2 = HC / 32;                              // 12 k slices of C, 8 of a hidden chunk
constexpr int NGMAX = 7;
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

// One tile of NG 16-row groups: rows [row0, row0 + nrows), nrows <= 16 NG.  wr is the wave's ring of weight groups
// (on entry groups 0 .. PD-1 of the stream are in it or on their way; the same holds on exit), w_rs the wave's stream.
template <int NG>
__device__ __forceinline__ void rs_tile(const RsArgs& g, char* smem, L4 (&wr)[PD][GW], const __amdgpu_buffer_rsrc_t w_rs, int row0,
                                        int nrows, int lane, int wave) {
    // fragment number `frag` of the wave's stream (plus a scalar byte offset): one lane-offset register, no 64-bit addresses
    auto wload = [&](int frag, uint32_t sbase) { return __builtin_amdgcn_raw_buffer_load_b128(w_rs, lane * 16, sbase + frag * 1024, 0); };
    const int fr = lane & 15, fq = lane >> 4;
    char* const xs = smem;
    char* const hs = smem + kOffHs;
    const float* const b1_s = reinterpret_cast<const float*>(smem + kOffPar);
    const float* const b2_s = b1_s + HID;
    const float* const gamma_s = b2_s + C;
    const float* const beta_s = gamma_s + C;
    const float* const bp_s = beta_s + C;
    const float* const gamma2_s = bp_s + C;
    const float* const beta2_s = gamma2_s + C;
    const int col0 = wave * WN + fq * 4;   // this lane's four columns of column group nt: col0 + 16 nt ..+ 3

    // ---- residual rows straight into the accumulators; attention rows global -> registers -> LDS (fragment order) ----
    const __amdgpu_buffer_rsrc_t x_rs = rows_rsrc(g.x, g.ldx * 4, row0, nrows);
    const uint32_t x_lane = (uint32_t)(fr * (int)g.ldx + col0) * 4u;
    const uint32_t x_rg = 64u * (uint32_t)g.ldx;   // bytes from one 16-row group to the next
    f32x4 acc2[NG][NT];   // rows rg*16 + fr, columns col0 + nt*16 + 0..3
#pragma unroll
    for (int rg = 0; rg < NG; ++rg)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#ifndef EQUSS_ABL_RSNOX
            // (read and written once per launch and not again before the next block's tail: non-temporal)
            acc2[rg][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, x_lane + nt * 64, rg * x_rg, 2));
#else
            acc2[rg][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
    {
        constexpr int NF = NG * KS, PF = (NF + NW - 1) / NW;
        // (rows past the end read as zeros; their outputs are not stored)
        const __amdgpu_buffer_rsrc_t a_rs = rows_rsrc(g.A0, g.lda0 * 2, row0, nrows);
        const uint32_t a_lane = (uint32_t)(fr * (int)g.lda0 + fq * 8) * 2u;
        L4 pre[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            int f = wave + NW * i;
            f = f < NF ? f : NF - 1;   // (the last few waves repeat the last fragment: same bytes to the same place)
            const int rg = f / KS, s = f % KS;
            pre[i] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_lane, (uint32_t)(rg * 16 * (int)g.lda0 + s * 32) * 2u, 0);
        }
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            int f = wave + NW * i;
            f = f < NF ? f : NF - 1;
            const int rg = f / KS, s = f % KS;
            *reinterpret_cast<L4*>(xs + (s * NGMAX + rg) * 1024 + lane * 16) = pre[i];
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bp_s + col0 + nt * 16);
#pragma unroll
        for (int rg = 0; rg < NG; ++rg) add4(acc2[rg][nt], b);
    }
    lds_barrier();

    // ---- G0: acc2 = x + bp + att Wp^T (12 k slices; stream groups 0 .. 11) ---------------------------------------------
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        L4 a[NG];
#pragma unroll
        for (int rg = 0; rg < NG; ++rg) a[rg] = *reinterpret_cast<const L4*>(xs + (s * NGMAX + rg) * 1024 + lane * 16);
#pragma unroll
        for (int rg = 0; rg < NG; ++rg)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc2[rg][nt] = MM(wr[s % PD][nt], a[rg], acc2[rg][nt]);
#pragma unroll
        for (int e = 0; e < GW; ++e) wr[s % PD][e] = wload((s + PD) * GW + e, 0);
    }

    // LayerNorm of the rows in acc2 with scale / shift at gs / bs: fn(rg, nt, lo, hi) gets four bf16 as two dwords.
    // Per row the sums run over the column groups in the order of the 96-column layout (see the header).
    auto layer_norm = [&](const float* gs, const float* bs, auto&& fn) {
        // (lane numbers the compiler cannot see through: the scratch addresses are recomputed here, a few integer
        // instructions, instead of living in registers across the tile, where they would not fit beside the accumulators)
        int lo_ = lane;
        asm volatile("" : "+v"(lo_));
        const int fr = lo_ & 15, fq = lo_ >> 4, col0 = wave * WN + fq * 4;
        float2* const P = reinterpret_cast<float2*>(hs + kStatP);
        float2* const S = reinterpret_cast<float2*>(hs + kStatS);
        const int pair = wave >> 1;
        auto chain = [&](int rg, float& s1, float& s2) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 v = acc2[rg][nt];
                s1 += (v.x + v.y) + (v.z + v.w);
                s2 += fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);
            }
        };
        if (!(wave & 1)) {
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) {
                float s1 = 0.f, s2 = 0.f;
                chain(rg, s1, s2);
                P[(pair * 4 + fq) * (NGMAX * 16) + rg * 16 + fr] = make_float2(s1, s2);
            }
        }
        lds_barrier();
        if (wave & 1) {
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) {
                const float2 p = P[(pair * 4 + fq) * (NGMAX * 16) + rg * 16 + fr];
                float s1 = p.x, s2 = p.y;
                chain(rg, s1, s2);
                s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                if (fq == 0) S[(rg * 16 + fr) * 4 + pair] = make_float2(s1, s2);
            }
        }
        lds_barrier();
        float mean[NG], rstd[NG];
#pragma unroll
        for (int rg = 0; rg < NG; ++rg) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(S + (rg * 16 + fr) * 4);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(S + (rg * 16 + fr) * 4 + 2);
            mean[rg] = ((q0.x + q0.z) + (q1.x + q1.z)) * (1.0f / C);
            const float ex2 = ((q0.y + q0.w) + (q1.y + q1.w)) * (1.0f / C);
            rstd[rg] = 1.0f / sqrtf(fmaxf(fmaf(-mean[rg], mean[rg], ex2), 0.f) + g.eps);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 ga = *reinterpret_cast<const f32x4*>(gs + col0 + nt * 16);
            const f32x4 be = *reinterpret_cast<const f32x4*>(bs + col0 + nt * 16);
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) {
                const f32x4 v = acc2[rg][nt];
                fn(rg, nt, equss::pack_bf16(fmaf((v.x - mean[rg]) * rstd[rg], ga.x, be.x), fmaf((v.y - mean[rg]) * rstd[rg], ga.y, be.y)),
                   equss::pack_bf16(fmaf((v.z - mean[rg]) * rstd[rg], ga.z, be.z), fmaf((v.w - mean[rg]) * rstd[rg], ga.w, be.w)));
            }
        }
    };

    // ---- LayerNorm2 of the updated rows -> xs (bf16, fragment order), over the attention rows ----------------------------
    // (every wave passes the two barriers inside only after its last read of the attention rows)
    {
        int ln2_off[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = col0 + nt * 16, kk = col & 31;   // slice col / 32, 16-byte piece kk / 8 of row fr, half (kk / 4) & 1
            ln2_off[nt] = (col >> 5) * (NGMAX * 1024) + ((kk >> 3) * 16 + fr) * 16 + ((kk >> 2) & 1) * 8;
        }
        layer_norm(gamma2_s, beta2_s, [&](int rg, int nt, uint32_t lo, uint32_t hi) {
            *reinterpret_cast<uint2*>(xs + ln2_off[nt] + rg * 1024) = uint2{lo, hi};
        });
    }
    lds_barrier();

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
        for (int s = 0; s < KS; ++s) {
            L4 a[NG];
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) a[rg] = *reinterpret_cast<const L4*>(xs + (s * NGMAX + rg) * 1024 + lane * 16);
#pragma unroll
            for (int rg = 0; rg < NG; ++rg)
#pragma unroll
                for (int n1 = 0; n1 < NT1; ++n1) {
                    const int f = s * NT1 + n1;   // fragment f of the chunk's W1 stream: group f / 3, member f % 3
                    acc1[rg][n1] = MM(wr[(f / GW) % PD][f % GW], a[rg], acc1[rg][n1]);
                }
            // a group whose last fragment was used in this step makes room for the group PD further on
#pragma unroll
            for (int j = 0; j < KS * NT1 / GW; ++j)
                if (GW * j + GW - 1 >= s * NT1 && GW * j + GW - 1 < (s + 1) * NT1) {
#pragma unroll
                    for (int e = 0; e < GW; ++e) wr[j % PD][e] = wload((j + PD) * GW + e, wc);
                }
        }
        // ---- gelu(h + b1) -> hs: this wave's 32 hidden units are k slice `wave` of the chunk ------------------------------
        lds_barrier();   // every wave is past its reads of the previous chunk's hidden tile (and of the statistics scratch)
#pragma unroll
        for (int n1 = 0; n1 < NT1; ++n1) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(b1_s + c * HC + wave * (HC / NW) + n1 * 16 + fq * 4);
            char* const dst = hs + wave * (NGMAX * 1024) + ((n1 * 2 + (fq >> 1)) * 16 + fr) * 16 + (fq & 1) * 8;
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) {
                f32x4 v = acc1[rg][n1];
                add4(v, b);
#ifndef EQUSS_ABL_RSNOGELU
                const equss::gelu_f32x2 g0 = equss::gelu_erf2(equss::gelu_f32x2{v.x, v.y});
                const equss::gelu_f32x2 g1 = equss::gelu_erf2(equss::gelu_f32x2{v.z, v.w});
#else
                const equss::gelu_f32x2 g0 = {v.x, v.y}, g1 = {v.z, v.w};
#endif
                *reinterpret_cast<uint2*>(dst + rg * 1024) = uint2{equss::pack_bf16(g0.x, g0.y), equss::pack_bf16(g1.x, g1.y)};
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
        lds_barrier();   // the hidden tile is complete
        // ---- G2: acc2 += h W2[:, chunk]^T (8 k slices; groups 8 .. 15 of the chunk) --------------------------------------
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) {
            L4 a[NG];
#pragma unroll
            for (int rg = 0; rg < NG; ++rg) a[rg] = *reinterpret_cast<const L4*>(hs + (ks * NGMAX + rg) * 1024 + lane * 16);
#pragma unroll
            for (int rg = 0; rg < NG; ++rg)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc2[rg][nt] = MM(wr[ks % PD][nt], a[rg], acc2[rg][nt]);
#pragma unroll
            for (int e = 0; e < GW; ++e) wr[ks % PD][e] = wload((KS * NT1 / GW + ks + PD) * GW + e, wc);
        }
    }
    lds_barrier();   // the hidden tile is free: statistics scratch

    // ---- epilogue: LayerNorm of the final rows as bf16, x back -------------------------------------------------------------
    if (g.ln) {
        const __amdgpu_buffer_rsrc_t ln_rs = rows_rsrc(g.ln, g.ld_ln * 2, row0, nrows);
        const uint32_t ln_lane = (uint32_t)(fr * (int)g.ld_ln + col0) * 2u;
        const uint32_t ln_rg = 32u * (uint32_t)g.ld_ln;
        layer_norm(gamma_s, beta_s, [&](int rg, int nt, uint32_t lo, uint32_t hi) {
            typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{lo, hi}, ln_rs, ln_lane + nt * 32, rg * ln_rg, 0);
        });
    }
#pragma unroll
    for (int rg = 0; rg < NG; ++rg)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#ifndef EQUSS_ABL_RSNOX
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(L4, acc2[rg][nt]), x_rs, x_lane + nt * 64, rg * x_rg, 2);
#else
            asm volatile("" ::"v"(acc2[rg][nt]));
#endif
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
    const int tq = g_len / n_tiles, tr = g_len % n_tiles;   // tr tiles of tq + 1 groups, then tiles of tq

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

    int gpos = g_begin;
#pragma unroll 1
    for (int t = 0; t < n_tiles; ++t) {
        const int ng = tq + (t < tr ? 1 : 0);
        const int row0 = gpos * 16;
        const int nrows = g.M - row0 < ng * 16 ? g.M - row0 : ng * 16;
        if (ng == 7) rs_tile<7>(g, smem, wr, w_rs, row0, nrows, lane, wave);
        else if (ng == 6) rs_tile<6>(g, smem, wr, w_rs, row0, nrows, lane, wave);
        else if (ng == 5) rs_tile<5>(g, smem, wr, w_rs, row0, nrows, lane, wave);
        else rs_tile<4>(g, smem, wr, w_rs, row0, nrows, lane, wave);
        gpos += ng;
    }
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

}  // namespace

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
    g.eps = eps; g.M = M;
    const int groups = (M + 15) / 16;
    const int grid = groups < 256 ? groups : 256;
    if (!lds_opt_in(reinterpret_cast<const void*>(block_rs_kernel), kLds))
        return fail(EQUSS_E_UNSUPPORTED, "block_rs: cannot reserve %d bytes of LDS", kLds);
    hipLaunchKernelGGL(block_rs_kernel, dim3(grid), dim3(kThreads), kLds, s, g);
    return check_launch("block_rs_kernel");
}

}  // namespace equss
