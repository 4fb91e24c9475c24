// The rest of a ViT-S block after attention (vision_transformer.py:110-111 with Mlp.forward :58-64) in producer /
// consumer form:   x += att Wp^T + bp ;  a = LayerNorm2(x) ;  x += fc2(gelu(fc1(a))) ;  ln = LayerNorm(x)
// Same arithmetic, operand rounding and fp32 operation order per element as mlp_kernel<true, *> (mlp_fused.hip), so the
// same bits; what differs is who does what and where the operands live.
//
//  * One workgroup per CU = 12 waves on a 64-row tile.  Waves 0-7 (2 x 4: 32 rows x 96 columns each) read fragments,
//    run MFMAs and the vector work (GELU, LayerNorm, residual rows); they request nothing but the tile's attention rows
//    and residual rows.  Waves 8-11 only request weight stages by LDS-DMA and wait for them, two stages ahead, across
//    tiles.  Requests of one wave and its own MFMAs do not overlap on this part, requests of other waves do
//    (gemm_ws.hip).  64 rows keep a consumer at 48 + 16 accumulator registers, under the 168 that 12 waves leave.
//  * The attention rows of the tile and, after the projection, its LayerNorm2 rows sit in LDS for the whole tile
//    (48 KiB, in the ring's slice order), so neither the twelve re-reads of the LayerNorm2 tile nor its round trip
//    through memory exist; the ring carries weights only: 108 stages of 24 KiB per tile (12 Wp, 48 W1, 48 W2
//    slices, all 32 wide), every one a run of consecutive memory in the stage images (pack_tail_weights).
//  * One raw s_barrier per stage for all 12 waves ("stage landed" + "slot of the previous stage free"), plus two per
//    tile for the row statistics of the two LayerNorms.
#include "equss_common.h"
#include <type_traits>

namespace {

struct BwArgs {
    const u16* A0; int64_t lda0;                       // attention output (M, 384) bf16
    const u16* Wpt; const u16* W1t; const u16* W2t;    // stage images
    const float* bp; const float* gamma2; const float* beta2;
    const float* b1; const float* b2;
    float* x; int64_t ldx;                             // residual stream, updated in place
    const float* gamma; const float* beta;
    u16* ln; int64_t ld_ln;                            // LayerNorm(x) for the next consumer (may be null)
    float eps;
    int M;
};

constexpr int C = 384, HID = 1536, BM = 64, HC = 128, NCH = HID / HC;
constexpr int NP = 4, WN = 4;
constexpr int kStage = 24576, S = 3, PW = 24 / NP;     // 24 pieces of 1 KiB per stage, 6 per producer wave
constexpr int G0S = C / 32, G1S = C / 96, G2S = HC / 32;
constexpr int kTileStages = G0S + NCH * (G1S + G2S);    // 108
constexpr int kImg = kStage / 2;                        // elements per stage image
constexpr int kX = BM * C * 2;                          // 48 KiB: attention rows, then LayerNorm2 rows (12 slices of 4 KiB)
constexpr int kHc = BM * HC * 2;                        // 16 KiB hidden tile (two 64-wide halves of 8 KiB)
constexpr int kOffX = S * kStage, kOffHc = kOffX + kX, kOffPar = kOffHc + kHc;
constexpr int kLds = kOffPar + HID * 4 + 6 * C * 4 + BM * WN * 8;
static_assert(kLds <= 163840, "LDS budget");

typedef equss::lds_u32x4 L4;
template <int OFF> __device__ __forceinline__ L4 LD(uint32_t a) { return equss::lds_load_b128_at<OFF>(a); }
__device__ __forceinline__ f32x4 MM(const L4& a, const L4& b, f32x4 c) {
#ifndef EQUSS_ABL_BWNOMFMA   // (EQUSS_ABL_*: developer builds that leave one cost out, tools/build_ablations.sh; wrong results by design)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#else
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#endif
}
__device__ __forceinline__ f32x4 as_f32x4(const L4& v) { return __builtin_bit_cast(f32x4, v); }
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int swz4(int q) { return (0x78 >> (2 * q)) & 3; }   // chunk swizzle of 64-byte rows (mlp_fused.hip)

// every fragment register of a group through the counted wait (equss::lds_wait_for for arrays)
template <int N, int K1, int K2>
__device__ __forceinline__ void wait_frags(L4 (&x)[K1], L4 (&y)[K2]) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#pragma unroll
    for (int i = 0; i < K1; ++i) equss::lds_tie(x[i]);
#pragma unroll
    for (int i = 0; i < K2; ++i) equss::lds_tie(y[i]);
}
// 16-row fragments i = 0 .. K - 1 at STRIDE bytes from each other
template <int OFF, int STRIDE, int K>
__device__ __forceinline__ void load_frags(L4 (&f)[K], uint32_t addr) {
    static_assert(K == 2 || K == 4, "two or four 16-row fragments");
    f[0] = LD<OFF>(addr); f[1] = LD<OFF + STRIDE>(addr);
    if constexpr (K == 4) { f[2] = LD<OFF + 2 * STRIDE>(addr); f[3] = LD<OFF + 3 * STRIDE>(addr); }
}

// WM = consumer wave rows: 2 = eight consumer waves of 32 rows x 96 columns (12 waves, 168 registers each);
// 1 = four consumer waves of 64 x 96, one per SIMD (8 waves, 256 registers each): 10 instead of 16 fragment reads per
// 24 MFMAs, and the LDS reads are what the 32-row layout spends most of its time on.
template <int WM>
__global__ __launch_bounds__((4 * WM + NP) * 64, 1) void block_ws_kernel(const BwArgs g) {
    constexpr int NC = 4 * WM, MT = 4 / WM, RW = BM / WM, kThreads = (NC + NP) * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* xs = smem + kOffX;
    char* hc = smem + kOffHc;
    float* bias1_s = reinterpret_cast<float*>(smem + kOffPar);
    float* bias2_s = bias1_s + HID;
    float* gamma_s = bias2_s + C;
    float* beta_s = gamma_s + C;
    float* biasp_s = beta_s + C;
    float* gamma2_s = biasp_s + C;
    float* beta2_s = gamma2_s + C;
    float2* red_s = reinterpret_cast<float2*>(beta2_s + C);

    // tiles of this workgroup: XCD-contiguous order, as the GEMM kernels
    const int ntiles = (g.M + BM - 1) / BM;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int x_start = xcd * q8 + (xcd < r8 ? xcd : r8);
    const int x_len = q8 + (xcd < r8 ? 1 : 0);
    const int nslots = (gridDim.x + 7 - xcd) >> 3;
    if (slot >= x_len) return;
    const int my_tiles = (x_len - slot + nslots - 1) / nslots;

    for (int i = tid; i < HID; i += kThreads) bias1_s[i] = g.b1[i];
    for (int i = tid; i < C; i += kThreads) {
        bias2_s[i] = g.b2[i];
        gamma_s[i] = g.ln ? g.gamma[i] : 1.f;
        beta_s[i] = g.ln ? g.beta[i] : 0.f;
        biasp_s[i] = g.bp[i];
        gamma2_s[i] = g.gamma2[i];
        beta2_s[i] = g.beta2[i];
    }
    __syncthreads();

    if (wave >= NC) {
        // ================= producers: weight stages only =================
        const int pw = wave - NC;
        const int total = my_tiles * kTileStages;
        int issued = 0, i_n = 0, i_buf = 0;      // i_n: stage within the tile
        const int img_lane = pw * 512 + lane * 8;
        auto issue_one = [&]() {
            if (issued >= total) return;
            const u16* src;
            if (i_n < G0S) src = g.Wpt + i_n * kImg;
            else {
                const int m = i_n - G0S, c = m / (G1S + G2S), r = m % (G1S + G2S);
                src = r < G1S ? g.W1t + (c * G1S + r) * kImg : g.W2t + (c * G2S + (r - G1S)) * kImg;
            }
            char* base = smem + i_buf * kStage;
#ifdef EQUSS_ABL_BWNODMA
            if (issued < 2)
#endif
#pragma unroll
            for (int i = 0; i < PW; ++i) glds16(src + img_lane + i * (NP * 512), base + (pw + NP * i) * 1024);
            ++issued;
            i_buf = i_buf + 1 == S ? 0 : i_buf + 1;
            i_n = i_n + 1 == kTileStages ? 0 : i_n + 1;
        };
        issue_one();
        issue_one();
        int waited = 0;   // stages known to have landed
        // before the barrier that hands stage `waited` to the consumers: it has landed when at most the stage
        // requested after it is outstanding (operations retire in order)
        auto stage_barrier = [&]() {
            if (issued - 1 - waited >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ++waited;
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_one();   // into the slot of the stage consumed before this barrier
        };
        for (int t = 0; t < my_tiles; ++t) {
            for (int n = 0; n < G0S; ++n) stage_barrier();
            __builtin_amdgcn_s_barrier();                       // LayerNorm2 statistics
            for (int n = 0; n < NCH * (G1S + G2S); ++n) stage_barrier();
            __builtin_amdgcn_s_barrier();                       // final LayerNorm statistics
        }
        return;
    }

    // ================= consumers =================
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    const int g0_off = (fq ^ swz4((fr >> 2) & 3)) * 16;       // fragment read in a 64-byte-row slice: chunk fq of row fr
    const uint32_t xs_a = equss::lds_addr(xs) + (wm * RW + fr) * 64 + g0_off;            // + slice * 4096 + mt * 1024
    const uint32_t w96_off = (wn * 96 + fr) * 64 + g0_off;                              // Wp / W2 slice rows, + i * 1024
    const uint32_t w32_off = (wn * 32 + fr) * 64 + g0_off;                              // W1 slice rows, + sub * 8192 + i * 1024
    int c_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) c_off[ks] = ((4 * ks + fq) ^ ((fr >> 1) & 7)) * 16;   // 128-byte rows of the hidden tile
    const uint32_t hc_a = equss::lds_addr(hc) + (wm * RW + fr) * 128;                    // + half * 8192 + mt * 2048 + c_off
    // attention rows of a tile -> xs, slice order: piece p (16 rows of one 32-wide slice) = wave + NC i, i < 48 / NC
    const int st_row = lane >> 2, st_chunk = ((lane & 3) ^ swz4((lane >> 4) & 3)) * 8;
    auto request_att = [&](int logical) {
#pragma unroll
        for (int i = 0; i < 48 / NC; ++i) {
            const int p = wave + NC * i, slice = p >> 2, rp = p & 3;      // 4 pieces of 16 rows per slice
            int row = logical * BM + rp * 16 + st_row;
            if (row > g.M - 1) row = g.M - 1;   // rows past the end re-read the last row; their outputs are not stored
            glds16(g.A0 + (size_t)row * g.lda0 + st_chunk + slice * 32, xs + p * 1024);
        }
    };
    int c_buf = 0;
    auto next_stage = [&]() -> uint32_t {   // the barrier that hands over the next weight stage
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const uint32_t base = equss::lds_addr(smem) + c_buf * kStage;
        c_buf = c_buf + 1 == S ? 0 : c_buf + 1;
        return base;
    };

    // The residual rows of a tile are requested before the previous tile's epilogue: they come from memory while that
    // tile's statistics, LayerNorm rows and stores are under way, and the wait at the start of the tile leaves those
    // stores in flight.
    f32x4 xn[6][MT];
    auto request_x = [&](int logical) {
        const int r0 = logical * BM + wm * RW + fr;
        const float* xr = g.x + (size_t)r0 * g.ldx + wn * 96 + fq * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 6; ++nt)
#ifndef EQUSS_ABL_BWNOX
                // (the residual rows are read and written once per launch and not again before the next block's tail: with
                // the non-temporal hint they leave the caches to the rows the neighbouring kernels re-read, +1 % end to end)
                xn[nt][mt] = r0 + mt * 16 < g.M ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + (size_t)(mt * 16) * g.ldx + nt * 16))
                                                : f32x4{0.f, 0.f, 0.f, 0.f};
#else
                xn[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
    };
    int c_tile = slot;
    request_att(x_start + c_tile);
    request_x(x_start + c_tile);
    for (int t = 0; t < my_tiles; ++t, c_tile += nslots) {
        const int m0 = (x_start + c_tile) * BM;
        float* xrow = g.x + (size_t)(m0 + wm * RW + fr) * g.ldx + wn * 96 + fq * 4;   // + mt * 16 rows, + nt * 16
        bool row_ok[MT];   // (only the last tile can be ragged)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) row_ok[mt] = m0 + wm * RW + mt * 16 + fr < g.M;
        // residual rows here, this wave's attention pieces in LDS; younger than both are only the previous tile's stores
        // (a tile with a successor is full: 6 MT stores of x, as many of the LayerNorm rows if they are wanted)
        if (WM == 1 && t > 0) request_x(x_start + c_tile);   // (one wave per SIMD: no room to hold them across the epilogue)
        if (t == 0 || WM == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (g.ln) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 * MT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * MT) : "memory");
        f32x4 acc2[6][MT];   // [nt][mt]: rows wm*32 + mt*16 + fr, columns wn*96 + nt*16 + fq*4 + 0..3
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 6; ++nt) acc2[nt][mt] = xn[nt][mt];
        {
            L4 b[6];
            const uint32_t sb = equss::lds_addr(biasp_s) + (wn * 96 + fq * 4) * 4;
            b[0] = LD<0>(sb); b[1] = LD<64>(sb); b[2] = LD<128>(sb); b[3] = LD<192>(sb); b[4] = LD<256>(sb); b[5] = LD<320>(sb);
            equss::lds_wait_for<0>(b[0], b[1], b[2], b[3], b[4], b[5]);
#pragma unroll
            for (int nt = 0; nt < 6; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f32x4 b4 = as_f32x4(b[nt]);
                    acc2[nt][mt].x += b4.x; acc2[nt][mt].y += b4.y; acc2[nt][mt].z += b4.z; acc2[nt][mt].w += b4.w;
                }
        }
        // one 32-wide step of acc2 += A W^T: A fragments at `sa` (+ 1024 per 16 rows), weight slice rows at `sw`
        auto step96 = [&](uint32_t sa, uint32_t sw, auto a_stride) {   // a_stride: bytes from one 16-row A fragment to the next
            L4 a[MT], w[6];
            load_frags<0, decltype(a_stride)::value>(a, sa);
            w[0] = LD<0>(sw); w[1] = LD<1024>(sw); w[2] = LD<2048>(sw);
            w[3] = LD<3072>(sw); w[4] = LD<4096>(sw); w[5] = LD<5120>(sw);
            __builtin_amdgcn_s_setprio(1);
            {
                L4 (&w012)[3] = reinterpret_cast<L4 (&)[3]>(w[0]);
                wait_frags<3>(a, w012);
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc2[nt][mt] = MM(w[nt], a[mt], acc2[nt][mt]);
            equss::lds_wait_for<0>(w[3], w[4], w[5]);
#pragma unroll
            for (int nt = 3; nt < 6; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc2[nt][mt] = MM(w[nt], a[mt], acc2[nt][mt]);
            __builtin_amdgcn_s_setprio(0);
        };
        // row sums of acc2 -> red_s (in-lane, two shuffles, then one value per wave and row)
        auto row_stats = [&]() {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int nt = 0; nt < 6; ++nt) {
                    const f32x4 v = acc2[nt][mt];
                    s1 += (v.x + v.y) + (v.z + v.w);
                    s2 += fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);
                }
                s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                if (fq == 0) equss::lds_store_f32x2(red_s + ((wm * RW + mt * 16 + fr) * WN + wn), s1, s2);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        // LayerNorm of the rows in acc2 with scale / shift at gs / bs: fn(nt, mt, lo, hi) gets four bf16 as two dwords
        auto normalise = [&](const float* gs, const float* bs, auto&& fn) {
            // (scale / shift two column groups at a time: the next tile's residual rows may be waiting in 48 registers)
            L4 pru[MT][2];
            const uint32_t sg = equss::lds_addr(gs) + (wn * 96 + fq * 4) * 4, sb = equss::lds_addr(bs) + (wn * 96 + fq * 4) * 4;
            const uint32_t sp = equss::lds_addr(red_s) + (wm * RW + fr) * WN * 8;
            pru[0][0] = LD<0>(sp); pru[0][1] = LD<16>(sp);
            pru[1][0] = LD<16 * WN * 8>(sp); pru[1][1] = LD<16 * WN * 8 + 16>(sp);
            if constexpr (MT == 4) {
                pru[2][0] = LD<32 * WN * 8>(sp); pru[2][1] = LD<32 * WN * 8 + 16>(sp);
                pru[3][0] = LD<48 * WN * 8>(sp); pru[3][1] = LD<48 * WN * 8 + 16>(sp);
            }
            {
                L4 (&lo)[MT] = reinterpret_cast<L4 (&)[MT]>(pru[0][0]);
                L4 (&hi)[MT] = reinterpret_cast<L4 (&)[MT]>(pru[MT / 2][0]);
                wait_frags<0>(lo, hi);
            }
            float mean[MT], rstd[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 q0 = as_f32x4(pru[mt][0]), q1 = as_f32x4(pru[mt][1]);   // (sum, sum of squares) of the four waves
                mean[mt] = ((q0.x + q0.z) + (q1.x + q1.z)) * (1.0f / C);
                const float ex2 = ((q0.y + q0.w) + (q1.y + q1.w)) * (1.0f / C);
                rstd[mt] = 1.0f / sqrtf(fmaxf(fmaf(-mean[mt], mean[mt], ex2), 0.f) + g.eps);
            }
#pragma unroll
            for (int np = 0; np < 3; ++np) {
                L4 gau[2], beu[2];
                gau[0] = equss::lds_load_b128(reinterpret_cast<const char*>(gs) + (wn * 96 + fq * 4) * 4 + np * 128);
                gau[1] = equss::lds_load_b128(reinterpret_cast<const char*>(gs) + (wn * 96 + fq * 4) * 4 + np * 128 + 64);
                beu[0] = equss::lds_load_b128(reinterpret_cast<const char*>(bs) + (wn * 96 + fq * 4) * 4 + np * 128);
                beu[1] = equss::lds_load_b128(reinterpret_cast<const char*>(bs) + (wn * 96 + fq * 4) * 4 + np * 128 + 64);
                equss::lds_wait_for<0>(gau[0], gau[1], beu[0], beu[1]);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int nt = 2 * np + h;
                        const f32x4 ga = as_f32x4(gau[h]), be = as_f32x4(beu[h]);
                        const f32x4 v = acc2[nt][mt];
                        fn(nt, mt, equss::pack_bf16(fmaf((v.x - mean[mt]) * rstd[mt], ga.x, be.x), fmaf((v.y - mean[mt]) * rstd[mt], ga.y, be.y)),
                           equss::pack_bf16(fmaf((v.z - mean[mt]) * rstd[mt], ga.z, be.z), fmaf((v.w - mean[mt]) * rstd[mt], ga.w, be.w)));
                    }
            }
            (void)sg; (void)sb;
        };

        // ---- G0: acc2 = x + bp + att Wp^T (12 steps of 32) ---------------------------------------------------
        for (int kt = 0; kt < G0S; ++kt) {
            const uint32_t base = next_stage();
            step96(xs_a + kt * 4096, base + w96_off, std::integral_constant<int, 1024>());
        }
        // ---- LayerNorm2 of the updated rows -> xs (bf16, slice order), over the attention rows ------------------
        row_stats();   // (every consumer is past its last read of the attention rows)
        normalise(gamma2_s, beta2_s, [&](int nt, int mt, uint32_t lo, uint32_t hi) {
            // columns wn*96 + nt*16 + fq*4 .. + 3: slice wn*3 + nt/2, chunk (nt & 1) * 2 + fq / 2, half fq & 1
            const int chunk = (nt & 1) * 2 + (fq >> 1);
            equss::lds_store_b64(xs + (wn * 3 + (nt >> 1)) * 4096 + (wm * MT + mt) * 1024 + (fr * 4 + (chunk ^ swz4(fr >> 2))) * 16 + (fq & 1) * 8,
                                 lo, hi);
        });
        for (int c = 0; c < NCH; ++c) {
            // ---- G1: h = a W1_c^T (4 steps of 96 = three 32-wide slices), wave tile 32 x 32 -----------------------
            f32x4 acc1[2][MT];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc1[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kt = 0; kt < G1S; ++kt) {
                const uint32_t base = next_stage();
                const uint32_t sa = xs_a + kt * 3 * 4096, sw = base + w32_off;
                L4 a0[MT], a1[MT], a2[MT], w0[2], w1[2], w2[2];   // row fragments of a, two of W1, per 32-wide slice
#define EQUSS_BW_LOAD(a, w, sub) \
                load_frags<sub * 4096, 1024>(a, sa); w[0] = LD<sub * 8192>(sw); w[1] = LD<sub * 8192 + 1024>(sw);
#define EQUSS_BW_MMS(a, w)                                           \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)     \
                    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc1[nt][mt] = MM(w[nt], a[mt], acc1[nt][mt]);
                EQUSS_BW_LOAD(a0, w0, 0)
                EQUSS_BW_LOAD(a1, w1, 1)
                EQUSS_BW_LOAD(a2, w2, 2)
                __builtin_amdgcn_s_setprio(1);
                wait_frags<2 * (MT + 2)>(a0, w0);
                EQUSS_BW_MMS(a0, w0)
                wait_frags<MT + 2>(a1, w1);
                EQUSS_BW_MMS(a1, w1)
                wait_frags<0>(a2, w2);
                EQUSS_BW_MMS(a2, w2)
                __builtin_amdgcn_s_setprio(0);
#undef EQUSS_BW_LOAD
#undef EQUSS_BW_MMS
            }
            // ---- gelu(h + b1) -> hc (two 64-wide halves, 128-byte rows) --------------------------------------------
            // (every consumer passed a barrier after its last read of the previous chunk's h)
            {
                L4 b[2];
                const uint32_t sb = equss::lds_addr(bias1_s) + (c * HC + wn * 32 + fq * 4) * 4;
                b[0] = LD<0>(sb); b[1] = LD<64>(sb);
                equss::lds_wait_for<0>(b[0], b[1]);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int nloc = wn * 32 + nt * 16 + fq * 4;          // column within the chunk
                    const f32x4 b4 = as_f32x4(b[nt]);
                    const int half = nloc >> 6, n6 = nloc & 63;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int row = wm * RW + mt * 16 + fr;
                        f32x4 v = acc1[nt][mt];
                        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
#ifndef EQUSS_ABL_BWNOGELU
                        const equss::gelu_f32x2 g0 = equss::gelu_erf2(equss::gelu_f32x2{v.x, v.y});
                        const equss::gelu_f32x2 g1 = equss::gelu_erf2(equss::gelu_f32x2{v.z, v.w});
#else
                        const equss::gelu_f32x2 g0 = {v.x, v.y}, g1 = {v.z, v.w};
#endif
                        equss::lds_store_b64(hc + half * 8192 + row * 128 + (((n6 >> 3) ^ ((row >> 1) & 7)) * 16) + ((n6 >> 2) & 1) * 8,
                                             equss::pack_bf16(g0.x, g0.y), equss::pack_bf16(g1.x, g1.y));
                    }
                }
            }
            if (c == 0) {
                L4 b[6];
                const uint32_t sb = equss::lds_addr(bias2_s) + (wn * 96 + fq * 4) * 4;
                b[0] = LD<0>(sb); b[1] = LD<64>(sb); b[2] = LD<128>(sb); b[3] = LD<192>(sb); b[4] = LD<256>(sb); b[5] = LD<320>(sb);
                equss::lds_wait_for<0>(b[0], b[1], b[2], b[3], b[4], b[5]);
#pragma unroll
                for (int nt = 0; nt < 6; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f32x4 b4 = as_f32x4(b[nt]);
                        acc2[nt][mt].x += b4.x; acc2[nt][mt].y += b4.y; acc2[nt][mt].z += b4.z; acc2[nt][mt].w += b4.w;
                    }
            }
            // ---- G2: acc2 += h W2[:, c]^T (4 steps of 32; the step's barrier publishes h) ---------------------------
            for (int ks = 0; ks < G2S; ++ks) {
                const uint32_t base = next_stage();
                step96(hc_a + (ks >> 1) * 8192 + c_off[ks & 1], base + w96_off, std::integral_constant<int, 2048>());
                // the tile's last G1 reads of xs are behind every consumer: the next tile's attention rows may come
                if (c == NCH - 1 && ks == 0 && t + 1 < my_tiles) request_att(x_start + c_tile + nslots);
            }
        }
        // the next tile's residual rows: requested here, under the epilogue (every path defines xn: its old value has
        // been dead since the tile began, and a conditional definition would keep 48 registers alive across the tile)
        if (WM == 2 && t + 1 < my_tiles) request_x(x_start + c_tile + nslots);
        else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 6; ++nt) xn[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- epilogue: x back, LayerNorm of the final rows as bf16 -------------------------------------------------
        row_stats();
        if (g.ln) {
            u16* lrow = g.ln + (size_t)(m0 + wm * RW + fr) * g.ld_ln + wn * 96 + fq * 4;
            normalise(gamma_s, beta_s, [&](int nt, int mt, uint32_t lo, uint32_t hi) {
                if (row_ok[mt]) *reinterpret_cast<uint2*>(lrow + (size_t)(mt * 16) * g.ld_ln + nt * 16) = uint2{lo, hi};
            });
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 6; ++nt)
#ifndef EQUSS_ABL_BWNOX
                if (row_ok[mt]) __builtin_nontemporal_store(acc2[nt][mt], reinterpret_cast<f32x4*>(xrow + (size_t)(mt * 16) * g.ldx + nt * 16));
#else
                asm volatile("" ::"v"(acc2[nt][mt]));
#endif
    }
}

int g_block_ws = 3;   // developer switch: 3 (default) = block_rs_kernel inside equss_vit_forward, 1 = this kernel, 0 = mlp_kernel<true, true>

}  // namespace

extern "C" void equss_dev_block_ws(int on) { g_block_ws = on; }  // tuning knob, not part of the ABI

namespace equss {

bool block_ws_enabled() { return g_block_ws != 0; }
int block_ws_mode() { return g_block_ws; }

// All three weight matrices as stage images (pack_tail_weights: Wp, W1, and W2 in 32-wide slices).
int block_ws(const void* att, int64_t lda0, const void* wp_image, const void* w1_image, const void* w2_image, const float* bp,
             const float* gamma2, const float* beta2, const float* b1, const float* b2, float* x, int64_t ldx, const float* gamma,
             const float* beta, void* ln, int64_t ld_ln, int M, float eps, hipStream_t s) {
    if (M <= 0) return fail(EQUSS_E_SHAPE, "block_ws: M=%d must be positive", M);
    if ((lda0 % 8) || (ldx % 4) || (ld_ln % 4) || ((uintptr_t)att % 16) || ((uintptr_t)x % 16) || ((uintptr_t)ln % 8))
        return fail(EQUSS_E_ALIGN, "block_ws: operands must be aligned with padded leading dimensions");
    BwArgs g;
    g.A0 = reinterpret_cast<const u16*>(att); g.lda0 = lda0;
    g.Wpt = reinterpret_cast<const u16*>(wp_image); g.W1t = reinterpret_cast<const u16*>(w1_image);
    g.W2t = reinterpret_cast<const u16*>(w2_image);
    g.bp = bp; g.gamma2 = gamma2; g.beta2 = beta2; g.b1 = b1; g.b2 = b2;
    g.x = x; g.ldx = ldx; g.gamma = gamma; g.beta = beta; g.ln = reinterpret_cast<u16*>(ln); g.ld_ln = ld_ln;
    g.eps = eps; g.M = M;
    const int ntiles = (M + BM - 1) / BM;
    const int grid = ntiles < 256 ? ntiles : 256;
    if (g_block_ws == 2) {
        if (!lds_opt_in(reinterpret_cast<const void*>(block_ws_kernel<1>), kLds))
            return fail(EQUSS_E_UNSUPPORTED, "block_ws: cannot reserve %d bytes of LDS", kLds);
        hipLaunchKernelGGL(block_ws_kernel<1>, dim3(grid), dim3((4 + NP) * 64), kLds, s, g);
    } else {
        if (!lds_opt_in(reinterpret_cast<const void*>(block_ws_kernel<2>), kLds))
            return fail(EQUSS_E_UNSUPPORTED, "block_ws: cannot reserve %d bytes of LDS", kLds);
        hipLaunchKernelGGL(block_ws_kernel<2>, dim3(grid), dim3((8 + NP) * 64), kLds, s, g);
    }
    return check_launch("block_ws_kernel");
}

}  // namespace equss
