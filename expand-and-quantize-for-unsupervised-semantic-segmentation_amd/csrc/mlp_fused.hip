// The whole MLP of a ViT-S block in one kernel (vision_transformer.py:58-64 inside :106-114):
//     x += fc2(gelu(fc1(a))) ,   ln = LayerNorm(x)          a = LayerNorm2(x) in bf16, C = 384, hidden = 1536
// without the (rows, 1536) hidden activation ever reaching HBM.  The two separate GEMMs write and
// re-read 6 KB per token of hidden activation per block (630 MB per 64-image launch), and on this
// part a GEMM's HBM time adds to its MFMA time instead of hiding behind it, so the round trip costs
// its full bandwidth time.
//
// One persistent workgroup (8 waves, one per CU) owns a 128-row tile and walks the hidden dimension
// in 12 chunks of 128:
//     G1:  h = gelu(a[128 x 384] W1_c^T + b1_c)     4 K steps of 96; h (128 x 128 bf16) -> LDS
//     G2:  acc2[128 x 384] += h W2[:, c]^T          2 K steps of 64, A operand straight from that LDS tile
// acc2 starts from x + b2 and ends in the same residual + LayerNorm epilogue as the fc2 kernel it
// replaces; the fp32 operation order per output element (k ascending, x-initialised accumulator) is
// that kernel's too.  Operand slices arrive through a 2-slot LDS-DMA ring (global_load_lds_dwordx4, retired
// with counted s_waitcnt vmcnt so that epilogue stores and residual loads may stay in flight);
// fragment layout, chunk swizzle and the C^T trick are those of gemm8_bf16.hip.
#include "equss_common.h"

namespace {

struct MlpArgs {
    // PROJ variant only: attention output (M, 384) bf16, projection weights (384, 384) / bias, LayerNorm2
    // parameters; A is then an output too (LayerNorm2(x) is written there before the MLP reads it back)
    const u16* A0; int64_t lda0;
    const u16* Wp; const float* bp;
    const float* gamma2; const float* beta2;
    const u16* A; int64_t lda;        // LayerNorm2 output, (M, 384) bf16
    const u16* W1; const float* b1;   // (1536, 384), (1536)
    const u16* W2; const float* b2;   // (384, 1536), (384)
    float* x; int64_t ldx;            // residual stream, updated in place
    const float* gamma; const float* beta;
    u16* ln; int64_t ld_ln;           // LayerNorm(x) for the next consumer (may be null)
    float eps;
    int M;
    // TILED variant: the projection and fc1 weights in ring-stage order (pack_tail_weights below), so that every LDS-DMA
    // request moves 1 KiB of consecutive memory
    const u16* Wpt; const u16* W1t;
};

constexpr int C = 384, HID = 1536, BM = 128, HC = 128, BK = 64;
constexpr int NW = 8, WN = 4;                    // waves as 2 (rows) x 4 (columns)
constexpr int kThreads = NW * 64;
constexpr int kStage = 384 * BK * 2;              // 48 KiB: the W2 slice of a G2 step; a G1 step (a slice + W1 slice) uses 32 KiB
constexpr int S = 2;                              // ring slots: one stage in flight (three 32-KiB slots with 32-wide G2 steps measured 5 % slower)
constexpr int kHc = BM * HC * 2;                  // 32 KiB
constexpr int NCH = HID / HC;                     // 12 chunks
constexpr int G1W = 96;                          // G1 steps are 96 wide: three 32-wide sub-tiles (a 8 KiB + W1 8 KiB each) fill a slot
constexpr int G1S = C / G1W, G2S = HC / BK;       // 4 + 2 steps per chunk
constexpr int G1L = 4, G2L = 6;                   // LDS-DMA loads per wave in a G1 / G2 stage
constexpr int G0S = C / 32;                       // PROJ variant: 12 steps of 32 (attention slice 8 KiB + Wp slice 24 KiB)
#ifdef EQUSS_MLP_STAMPS
constexpr int kLds = S * kStage + kHc + HID * 4 + 6 * C * 4 + BM * WN * 8 + 8 * 160 * 8;
#else
constexpr int kLds = S * kStage + kHc + HID * 4 + 6 * C * 4 + BM * WN * 8;
#endif
static_assert(kLds <= 163840, "LDS budget");

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const void __attribute__((address_space(1)))*)gsrc,
        (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// Fragment reads and MFMA groups of the ring steps are scheduled by hand: inline-asm ds_read_b128 in the order of first
// use, one counted wait per MFMA group (LD / equss::lds_wait_for), each group at raised wave priority.  Left to the
// compiler the reads were re-ordered and most groups waited for every read of the step; s_setprio makes the SIMD's
// other wave issue its reads and its LDS-DMA requests into the gaps of the group that is running (-5 % together).
typedef equss::lds_u32x4 L4;
template <int OFF> __device__ __forceinline__ L4 LD(uint32_t a) { return equss::lds_load_b128_at<OFF>(a); }
__device__ __forceinline__ f32x4 MM(const L4& a, const L4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#ifdef EQUSS_MLP_STAMPS   // developer build (tools/ubench/mlp_stamps.hip): 100 MHz clock at the phase boundaries of every wave of one workgroup
constexpr int kStampMax = 160;
__device__ unsigned long long g_mlp_stamps[NW * kStampMax];
#define MLP_STAMP() do { if (lane == 0 && n_stamp < kStampMax) { asm volatile("" ::: "memory"); stamps_s[wave * kStampMax + n_stamp] = wall_clock64(); } ++n_stamp; } while (0)
#else
#define MLP_STAMP() do { } while (0)
#endif

// TILED (with PROJ): the 32-wide K slices of Wp and W1 come from stage images of those matrices, and a full tile's
// LayerNorm2 rows are written to g.A in the image order of its G1 stages too (g.A is scratch of this kernel: rows
// [m0, m0 + 128) x 384 are the tile's own 96 KiB, whatever their order inside), so all of a G1 stage and three quarters of a
// projection stage arrive as 1-KiB runs of consecutive memory.  A request for a 64-byte row piece costs the address path what
// one for 128 consecutive bytes costs: 66 against 115 GB/s per CU (tools/ubench/ring_ws.hip), and these stages are what the
// ring steps wait for.  Image order of a 32-wide slice: [16-row piece][row in piece][16-byte chunk, XORed as in LDS].
template <bool PROJ, bool TILED>
__global__ __launch_bounds__(kThreads, 2) void mlp_kernel(const MlpArgs g) {
    static_assert(PROJ || !TILED, "the stage-image operands belong to the block-tail variant");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    char* hc = smem + S * kStage;
    float* bias1_s = reinterpret_cast<float*>(hc + kHc);
    float* bias2_s = bias1_s + HID;
    float* gamma_s = bias2_s + C;
    float* beta_s = gamma_s + C;
    float* biasp_s = beta_s + C;
    float* gamma2_s = biasp_s + C;
    float* beta2_s = gamma2_s + C;
    float2* red_s = reinterpret_cast<float2*>(beta2_s + C);
#ifdef EQUSS_MLP_STAMPS
    unsigned long long* stamps_s = reinterpret_cast<unsigned long long*>(red_s + BM * WN);
    int n_stamp = 0;
#endif

    // ---- tiles of this workgroup: XCD-contiguous order, as the GEMM kernels ---------------------
    const int ntiles = (g.M + BM - 1) / BM;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int x_start = xcd * q8 + (xcd < r8 ? xcd : r8);
    const int x_len = q8 + (xcd < r8 ? 1 : 0);
    const int nslots = (gridDim.x + 7 - xcd) >> 3;
    if (slot >= x_len) return;
    const int my_tiles = (x_len - slot + nslots - 1) / nslots;
    const int total = my_tiles * ((PROJ ? G0S : 0) + NCH * (G1S + G2S));

    for (int i = tid; i < HID; i += kThreads) bias1_s[i] = g.b1[i];
    for (int i = tid; i < C; i += kThreads) {
        bias2_s[i] = g.b2[i];
        gamma_s[i] = g.ln ? g.gamma[i] : 1.f;
        beta_s[i] = g.ln ? g.beta[i] : 0.f;
        if (PROJ) {
            biasp_s[i] = g.bp[i];
            gamma2_s[i] = g.gamma2[i];
            beta2_s[i] = g.beta2[i];
        }
    }
    __syncthreads();

    // ---- LDS fragment offsets (rows of 128 B, 16-B chunks XOR-swizzled by (row >> 1) & 7) --------
    int c_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) c_off[ks] = ((4 * ks + fq) ^ ((fr >> 1) & 7)) * 16;
    const int a_row = (wm * 64 + fr) * 128;        // + mt * 2048
    const int w2_row = (wn * 96 + fr) * 128;

    // ---- residual / LayerNorm rows go through buffer instructions: a resource per 16-row group of a wave pair
    // (base = rows m0 + 16 mt and m0 + 64 + 16 mt, extent = the rows the matrix still has from there, so rows
    // past M load zeros and drop their stores with no per-lane predicate; all of it scalar arithmetic), a 32-bit
    // lane offset and an immediate.  (Written as one 64-bit pointer per access,
    // the compiler kept the tile-invariant part of all 12 + 12 + 24 of them in registers across the kernel and
    // spilled; a reload of a spilled register is a vector-memory load whose s_waitcnt vmcnt(0) also waits for
    // the stores just issued, so the epilogues stored one row piece per memory round trip.)
    // The lane offsets are recomputed where they are used, from a lane number the compiler cannot see through
    // (a dozen integer instructions per tile): kept across the kernel they would not fit beside the accumulators.
    auto swz4t = [](int q) { return (0x78 >> (2 * q)) & 3; };   // (= swz4 below)
    auto opaque_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto x_lane_of = [&](int l) { return (uint32_t)((wm * 64 + (l & 15)) * (int)g.ldx + wn * 96 + (l >> 4) * 4) * 4u; };   // + nt * 64 bytes
    constexpr int OCH = 12, PER_LANE = 16 * OCH / 64;     // 16-byte chunks per output row piece, stores per lane
    // piece j of a lane in the bf16 row staging: offset in the wave's staging rows, byte offset in the output rows
    auto stg_off_of = [&](int l, int j) { const int idx = j * 64 + l; return (idx / OCH) * (96 * 2 + 16) + (idx % OCH) * 16; };
    auto out_lane_of = [&](int l, int j, int64_t ld) {
        const int idx = j * 64 + l;
        return (uint32_t)((wm * 64 + idx / OCH) * (int)ld + wn * 96 + (idx % OCH) * 8) * 2u;
    };
    // LayerNorm2 rows in stage-image order: byte offset of the lane's piece j in the tile's 96 KiB, + 1024 per 16-row group:
    // columns wn * 96 .. + 96 are G1 step wn, 32-wide slice (idx % OCH) / 4, 16-byte chunk (idx % OCH) % 4 of row idx / OCH
    auto out_tiled_of = [&](int l, int j) {
        const int idx = j * 64 + l, r16 = idx / OCH, c12 = idx % OCH;
        return (uint32_t)((wn * 3 + c12 / 4) * 8192 + wm * 4096 + (r16 * 4 + ((c12 & 3) ^ swz4t(r16 >> 2))) * 16);
    };
    auto rows_rsrc = [&](const void* p, int64_t ld, int elem, int m0, int mt) {
        const int row0 = m0 + 16 * mt;
        int rows = g.M - row0 < BM - 16 * mt ? g.M - row0 : BM - 16 * mt;
        rows = rows < 0 ? 0 : rows;
        // (every word through readfirstlane: the values are wave-uniform, but left to itself the compiler keeps some of
        // them in vector registers and wraps every buffer instruction in a loop that reads them back lane by lane)
        const uint64_t base = reinterpret_cast<uint64_t>(p) + (uint64_t)row0 * (uint64_t)ld * elem;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(rows * (int)ld * elem), 0x00020000);
    };

    // ---- staging: 1-KiB pieces of 8 rows; piece p = wave + 8 i, so a wave's pieces share the swizzle --
    const int st_row = lane >> 3;
    const int st_chunk = ((lane & 7) ^ ((4 * (wave & 1) + (st_row >> 1)) & 7)) * 8;   // element offset in the row
    const int w2_off = (wave * 8 + st_row) * HID + st_chunk;        // + 64 * HID per further piece
    // G0 / G1 slices are 32 wide: rows of 64 B, pieces of 16 rows, 16-B chunks XORed with swz4((row >> 2) & 3).
    // A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): a group reads
    // rows r, r + 12 (chunk fq) and r + 4, r + 8 (chunk fq ^ 1) of each bank quarter, and the permutation
    // 0,2,3,1 sends those four to four different 16-B columns (the plain (row >> 2) & 3 puts them two and two
    // on the same banks: every fragment read took twice its LDS cycles).
    auto swz4 = [](int q) { return (0x78 >> (2 * q)) & 3; };
    const int g0_chunk = ((lane & 3) ^ swz4((lane >> 4) & 3)) * 8;
    const int g0_row = wave * 16 + (lane >> 2);                     // + 128 per further piece
    const int g0_off = (fq ^ swz4((fr >> 2) & 3)) * 16;             // fragment read: chunk fq of row fr
    const u16* a_src;   // this lane's 16 bytes of the wave's 16-row piece of the a slice (32-wide sub-tile format)
    const u16* a0_src = nullptr;
    int i_tile = slot, i_p = 0, i_c = 0, i_j = 0, i_buf = 0, issued = 0;
    bool hold_ok = false;   // PROJ: the tile's LayerNorm2 rows are in memory, the first G1 stage may be requested
    bool i_tiled = false;   // TILED: the tile being requested is full, its LayerNorm2 rows are in stage-image order
    const int img_lane = wave * 512 + lane * 8;   // element offset of this lane's 16 bytes in an 8-KiB slice image (piece = wave)
    auto set_a = [&](int logical) {
        const int m0 = logical * BM;
        if (TILED) {
            i_tiled = m0 + BM <= g.M;
            if (i_tiled) a_src = g.A + (size_t)m0 * C + img_lane;
        }
        if (!(TILED && i_tiled))
        {
            int am = m0 + g0_row;
            if (am > g.M - 1) am = g.M - 1;   // tail rows re-read the last row; outputs are masked
            a_src = g.A + (size_t)am * g.lda + g0_chunk;
        }
        if (PROJ) {
            int am = m0 + g0_row;
            if (am > g.M - 1) am = g.M - 1;
            a0_src = g.A0 + (size_t)am * g.lda0 + g0_chunk;
        }
    };
    auto issue_one = [&]() {
        if (issued >= total) return;
        char* base = smem + i_buf * kStage;
        if (PROJ && i_p < G0S) {
            glds16(a0_src + i_p * 32, base + wave * 1024);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (TILED) glds16(g.Wpt + i_p * 12288 + i * 4096 + img_lane, base + 8192 + (wave + 8 * i) * 1024);
                else glds16(g.Wp + (size_t)(g0_row + 128 * i) * C + g0_chunk + i_p * 32, base + 8192 + (wave + 8 * i) * 1024);
            }
            ++i_p;
            ++issued;
            i_buf = i_buf + 1 == S ? 0 : i_buf + 1;
            return;
        }
        if (PROJ && i_c == 0 && i_j == 0) {
            if (!hold_ok) return;   // asked for again once the tile's LayerNorm2 rows have been written
            hold_ok = false;
        }
        if (i_j < G1S) {
            const u16* w1 = TILED ? g.W1t + (i_c * G1S + i_j) * 12288 + img_lane
                                  : g.W1 + (size_t)(i_c * HC + g0_row) * C + g0_chunk + i_j * G1W;
            const u16* as = TILED && i_tiled ? a_src + i_j * 12288 : a_src + i_j * G1W;
            const int sub_step = TILED && i_tiled ? 4096 : 32;   // elements from one 32-wide slice to the next
#pragma unroll
            for (int sub = 0; sub < 3; ++sub) {
                glds16(as + sub * sub_step, base + sub * 8192 + wave * 1024);
                glds16(w1 + sub * (TILED ? 4096 : 32), base + 24576 + sub * 8192 + wave * 1024);
            }
        } else {
            const u16* w2 = g.W2 + w2_off + i_c * HC + (i_j - G1S) * BK;
#pragma unroll
            for (int i = 0; i < 6; ++i) glds16(w2 + (size_t)(64 * i) * HID, base + (wave + 8 * i) * 1024);
        }
        ++issued;
        i_buf = i_buf + 1 == S ? 0 : i_buf + 1;
        if (++i_j == G1S + G2S) {
            i_j = 0;
            if (++i_c == NCH) {
                i_c = 0;
                i_p = 0;
                i_tile += nslots;
                if (issued < total) set_a(x_start + i_tile);
            }
        }
    };
    set_a(x_start + i_tile);
#pragma unroll
    for (int d = 0; d < S - 1; ++d) issue_one();

    int c_buf = 0, step = 0;
    // `extra`: vector-memory operations (epilogue stores, residual loads) this wave is known to have
    // issued after the stage that was requested last; they may stay in flight across the next S - 1 waits.
    int extra = 0, extra_age = 0;
    // One ring step.  Operations retire in order, so "stage `step` has landed" is "at most what was issued
    // after it is outstanding"; the barrier then covers the other waves' pieces and tells everyone that the
    // slot consumed last step is free for the next stage.
    auto sync_step = [&](bool next_is_g1) -> const char* {
        const int younger = (S > 2 && step + 1 < total) ? (next_is_g1 ? G1L : G2L) : 0;
        const int allow = younger + (step + 1 < total ? extra : 0);
#define EQUSS_W(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(n) : "memory")
        if (allow >= 60) EQUSS_W(60);
        else if (allow >= 24) EQUSS_W(24);
        else EQUSS_W(0);
#undef EQUSS_W
        if (extra_age > 0 && --extra_age == 0) extra = 0;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_one();
        const char* base = smem + c_buf * kStage;
        c_buf = c_buf + 1 == S ? 0 : c_buf + 1;
        ++step;
        return base;
    };

    int c_tile = slot;
    for (int t = 0; t < my_tiles; ++t, c_tile += nslots) {
        const int m0 = (x_start + c_tile) * BM;
        MLP_STAMP();   // tile start
        const uint32_t x_lane = x_lane_of(opaque_lane());
        f32x4 acc2[6][4];
        // The rows of the tile as they stand in acc2: (with_x) fp32 rows back to x; (with_ln) their LayerNorm with
        // the given scale / shift as bf16 rows to `dst`.  Row sums: in-lane, two shuffles, one exchange through
        // red_s between the four waves that share a row; the bf16 rows are re-shaped through the wave's part of
        // the ring slot consumed last (8-byte fragments in, 16-byte row pieces out).  Every address below derives
        // from opaque_lane(), so nothing of it stays in registers between tiles.
        auto rows_out = [&](bool with_x, bool with_ln, const float* gs, const float* bs, const void* dst, int64_t ld_dst, bool tiled) {
            constexpr int ROWSTRIDE = 96 * 2 + 16;
            const int ol = opaque_lane();
            const int fr_e = ol & 15, fq_e = ol >> 4;
            const uint32_t x_lane_e = x_lane_of(ol);
            float s1[4], s2[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                s1[mt] = 0.f; s2[mt] = 0.f;
#pragma unroll
                for (int nt = 0; nt < 6; ++nt) {
                    const f32x4 v = acc2[nt][mt];
                    s1[mt] += (v.x + v.y) + (v.z + v.w);
                    s2[mt] += fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);
                }
                s1[mt] += __shfl_xor(s1[mt], 16, 64); s2[mt] += __shfl_xor(s2[mt], 16, 64);
                s1[mt] += __shfl_xor(s1[mt], 32, 64); s2[mt] += __shfl_xor(s2[mt], 32, 64);
                if (fq_e == 0) equss::lds_store_f32x2(red_s + ((wm * 64 + mt * 16 + fr_e) * WN + wn), s1[mt], s2[mt]);
            }
            // x goes back last: the first LDS load the compiler emits after a global_load_lds is preceded by
            // s_waitcnt vmcnt(0) (the DMA might have written what it reads), and with the 24 stores of x already
            // in the queue that wait would sit out their round trip to memory
            auto store_x = [&]() {
#ifndef EQUSS_ABL_NOSTORE
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const __amdgpu_buffer_rsrc_t x_rs = rows_rsrc(g.x, g.ldx, 4, m0, mt);
#pragma unroll
                    for (int nt = 0; nt < 6; ++nt)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc2[nt][mt]), x_rs, x_lane_e + nt * 64, 0, 0);
                }
#endif
            };
            if (!with_ln) {
                if (with_x) store_x();
                return;
            }
            char* stg = smem + (c_buf == 0 ? S - 1 : c_buf - 1) * kStage + wave * (kStage / NW);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // partial sums visible; that ring slot no longer read
            asm volatile("" ::: "memory");
            // scale / shift of the wave's 96 columns and the four partial sums of its rows: one batch of LDS reads, one wait
            L4 gau[6], beu[6], pru[4][2];
            {
                const uint32_t sg = equss::lds_addr(gs) + (wn * 96 + fq_e * 4) * 4, sb = equss::lds_addr(bs) + (wn * 96 + fq_e * 4) * 4;
                const uint32_t sp = equss::lds_addr(red_s) + (wm * 64 + fr_e) * WN * 8;
                pru[0][0] = LD<0>(sp); pru[0][1] = LD<16>(sp);
                gau[0] = LD<0>(sg); gau[1] = LD<64>(sg); gau[2] = LD<128>(sg); gau[3] = LD<192>(sg); gau[4] = LD<256>(sg); gau[5] = LD<320>(sg);
                beu[0] = LD<0>(sb); beu[1] = LD<64>(sb); beu[2] = LD<128>(sb); beu[3] = LD<192>(sb); beu[4] = LD<256>(sb); beu[5] = LD<320>(sb);
                pru[1][0] = LD<16 * WN * 8>(sp); pru[1][1] = LD<16 * WN * 8 + 16>(sp);
                pru[2][0] = LD<32 * WN * 8>(sp); pru[2][1] = LD<32 * WN * 8 + 16>(sp);
                pru[3][0] = LD<48 * WN * 8>(sp); pru[3][1] = LD<48 * WN * 8 + 16>(sp);
                equss::lds_wait_for<0>(gau[0], gau[1], gau[2], gau[3], gau[4], gau[5], beu[0], beu[1], beu[2], beu[3], beu[4], beu[5],
                                       pru[0][0], pru[0][1], pru[1][0], pru[1][1], pru[2][0], pru[2][1], pru[3][0], pru[3][1]);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 q0 = __builtin_bit_cast(f32x4, pru[mt][0]), q1 = __builtin_bit_cast(f32x4, pru[mt][1]);   // (sum, sum of squares) x 2
                const float mean = ((q0.x + q0.z) + (q1.x + q1.z)) * (1.0f / C);
                const float ex2 = ((q0.y + q0.w) + (q1.y + q1.w)) * (1.0f / C);
                const float rstd = 1.0f / sqrtf(fmaxf(fmaf(-mean, mean, ex2), 0.f) + g.eps);
#pragma unroll
                for (int nt = 0; nt < 6; ++nt) {
                    const f32x4 ga = __builtin_bit_cast(f32x4, gau[nt]), be = __builtin_bit_cast(f32x4, beu[nt]);
                    const f32x4 v = acc2[nt][mt];
                    const uint32_t lo = equss::pack_bf16(fmaf((v.x - mean) * rstd, ga.x, be.x), fmaf((v.y - mean) * rstd, ga.y, be.y));
                    const uint32_t hi = equss::pack_bf16(fmaf((v.z - mean) * rstd, ga.z, be.z), fmaf((v.w - mean) * rstd, ga.w, be.w));
                    equss::lds_store_b64(stg + fr_e * ROWSTRIDE + (nt * 16 + fq_e * 4) * 2, lo, hi);
                }
                asm volatile("" ::: "memory");
                // (tiled: one resource over the tile's 96 KiB, the 16-row group as the scalar offset)
                const __amdgpu_buffer_rsrc_t out_rs = TILED && tiled ? rows_rsrc(dst, C, 2, m0, 0) : rows_rsrc(dst, ld_dst, 2, m0, mt);
#pragma unroll
                for (int j = 0; j < PER_LANE; ++j) {
                    const u32x4 d = *reinterpret_cast<const u32x4*>(stg + stg_off_of(ol, j));
#ifndef EQUSS_ABL_NOSTORE
                    if (TILED && tiled) __builtin_amdgcn_raw_buffer_store_b128(d, out_rs, out_tiled_of(ol, j), mt * 1024, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(d, out_rs, out_lane_of(ol, j, ld_dst), 0, 0);
#else
                    asm volatile("" :: "v"(d));
#endif
                }
                asm volatile("" ::: "memory");
            }
            if (with_x) store_x();
        };   // [nt][mt]: rows wm*64 + mt*16 + fr, columns wn*96 + nt*16 + fq*4 + 0..3
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const __amdgpu_buffer_rsrc_t x_rs = rows_rsrc(g.x, g.ldx, 4, m0, mt);
#pragma unroll
            for (int nt = 0; nt < 6; ++nt) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#ifndef EQUSS_ABL_NOXLOAD   // (EQUSS_ABL_*: developer builds that leave one cost out, tools/build_ablations.sh; wrong results by design)
                o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, x_lane + nt * 64, 0, 0));
#endif
                acc2[nt][mt] = o;   // (+ bias before the first use below: the loads get the six G1 steps to arrive)
            }
        }
        // residual loads (and, after the first tile, the previous epilogue's stores) of a full tile are a
        // known number of operations younger than the stage in flight
        const bool full = m0 + BM <= g.M;
        extra = full ? 24 + (t > 0 ? 24 + (g.ln ? 12 : 0) : 0) : 0;
        extra_age = S - 1;
        MLP_STAMP();   // residual loads issued
        if (PROJ) {
            // ---- G0: acc2 = x + bp + att Wp^T  (the projection with its residual, vision_transformer.py:110)
            for (int kt = 0; kt < G0S; ++kt) {
                const char* base = sync_step(false);
                if (kt == 0) {
#pragma unroll
                    for (int nt = 0; nt < 6; ++nt) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(biasp_s + wn * 96 + nt * 16 + fq * 4);
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            acc2[nt][mt].x += b4.x; acc2[nt][mt].y += b4.y; acc2[nt][mt].z += b4.z; acc2[nt][mt].w += b4.w;
                        }
                    }
#ifdef EQUSS_MLP_STAMPS
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (changes the timing a little: the ring's next stage is waited for too)
#endif
                    MLP_STAMP();   // residual rows have arrived
                }
                {
                    const uint32_t sa = equss::lds_addr(base) + (wm * 64 + fr) * 64 + g0_off;
                    const uint32_t sw = equss::lds_addr(base) + 8192 + (wn * 96 + fr) * 64 + g0_off;
                    L4 a[4], w[6];
                    a[0] = LD<0>(sa); a[1] = LD<1024>(sa); a[2] = LD<2048>(sa); a[3] = LD<3072>(sa);
                    w[0] = LD<0>(sw); w[1] = LD<1024>(sw); w[2] = LD<2048>(sw);
                    w[3] = LD<3072>(sw); w[4] = LD<4096>(sw); w[5] = LD<5120>(sw);
                    __builtin_amdgcn_s_setprio(1);
                    equss::lds_wait_for<3>(a[0], a[1], a[2], a[3], w[0], w[1], w[2]);
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) acc2[nt][mt] = MM(w[nt], a[mt], acc2[nt][mt]);
                    equss::lds_wait_for<0>(w[3], w[4], w[5]);
#pragma unroll
                    for (int nt = 3; nt < 6; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) acc2[nt][mt] = MM(w[nt], a[mt], acc2[nt][mt]);
                    __builtin_amdgcn_s_setprio(0);
                }
                asm volatile("" ::: "memory");
            }
            MLP_STAMP();   // projection steps done
            // ---- LayerNorm2 of the updated rows -> memory (g.A), from where the G1 steps stage it ------------
            {
                rows_out(false, true, gamma2_s, beta2_s, g.A, g.lda, full);
                // Workgroup-scope release / acquire: the stores have completed, every wave of the workgroup
                // (one CU, one vector L1) is past the barrier, then the rows are read back by LDS-DMA.
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                hold_ok = true;
                issue_one();
            }
            MLP_STAMP();   // LayerNorm2 rows written, first G1 stage requested
        }
        for (int c = 0; c < NCH; ++c) {
            // ---- G1: h = a W1_c^T -------------------------------------------------------------------
            f32x4 acc1[2][4];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc1[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kt = 0; kt < G1S; ++kt) {
                const char* base = sync_step(kt + 1 < G1S);
                {   // three 32-wide sub-tiles: the reads of sub-tile s + 1 are in flight during the MFMAs of sub-tile s
                    const uint32_t sa = equss::lds_addr(base) + (wm * 64 + fr) * 64 + g0_off;
                    const uint32_t sw = equss::lds_addr(base) + 24576 + (wn * 32 + fr) * 64 + g0_off;
                    L4 f0[6], f1[6], f2[6];   // four row fragments of a, two of W1
#define EQUSS_G1_LOAD(f, sub)                                                                                      \
                    f[0] = LD<sub * 8192>(sa); f[1] = LD<sub * 8192 + 1024>(sa); f[2] = LD<sub * 8192 + 2048>(sa); \
                    f[3] = LD<sub * 8192 + 3072>(sa); f[4] = LD<sub * 8192>(sw); f[5] = LD<sub * 8192 + 1024>(sw);
#define EQUSS_G1_MMS(f)                                                      \
                    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)         \
                        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) acc1[nt][mt] = MM(f[4 + nt], f[mt], acc1[nt][mt]);
                    EQUSS_G1_LOAD(f0, 0)
                    EQUSS_G1_LOAD(f1, 1)
                    __builtin_amdgcn_s_setprio(1);
                    equss::lds_wait_for<6>(f0[0], f0[1], f0[2], f0[3], f0[4], f0[5]);
                    EQUSS_G1_MMS(f0)
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    EQUSS_G1_LOAD(f2, 2)
                    __builtin_amdgcn_s_setprio(1);
                    equss::lds_wait_for<6>(f1[0], f1[1], f1[2], f1[3], f1[4], f1[5]);
                    EQUSS_G1_MMS(f1)
                    equss::lds_wait_for<0>(f2[0], f2[1], f2[2], f2[3], f2[4], f2[5]);
                    EQUSS_G1_MMS(f2)
                    __builtin_amdgcn_s_setprio(0);
#undef EQUSS_G1_LOAD
#undef EQUSS_G1_MMS
                }
                asm volatile("" ::: "memory");
            }
            MLP_STAMP();   // G1 steps done
            // ---- gelu(h + b1) -> LDS as two 64-wide K stages in the ring's row format -----------------
            // (every wave passed a barrier after its last read of the previous chunk's h)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int nloc = wn * 32 + nt * 16 + fq * 4;          // column within the chunk
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias1_s + c * HC + nloc);
                const int half = nloc >> 6, n6 = nloc & 63;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int row = wm * 64 + mt * 16 + fr;
                    f32x4 v = acc1[nt][mt];
                    v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
#ifndef EQUSS_ABL_NOGELU
                    const equss::gelu_f32x2 g0 = equss::gelu_erf2(equss::gelu_f32x2{v.x, v.y});
                    const equss::gelu_f32x2 g1 = equss::gelu_erf2(equss::gelu_f32x2{v.z, v.w});
#else
                    const equss::gelu_f32x2 g0 = {v.x, v.y}, g1 = {v.z, v.w};
#endif
                    uint2 pk;
                    pk.x = equss::pack_bf16(g0.x, g0.y);
                    pk.y = equss::pack_bf16(g1.x, g1.y);
                    equss::lds_store_b64(hc + half * 16384 + row * 128 + (((n6 >> 3) ^ ((row >> 1) & 7)) * 16) + ((n6 >> 2) & 1) * 8,
                                         pk.x, pk.y);
                }
            }
            if (c == 0) {
#pragma unroll
                for (int nt = 0; nt < 6; ++nt) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias2_s + wn * 96 + nt * 16 + fq * 4);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        acc2[nt][mt].x += b4.x; acc2[nt][mt].y += b4.y; acc2[nt][mt].z += b4.z; acc2[nt][mt].w += b4.w;
                    }
                }
            }
            MLP_STAMP();   // GELU done
            // ---- G2: acc2 += h W2[:, c]^T  (the step's wait + barrier publishes h) ---------------------
            for (int kt = 0; kt < G2S; ++kt) {
                const char* base = sync_step(kt + 1 == G2S);
                const char* hb = hc + kt * 16384;
                {   // two 32-wide sub-steps in four groups of twelve MFMAs; every fragment is requested at least one group
                    // before its first use, the second sub-step's weight fragments into the registers a group has finished with
                    const uint32_t sh0 = equss::lds_addr(hb) + a_row + c_off[0], sh1 = equss::lds_addr(hb) + a_row + c_off[1];
                    const uint32_t sw0 = equss::lds_addr(base) + w2_row + c_off[0], sw1 = equss::lds_addr(base) + w2_row + c_off[1];
                    L4 h0[4], h1[4], wa[3], wb[3], wc[3], wd[3];
                    h0[0] = LD<0>(sh0); h0[1] = LD<2048>(sh0); h0[2] = LD<4096>(sh0); h0[3] = LD<6144>(sh0);
                    wa[0] = LD<0>(sw0); wa[1] = LD<2048>(sw0); wa[2] = LD<4096>(sw0);
                    wb[0] = LD<6144>(sw0); wb[1] = LD<8192>(sw0); wb[2] = LD<10240>(sw0);
                    h1[0] = LD<0>(sh1); h1[1] = LD<2048>(sh1); h1[2] = LD<4096>(sh1); h1[3] = LD<6144>(sh1);
#define EQUSS_G2_MMS(W, H, NT0)                                              \
                    _Pragma("unroll") for (int nt = 0; nt < 3; ++nt)         \
                        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) acc2[NT0 + nt][mt] = MM(W[nt], H[mt], acc2[NT0 + nt][mt]);
                    __builtin_amdgcn_s_setprio(1);
                    equss::lds_wait_for<7>(h0[0], h0[1], h0[2], h0[3], wa[0], wa[1], wa[2]);
                    EQUSS_G2_MMS(wa, h0, 0)
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    wc[0] = LD<0>(sw1); wc[1] = LD<2048>(sw1); wc[2] = LD<4096>(sw1);
                    __builtin_amdgcn_s_setprio(1);
                    equss::lds_wait_for<7>(wb[0], wb[1], wb[2]);
                    EQUSS_G2_MMS(wb, h0, 3)
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    wd[0] = LD<6144>(sw1); wd[1] = LD<8192>(sw1); wd[2] = LD<10240>(sw1);
                    __builtin_amdgcn_s_setprio(1);
                    equss::lds_wait_for<3>(h1[0], h1[1], h1[2], h1[3], wc[0], wc[1], wc[2]);
                    EQUSS_G2_MMS(wc, h1, 0)
                    equss::lds_wait_for<0>(wd[0], wd[1], wd[2]);
                    EQUSS_G2_MMS(wd, h1, 3)
                    __builtin_amdgcn_s_setprio(0);
#undef EQUSS_G2_MMS
                }
                asm volatile("" ::: "memory");
            }
            MLP_STAMP();   // G2 steps done
        }
        // ---- epilogue: x back, LayerNorm statistics, normalised rows as bf16 (as gemm8's residual epilogue) --
        // (the slot consumed by the last step is free: the other one is receiving the next tile's first stage;
        // red_s is rewritten by the next tile only after ~100 barriers)
        rows_out(true, g.ln != nullptr, gamma_s, beta_s, g.ln, g.ld_ln, false);
        MLP_STAMP();   // epilogue issued
    }
#ifdef EQUSS_MLP_STAMPS
    __syncthreads();
    if (blockIdx.x == 77)
        for (int i = tid; i < NW * kStampMax; i += kThreads) g_mlp_stamps[i] = stamps_s[i];
#endif
}

// Stage images of a block's projection and fc1 weights (see mlp_kernel<true, true>): one thread per 16 bytes.
//   wpt: [step 0..11][piece 0..23][64 lanes][8]   <- Wp rows piece*16 + lane/4, columns step*32 + chunk*8
//   w1t: [chunk 0..11][step 0..3][slice 0..2][piece 0..7][64 lanes][8]  <- W1 rows chunk*128 + piece*16 + lane/4,
//        columns step*96 + slice*32 + chunk*8;       chunk = (lane & 3) ^ swz4(lane >> 4), the LDS chunk swizzle
//   w2t: [chunk 0..11][step 0..3][piece 0..23][64 lanes][8]  <- W2 rows piece*16 + lane/4, columns chunk*128 + step*32 + chunk*8
//        (32-wide slices of W2: what block_ws.hip stages; mlp_kernel reads W2 row-major in 64-wide slices)
constexpr int kWptUnits = G0S * 24 * 64, kW1tUnits = NCH * G1S * 3 * 8 * 64, kW2tUnits = NCH * 4 * 24 * 64;
struct PackArgs { const u16* wp[16]; const u16* w1[16]; const u16* w2[16]; u16* out; size_t stride; };   // stride: elements between blocks
__global__ __launch_bounds__(256) void pack_tail_weights_kernel(const PackArgs a) {
    const int blk = blockIdx.y;
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= kWptUnits + kW1tUnits + kW2tUnits) return;
    const int l = u & 63, row16 = l >> 2, chunk = ((l & 3) ^ ((0x78 >> (2 * (l >> 4))) & 3)) * 8;
    const u16* src;
    if (u < kWptUnits) {
        const int piece = (u >> 6) % 24, step = (u >> 6) / 24;
        src = a.wp[blk] + (size_t)(piece * 16 + row16) * C + step * 32 + chunk;
    } else if (u < kWptUnits + kW1tUnits) {
        const int v = (u - kWptUnits) >> 6, piece = v & 7, slice = (v >> 3) % 3, step = (v / 24) % G1S, ch = v / (24 * G1S);
        src = a.w1[blk] + (size_t)(ch * HC + piece * 16 + row16) * C + step * G1W + slice * 32 + chunk;
    } else {
        const int v = (u - kWptUnits - kW1tUnits) >> 6, piece = v % 24, step = (v / 24) & 3, ch = v / 96;
        src = a.w2[blk] + (size_t)(piece * 16 + row16) * HID + ch * HC + step * 32 + chunk;
    }
    u16* dst = a.out + (size_t)blk * a.stride + (size_t)u * 8;
    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
}

}  // namespace

namespace equss {

// Bytes of the stage images of one block's projection + fc1 + fc2 weights, and the launch that writes them for `depth`
// blocks (depth <= 16), block i at out + i * stride bytes: the Wp image, then the W1 image, then the W2 image.
size_t tail_weight_image_bytes() { return (size_t)(kWptUnits + kW1tUnits + kW2tUnits) * 16; }
size_t tail_weight_wp_bytes() { return (size_t)kWptUnits * 16; }
size_t tail_weight_w1_bytes() { return (size_t)kW1tUnits * 16; }
int pack_tail_weights(const void* const* wp, const void* const* w1, const void* const* w2, int depth, void* out, size_t stride, hipStream_t s) {
    if (depth < 1 || depth > 16) return fail(EQUSS_E_UNSUPPORTED, "pack_tail_weights: depth=%d (1..16)", depth);
    PackArgs a;
    for (int i = 0; i < depth; ++i) {
        if (!wp[i] || !w1[i] || !w2[i] || ((uintptr_t)wp[i] % 16) || ((uintptr_t)w1[i] % 16) || ((uintptr_t)w2[i] % 16))
            return fail(EQUSS_E_ALIGN, "pack_tail_weights: weights must be non-null and 16-byte aligned");
        a.wp[i] = reinterpret_cast<const u16*>(wp[i]);
        a.w1[i] = reinterpret_cast<const u16*>(w1[i]);
        a.w2[i] = reinterpret_cast<const u16*>(w2[i]);
    }
    a.out = reinterpret_cast<u16*>(out); a.stride = stride / 2;   // (bytes between blocks, >= tail_weight_image_bytes())
    hipLaunchKernelGGL(pack_tail_weights_kernel, dim3((kWptUnits + kW1tUnits + kW2tUnits + 255) / 256, depth), dim3(256), 0, s, a);
    return check_launch("pack_tail_weights_kernel");
}

bool mlp_fused_supports(int C_, int hidden) { return C_ == C && hidden == HID; }

namespace {
template <bool PROJ, bool TILED>
int launch_mlp(const MlpArgs& g, hipStream_t s) {
    if (!lds_opt_in(reinterpret_cast<const void*>(mlp_kernel<PROJ, TILED>), kLds))
        return fail(EQUSS_E_UNSUPPORTED, "mlp_fused: cannot reserve %d bytes of LDS", kLds);
    const int ntiles = (g.M + BM - 1) / BM;
    const int grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL((mlp_kernel<PROJ, TILED>), dim3(grid), dim3(kThreads), kLds, s, g);
    return check_launch("mlp_kernel");
}
}  // namespace

int mlp_fused(const void* a, int64_t lda, const void* w1, const float* b1, const void* w2, const float* b2, float* x,
              int64_t ldx, const float* gamma, const float* beta, void* ln, int64_t ld_ln, int M, float eps, hipStream_t s) {
    MlpArgs g;
    g.A0 = nullptr; g.lda0 = 0; g.Wp = nullptr; g.bp = nullptr; g.gamma2 = nullptr; g.beta2 = nullptr;
    g.A = reinterpret_cast<const u16*>(a); g.lda = lda;
    g.W1 = reinterpret_cast<const u16*>(w1); g.b1 = b1;
    g.W2 = reinterpret_cast<const u16*>(w2); g.b2 = b2;
    g.x = x; g.ldx = ldx; g.gamma = gamma; g.beta = beta;
    g.ln = reinterpret_cast<u16*>(ln); g.ld_ln = ld_ln; g.eps = eps; g.M = M;
    g.Wpt = nullptr; g.W1t = nullptr;
    return launch_mlp<false, false>(g, s);
}

// The rest of a transformer block after attention in one kernel (vision_transformer.py:110-111):
//   x += att Wp^T + bp ;  ln_mid = LayerNorm2(x) ;  x += fc2(gelu(fc1(ln_mid))) ;  ln = LayerNorm(x)
// ln_mid (M, 384) bf16 is scratch in memory (written and read back by the same workgroup); ln may be the
// same buffer, or null.
int block_fused(const void* att, int64_t lda0, const void* wp, const float* bp, const float* gamma2, const float* beta2,
                void* ln_mid, int64_t ld_mid, const void* w1, const float* b1, const void* w2, const float* b2, float* x,
                int64_t ldx, const float* gamma, const float* beta, void* ln, int64_t ld_ln, int M, float eps, hipStream_t s,
                const void* wp_image, const void* w1_image) {
    MlpArgs g;
    g.A0 = reinterpret_cast<const u16*>(att); g.lda0 = lda0;
    g.Wp = reinterpret_cast<const u16*>(wp); g.bp = bp; g.gamma2 = gamma2; g.beta2 = beta2;
    g.A = reinterpret_cast<const u16*>(ln_mid); g.lda = ld_mid;
    g.W1 = reinterpret_cast<const u16*>(w1); g.b1 = b1;
    g.W2 = reinterpret_cast<const u16*>(w2); g.b2 = b2;
    g.x = x; g.ldx = ldx; g.gamma = gamma; g.beta = beta;
    g.ln = reinterpret_cast<u16*>(ln); g.ld_ln = ld_ln; g.eps = eps; g.M = M;
    // stage images of the weights (pack_tail_weights), if the caller has them: ln_mid is then written in stage order too
    g.Wpt = reinterpret_cast<const u16*>(wp_image); g.W1t = reinterpret_cast<const u16*>(w1_image);
    if (wp_image && w1_image && ld_mid == C) return launch_mlp<true, true>(g, s);
    return launch_mlp<true, false>(g, s);
}

}  // namespace equss

// x += fc2(gelu(fc1(a))), ln_out = LayerNorm(x) (ln_out may be null).  C = 384, hidden = 1536 only.
extern "C" int equss_mlp_resid_ln_bf16(const void* a, int64_t lda, const void* w1, const float* b1, const void* w2,
                                       const float* b2, float* x, int64_t ldx, const float* gamma, const float* beta,
                                       void* ln_out, int64_t ld_ln, int M, int C_, int hidden, float eps, void* stream) {
    if (!a || !w1 || !b1 || !w2 || !b2 || !x || (ln_out && (!gamma || !beta)))
        return equss::fail(EQUSS_E_NULL, "mlp_resid_ln: null pointer argument");
    if (M <= 0) return equss::fail(EQUSS_E_SHAPE, "mlp_resid_ln: M=%d must be positive", M);
    if (!equss::mlp_fused_supports(C_, hidden))
        return equss::fail(EQUSS_E_UNSUPPORTED, "mlp_resid_ln: C=%d hidden=%d (384 and 1536 only)", C_, hidden);
    // (row offsets inside a tile are 32-bit byte offsets of buffer instructions)
    if (lda > (1 << 22) || ldx > (1 << 22) || ld_ln > (1 << 22))
        return equss::fail(EQUSS_E_UNSUPPORTED, "mlp_resid_ln: leading dimensions above 2^22 elements are not supported");
    if (lda < C_ || ldx < C_ || (ln_out && ld_ln < C_) || (lda % 8) || (ldx % 4) || (ld_ln % 8) || ((uintptr_t)a % 16) ||
        ((uintptr_t)w1 % 16) || ((uintptr_t)w2 % 16) || ((uintptr_t)x % 16) || ((uintptr_t)ln_out % 16))
        return equss::fail(EQUSS_E_ALIGN, "mlp_resid_ln: operands must be 16-byte aligned with padded leading dimensions");
    return equss::mlp_fused(a, lda, w1, b1, w2, b2, x, ldx, gamma, beta, ln_out, ld_ln, M, eps,
                            reinterpret_cast<hipStream_t>(stream));
}

// x += att Wp^T + bp; ln_mid = LayerNorm2(x); x += fc2(gelu(fc1(ln_mid))); ln_out = LayerNorm(x) (or null).
extern "C" int equss_block_tail_bf16(const void* att, int64_t ld_att, const void* wp, const float* bp, const float* gamma2,
                                     const float* beta2, void* ln_mid, int64_t ld_mid, const void* w1, const float* b1,
                                     const void* w2, const float* b2, float* x, int64_t ldx, const float* gamma,
                                     const float* beta, void* ln_out, int64_t ld_ln, int M, int C_, int hidden, float eps,
                                     void* stream) {
    if (!att || !wp || !bp || !gamma2 || !beta2 || !ln_mid || !w1 || !b1 || !w2 || !b2 || !x || (ln_out && (!gamma || !beta)))
        return equss::fail(EQUSS_E_NULL, "block_tail: null pointer argument");
    if (M <= 0) return equss::fail(EQUSS_E_SHAPE, "block_tail: M=%d must be positive", M);
    if (!equss::mlp_fused_supports(C_, hidden))
        return equss::fail(EQUSS_E_UNSUPPORTED, "block_tail: C=%d hidden=%d (384 and 1536 only)", C_, hidden);
    if (ld_att > (1 << 22) || ld_mid > (1 << 22) || ldx > (1 << 22) || ld_ln > (1 << 22))
        return equss::fail(EQUSS_E_UNSUPPORTED, "block_tail: leading dimensions above 2^22 elements are not supported");
    if (ld_att < C_ || ld_mid < C_ || ldx < C_ || (ln_out && ld_ln < C_) || (ld_att % 8) || (ld_mid % 8) || (ldx % 4) ||
        (ld_ln % 8) || ((uintptr_t)att % 16) || ((uintptr_t)wp % 16) || ((uintptr_t)ln_mid % 16) || ((uintptr_t)w1 % 16) ||
        ((uintptr_t)w2 % 16) || ((uintptr_t)x % 16) || ((uintptr_t)ln_out % 16))
        return equss::fail(EQUSS_E_ALIGN, "block_tail: operands must be 16-byte aligned with padded leading dimensions");
    return equss::block_fused(att, ld_att, wp, bp, gamma2, beta2, ln_mid, ld_mid, w1, b1, w2, b2, x, ldx, gamma, beta, ln_out,
                              ld_ln, M, eps, reinterpret_cast<hipStream_t>(stream), nullptr, nullptr);
}
