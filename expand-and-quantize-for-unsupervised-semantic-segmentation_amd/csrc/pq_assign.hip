// Product-quantisation assign for gfx950: per sub-space nearest-codeword search in exact fp32.
//
// Follows Codebook.forward / ProductQuantizerWrapper.forward (model/dino_pqgo.py:579-705,:745-776)
// and the same core of model/quantizer.py (:120-153, :419-476): normalise rows and codebook,
// dist = (sum zn^2 + sum cn^2) - 2 <zn, cn>, first arg-min, gather, straight-through output and
// the mse numerator.
//
// Mapping to the machine
//  * One workgroup serves one sub-space p and walks a strided list of row tiles.  Its waves pin
//    the (normalised) codebook of p in registers for the whole launch: wave `wk` holds NT tiles
//    of 32 codewords as the A operand of v_mfma_f32_32x32x2_f32, which is an exact fp32 fma chain
//    (k ascending), so no precision is traded for the matrix pipe.
//  * Rows stream HBM -> registers (coalesced 16 B per lane) -> normalise (lanes that share a row
//    reduce with wave shuffles) -> XOR-swizzled LDS tile.  Every wave reads the tile back as the B
//    operand with conflict-free ds_read_b128.
//  * The MFMA result puts a row on each lane and 16 codewords in its registers, so the arg-min is
//    an in-lane scan, one cross-half shuffle, and (when several waves split K) a tiny LDS pass.
//  * The epilogue re-reads the normalised row from LDS, gathers the chosen codeword (L2 resident)
//    and writes zq / idx coalesced; per-workgroup loss partials are summed in a fixed order by a
//    second tiny kernel, so the loss is bitwise reproducible run to run.
#include "equss_common.h"
#include <mutex>

namespace {

using equss::group_sum;

constexpr int kMaxGridX = 1024;

template <int D>
__device__ __forceinline__ int swz_chunk(int r, int c) {
    constexpr int CPR = D / 4;  // 16-byte chunks per row
    if constexpr (CPR >= 16) {
        return c ^ (r & 15);
    } else {
        constexpr int RPB = 16 / CPR;  // rows per 256-byte bank row
        return c ^ ((r / RPB) & (CPR - 1));
    }
}

// Normalise one row held as one float4 per lane over D/4 consecutive lanes.
// Returns sum(v_normalised^2) (the "sum z^2" term of the distance).
template <int D>
__device__ __forceinline__ float normalize_row(f32x4& v, int mode) {
    constexpr int LPR = D / 4;
    if (mode == EQUSS_NORM_L2) {  // F.normalize(dim=1, eps=1e-12): x / max(||x||, eps)
        float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        s = group_sum<LPR>(s);
        float den = fmaxf(sqrtf(s), 1e-12f);
        v.x = v.x / den; v.y = v.y / den; v.z = v.z / den; v.w = v.w / den;
    } else if (mode == EQUSS_NORM_ZSCORE) {  // (x - mean) / (unbiased std + 1e-5)
        float m = group_sum<LPR>(v.x + v.y + v.z + v.w) / (float)D;
        v.x -= m; v.y -= m; v.z -= m; v.w -= m;
        float q = group_sum<LPR>(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
        float den = sqrtf(q / (float)(D - 1)) + 1e-5f;
        v.x = v.x / den; v.y = v.y / den; v.z = v.z / den; v.w = v.w / den;
    }
    float a = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    return group_sum<LPR>(a);
}

// cn = normalise(codebook rows), csq = sum(cn^2) per row.
template <int D>
__global__ void pq_prep_codebook(const float* __restrict__ cb, float* __restrict__ cn,
                                 float* __restrict__ csq, int total_rows, int mode) {
    constexpr int LPR = D / 4;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = gid / LPR, c = gid % LPR;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < total_rows) v = *reinterpret_cast<const f32x4*>(cb + (size_t)row * D + 4 * c);
    const float a = normalize_row<D>(v, mode);
    if (row < total_rows) {
        *reinterpret_cast<f32x4*>(cn + (size_t)row * D + 4 * c) = v;
        if (c == 0) csq[row] = a;
    }
}

struct PqArgs {
    const float* code; int64_t ld_code;
    const float* cn; const float* csq; const float* gather;  // gather: raw or normalised codebooks
    int64_t* idx; float* zq; int64_t ld_zq; float* loss_partial;
    int rows, num_pq, K, ntiles, mode, WK, WR, RT;
};

constexpr int kPqThreads = 512;   // 8 waves: two per SIMD, one workgroup per CU at <= 256 VGPRs
constexpr int kPqMaxItems = 8;    // float4 items a thread stages per row tile (upper bound)

template <int D, int NT, int ITEMS>
__global__ __launch_bounds__(kPqThreads, 2) void pq_assign_kernel(const PqArgs a) {
    constexpr int LPR = D / 4;
    constexpr int THREADS = kPqThreads;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int WK = a.WK, RT = a.RT, K = a.K;
    const int R = 32 * a.WR * RT;
    const int wk = wave % WK, wr = wave / WK;
    const int p = blockIdx.y;
    const int li = lane & 31, lh = lane >> 5;

    float* zt = reinterpret_cast<float*>(smem);  // R x D, chunk-swizzled
    float* rowA = zt + (size_t)R * D;            // R
    float* csq_s = rowA + R;                     // K
    float* cand_d = csq_s + K;                   // WK x R
    int* cand_i = reinterpret_cast<int*>(cand_d + (size_t)WK * R);
    int* fin_idx = cand_i + (size_t)WK * R;      // R

    // Codebook slice of this wave -> registers, laid out as MFMA A operands: lane (i, h) keeps
    // cn[cw0 + 32 nt + i][8 t + 4 h + u] in cbr[nt][4 t + u]; the B operand uses the same k order.
    const int cw0 = wk * 32 * NT;
    float cbr[NT][D / 2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.cn + ((size_t)p * K + cw0 + nt * 32 + li) * D);
#pragma unroll
        for (int t = 0; t < D / 8; ++t) {
            const f32x4 c = src[2 * t + lh];
            cbr[nt][4 * t + 0] = c.x; cbr[nt][4 * t + 1] = c.y;
            cbr[nt][4 * t + 2] = c.z; cbr[nt][4 * t + 3] = c.w;
        }
    }
    for (int k = tid; k < K; k += THREADS) csq_s[k] = a.csq[(size_t)p * K + k];

    float loss = 0.f;

    // Rows of the next tile travel HBM -> registers while the current tile is on the matrix pipe.
    f32x4 pre[ITEMS];  // ITEMS = R * LPR / THREADS float4 per thread per tile
    auto fetch = [&](int tile) {
        const int row0 = tile * R;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int item = it * THREADS + tid;
            const int grow = row0 + item / LPR;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (tile < a.ntiles && grow < a.rows)
                v = *reinterpret_cast<const f32x4*>(a.code + (size_t)grow * a.ld_code + (size_t)p * D +
                                                    4 * (item % LPR));
            pre[it] = v;
        }
    };
    fetch(blockIdx.x);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int row0 = tile * R;
        // ---- stage: normalise the fetched rows -> LDS ----------------------------------------
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int item = it * THREADS + tid;
            const int r = item / LPR, c = item % LPR;
            f32x4 v = pre[it];
            const float sa = normalize_row<D>(v, a.mode);
            *reinterpret_cast<f32x4*>(zt + ((size_t)r * LPR + swz_chunk<D>(r, c)) * 4) = v;
            if (c == 0) rowA[r] = sa;
        }
        __syncthreads();
        fetch(tile + gridDim.x);
        // ---- distances on the matrix pipe + in-lane arg-min ---------------------------------
        for (int rt = 0; rt < RT; ++rt) {
            const int row = (wr * RT + rt) * 32 + li;
            const float* zrow = zt + (size_t)row * D;
            const float arow = rowA[row];
            float best = __builtin_inff();
            int besti = 0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                // one 16-byte chunk of the row feeds four MFMAs; the next chunk is read while they
                // issue, and the scheduling fence keeps the compiler from hoisting every read of the
                // row (64 registers at d = 128) to the top.
                f32x4 z = *reinterpret_cast<const f32x4*>(zrow + swz_chunk<D>(row, lh) * 4);
#pragma unroll
                for (int t = 0; t < D / 8; ++t) {
                    f32x4 zn = z;
                    if (t + 1 < D / 8)
                        zn = *reinterpret_cast<const f32x4*>(zrow + swz_chunk<D>(row, 2 * (t + 1) + lh) * 4);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cbr[nt][4 * t + 0], z.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cbr[nt][4 * t + 1], z.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cbr[nt][4 * t + 2], z.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cbr[nt][4 * t + 3], z.w, acc, 0, 0, 0);
                    z = zn;
                    __builtin_amdgcn_sched_barrier(0);
                }
                // acc[reg] = <cn[cw], zn[row]> with cw = cw0 + 32 nt + (reg&3) + 8 (reg>>2) + 4 h
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int cw = cw0 + nt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                    const float t0 = arow + csq_s[cw];
                    const float dist = t0 - 2.0f * acc[reg];
                    if (dist < best) { best = dist; besti = cw; }
                }
            }
            const float od = __shfl_xor(best, 32, 64);
            const int oi = __shfl_xor(besti, 32, 64);
            if (od < best || (od == best && oi < besti)) { best = od; besti = oi; }
            if (lh == 0) { cand_d[wk * R + row] = best; cand_i[wk * R + row] = besti; }
        }
        __syncthreads();
        // ---- first minimum across the waves that split K -------------------------------------
        if (tid < R) {
            float b = cand_d[tid];
            int bi = cand_i[tid];
            for (int w = 1; w < WK; ++w) {
                const float d2 = cand_d[w * R + tid];
                if (d2 < b) { b = d2; bi = cand_i[w * R + tid]; }
            }
            fin_idx[tid] = bi;
            if (row0 + tid < a.rows) a.idx[(size_t)p * a.rows + row0 + tid] = (int64_t)bi;
        }
        __syncthreads();
        // ---- gather + straight-through output + mse numerator --------------------------------
        for (int it = 0; it < ITEMS; ++it) {
            const int item = it * THREADS + tid;
            const int r = item / LPR, c = item % LPR;
            const int grow = row0 + r;
            if (grow < a.rows) {
                const int k = fin_idx[r];
                const f32x4 zn = *reinterpret_cast<const f32x4*>(zt + ((size_t)r * LPR + swz_chunk<D>(r, c)) * 4);
                const f32x4 e = *reinterpret_cast<const f32x4*>(a.gather + ((size_t)p * K + k) * D + 4 * c);
                f32x4 df; df.x = e.x - zn.x; df.y = e.y - zn.y; df.z = e.z - zn.z; df.w = e.w - zn.w;
                loss += df.x * df.x + df.y * df.y + df.z * df.z + df.w * df.w;
                if (a.zq) {
                    f32x4 q; q.x = zn.x + df.x; q.y = zn.y + df.y; q.z = zn.z + df.z; q.w = zn.w + df.w;
                    *reinterpret_cast<f32x4*>(a.zq + (size_t)grow * a.ld_zq + (size_t)p * D + 4 * c) = q;
                }
            }
        }
        __syncthreads();
    }
    // ---- per-workgroup loss partial (fixed order, no atomics) ----------------------------------
    loss = group_sum<64>(loss);
    if (lane == 0) cand_d[wave] = loss;  // cand_d is free after the last barrier
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < THREADS / 64; ++w) s += cand_d[w];
        a.loss_partial[(size_t)p * kMaxGridX + blockIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// Screened search (l2 normalisation, d = 128, K * d * 2 <= 128 KiB): the exact fp32 search above
// spends 2 K flop per 8 bytes of row traffic and is bound by the fp32 matrix rate, not by HBM.
// Here an fp16 MFMA pass scores every codeword with a proven error bound, and only codewords whose
// fp16 score is within that bound of the best one are re-scored in exact fp32 -- on typical data
// 1.1 per row, so the kernel streams rows near memory speed while returning the same index as the
// exact search.
//
// Bound.  After l2 normalisation |zn_i|, |cn_i| <= 1, so fp16 (11 significant bits, round to
// nearest) carries each with relative error <= 2^-11, or absolute error <= 2^-25 below the fp16
// normal range.  Products are exact in fp32; the fp32 accumulation over d = 128 terms adds at most
// 2^-17 relative.  Hence |<zn,cn>_fp16 - <zn,cn>| <= e |zn| |cn| + 1e-6 with e = 2^-10 + 2^-16
// (Cauchy-Schwarz on sum |zn_i| |cn_i|).  Maximising v_k = <zn,cn_k> - csq_k / 2 is minimising the
// distance; the exact maximiser k* has v~_k* >= max v~ - 2 e |zn| max|cn| - 2e-6, so every k with
// v~_k >= max v~ - T is kept, T = 2 e |zn| max|cn| + 2e-6 plus a margin of 1e-5 (|zn|^2 + max|cn|^2) / 2
// for the fp32 rounding of the exact evaluation.  Whatever is dropped loses by more than that margin
// in exact arithmetic.
//
// Each wave works alone on 16-row tiles (no workgroup barrier in the loop): rows HBM -> registers
// (the next tile is requested after the MFMA passes, so that it occupies registers only while the
// latency-bound tail of the tile runs) -> normalise -> fp16 B fragments through a private 4 KiB LDS
// tile; the fp16 codebook of the sub-space sits in LDS (A operand, conflict-free ds_read_b128); pass 1
// finds the best score per row, pass 2 recomputes the scores into a per-lane bit mask of survivors
// (branch free, two instructions per score); a single survivor is the answer, several are listed per
// row and scored in exact fp32 by a half-wave each, straight from the row registers; the chosen raw
// codewords of the whole tile are gathered in one L2 round trip.
// The kernel is bound by vector-instruction issue and latency, not by the matrix pipe (17 % busy) or
// LDS: per 16-row tile a wave issues ~990 vector instructions (it was 1470 with IEEE division / square
// root sequences and compare-select masks), 128 MFMAs and 128 fragment reads.
// ---------------------------------------------------------------------------------------------
// Sum over the 32 lanes that hold one row: four DPP adds (quad swaps, half-row and row mirrors)
// cover 16 lanes without touching the LDS crossbar, one shuffle joins the two 16-lane rows.
__device__ __forceinline__ float row32_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false));
    return x + __shfl_xor(x, 16, 64);
}

// Row normalisation of the screened kernel, x / max(|x|, 1e-12), without the compiler's IEEE sequences
// (17 instructions per square root, 11 per division: they were a third of this kernel's vector
// instructions, and the kernel is bound by how many of those a SIMD can issue).
//  * sqrt_rn: v_sqrt_f32 (1 ulp) moved to the correctly rounded result by the usual test of its two
//    neighbours; no pre-scaling of tiny arguments, because below 1e-24 the 1e-12 floor replaces the root.
//  * the quotients share one reciprocal (v_rcp_f32 + a Newton step, < 1 ulp) and are one multiply each,
//    faithful to ~1 ulp.  The norm itself is already summed in this kernel's own order (a lane tree), so a
//    normalised element equals the reference's only to an ulp or two either way; the distances that
//    decide an index are exact fp32 functions of these normalised values, as before.
__device__ __forceinline__ float sqrt_rn(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float lo = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) - 1u);
    const float hi = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) + 1u);
    const float r_lo = fmaf(-lo, s, x), r_hi = fmaf(-hi, s, x);
    float out = r_lo <= 0.0f ? lo : s;
    out = r_hi > 0.0f ? hi : out;
    return (x == 0.0f || x == __builtin_inff()) ? x : out;
}
__device__ __forceinline__ float shared_rcp(float den) {
    const float r = __builtin_amdgcn_rcpf(den);
    return fmaf(fmaf(-den, r, 1.0f), r, r);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// two fp32 -> one dword of two fp16, round to nearest even
__device__ __forceinline__ uint32_t pack_f16(float lo, float hi) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const f16x2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

#ifdef EQUSS_PQ_STAMPS   // developer build (tools/ubench/pq_stamps.hip): 100 MHz clock at the phase boundaries of one wave
__device__ unsigned long long g_pq_stamps[64 * 8];
#define PQ_STAMP(ph) do { if (stamp_it < 64 && wave_g == 37 && lane == 0 && blockIdx.y == 3) g_pq_stamps[stamp_it * 8 + (ph)] = wall_clock64(); } while (0)
#else
#define PQ_STAMP(ph) do { } while (0)
#endif

// mask = 2 mask + (d < 0): the sign bit of d = score - threshold shifted into the mask by one
// v_alignbit_b32 ((mask:d) >> 31), with no compare, no VCC and no select.  A survivor (d >= 0, including
// +0 from score == threshold) therefore shows as a ZERO bit; the caller inverts the finished word.
__device__ __forceinline__ uint32_t shift_in_sign(uint32_t mask, float d) {
    return __builtin_amdgcn_alignbit(mask, __builtin_bit_cast(uint32_t, d), 31);
}

constexpr int kScrCap = 8;      // survivors listed per row; more -> exact scan of all K for that row
constexpr float kScrE = 0.0009765625f + 0.0000152587890625f;  // 2^-10 + 2^-16
constexpr int kScrRows = 16;        // rows per wave tile (v_mfma_f32_16x16x32_f16)
constexpr int kScrWaveLds = kScrRows * 256 + kScrRows * 4 * 2 + kScrRows * kScrCap * 4;

// NWAVES independent waves per workgroup: 12 (three per SIMD, <= 168 VGPRs) when the fp16 codebook
// takes <= 64 KiB of LDS, 6 when it takes up to 128 KiB (K = 512).  MW = 32-bit survivor-mask words
// per lane = K / 128.
// KEEP: the scores of pass 1 stay in registers (K / 4 per lane) and pass 2 only compares them: half the MFMAs and
// fragment reads, for eight waves of 256 registers instead of twelve of 168 (K = 256), or four of 512 instead of six of
// 256 (K = 512); K = 128 MW exactly.
template <int D, int NWAVES, int MW, bool KEEP = false>
__global__ __launch_bounds__(NWAVES * 64, NWAVES == 12 ? 3 : NWAVES == 4 ? 1 : 2) void pq_screen_kernel(const PqArgs a) {
    constexpr int kScrThreads = NWAVES * 64;
    static_assert(D == 128, "fragment and swizzle arithmetic below is written for d = 128");
    constexpr int LPR = D / 4;            // lanes holding one row as float4 each
    constexpr int RPI = 64 / LPR;         // rows per wave-wide load
    constexpr int NI = kScrRows / RPI;    // float4 per lane per tile
    constexpr int KB = D / 32;            // MFMA k blocks
    constexpr int GB = NI;                // codeword gathers in flight per lane
    constexpr int ROWB = D * 2;           // bytes per fp16 row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, p = blockIdx.y;
    const int j = lane & 15, q = lane >> 4;

    char* cb_s = smem;                                              // K x D fp16, chunk-swizzled
    float* csq_s = reinterpret_cast<float*>(cb_s + (size_t)K * ROWB);  // K
    float* red_s = csq_s + K;                                       // 16
    char* wbase = reinterpret_cast<char*>(red_s + 16) + wave * kScrWaveLds;
    char* zb = wbase;                                               // 16 x D fp16, chunk-swizzled
    int* cnt_s = reinterpret_cast<int*>(zb + kScrRows * ROWB);      // 16
    int* fin_s = cnt_s + kScrRows;                                  // 16
    int* cand_s = fin_s + kScrRows;                                 // 16 x kScrCap

    // ---- codebook -> LDS as fp16; max |cn|^2 ----------------------------------------------------
    float cmax2 = 0.f;
    for (int item = tid; item < K * LPR; item += kScrThreads) {
        const int row = item / LPR, c = item % LPR;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.cn + ((size_t)p * K + row) * D + 4 * c);
        uint2 pk;
        pk.x = pack_f16(v.x, v.y);
        pk.y = pack_f16(v.z, v.w);
        *reinterpret_cast<uint2*>(cb_s + row * ROWB + (((c >> 1) ^ (row & 15)) * 16) + (c & 1) * 8) = pk;
    }
    for (int k = tid; k < K; k += kScrThreads) {
        const float sq = a.csq[(size_t)p * K + k];
        csq_s[k] = -0.5f * sq;      // the accumulators' start value; |cn|^2 itself is -2 x this, exactly
        cmax2 = fmaxf(cmax2, sq);
    }
    cmax2 = equss::group_max<64>(cmax2);
    if (lane == 0) red_s[wave] = cmax2;
    __syncthreads();
    cmax2 = red_s[0];
#pragma unroll
    for (int w = 1; w < NWAVES; ++w) cmax2 = fmaxf(cmax2, red_s[w]);
    __syncthreads();   // red_s is reused for the loss at the end

    const int wave_g = blockIdx.x * NWAVES + wave;
    const int nwaves = gridDim.x * NWAVES;
    const int rsub = lane / LPR, c4 = lane % LPR;   // this lane's row within a load and its float4 column
    float loss = 0.f;

    // Loads are unconditional: rows past the end re-read the last row (their outputs are never stored).
    f32x4 pre[NI];
    const float* code_p = a.code + (size_t)p * D + 4 * c4;
    auto fetch = [&](int tile) {
        const int t = min(tile, a.ntiles - 1);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int grow = min(t * kScrRows + i * RPI + rsub, a.rows - 1);
            // (streamed once: the non-temporal hints on these loads and on the z_q stores are worth 6 % of the launch)
            pre[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(code_p + (size_t)grow * a.ld_code));
        }
    };
    fetch(wave_g);

#ifdef EQUSS_PQ_STAMPS
    int stamp_it = -1;
#endif
    for (int tile = wave_g; tile < a.ntiles; tile += nwaves) {
#ifdef EQUSS_PQ_STAMPS
        ++stamp_it;
#endif
        PQ_STAMP(0);
        // ---- normalise; fp16 copy -> private LDS tile ----------------------------------------------
        f32x4 cur[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            // F.normalize(dim=1, eps=1e-12): x / max(||x||, eps)
            f32x4 v = pre[i];
            const float den = fmaxf(sqrt_rn(row32_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w)), 1e-12f);
            const float rden = shared_rcp(den);
            v.x *= rden; v.y *= rden; v.z *= rden; v.w *= rden;
            cur[i] = v;
            const int r = i * RPI + rsub;
            uint2 pk;
            pk.x = pack_f16(v.x, v.y);
            pk.y = pack_f16(v.z, v.w);
            *reinterpret_cast<uint2*>(zb + r * ROWB + (((c4 >> 1) ^ (r & 15)) * 16) + (c4 & 1) * 8) = pk;
        }
        if (lane < kScrRows) cnt_s[lane] = 0;
        if constexpr (KEEP) fetch(tile + nwaves);   // (room for the next tile's rows from the start: the whole tile to land)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS writes are visible to its reads
        // B fragments: lane (j, q) holds zn[j][32 t + 8 q .. + 7]
        f16x8 bf[KB];
#pragma unroll
        for (int t = 0; t < KB; ++t)
            bf[t] = *reinterpret_cast<const f16x8*>(zb + j * ROWB + (((4 * t + q) ^ j) * 16));

        // scores of codeword tile ct: lane (j, q) gets v~ = <zn_j, cn_k> - csq_k / 2 for k = 16 ct + 4 q + e in acc[e]
        auto scores = [&](int ct) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(csq_s + ct * 16 + 4 * q);
            const char* arow = cb_s + (ct * 16 + j) * ROWB;
#pragma unroll
            for (int t = 0; t < KB; ++t) {
                const f16x8 af = *reinterpret_cast<const f16x8*>(arow + (((4 * t + q) ^ j) * 16));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[t], acc, 0, 0, 0);
            }
            return acc;
        };

        PQ_STAMP(1);
        // ---- pass 1: best and second-best score per row ------------------------------------------------
        float best = -__builtin_inff();
        constexpr int NCT = 8 * MW;
        f32x4 sc[KEEP ? NCT : 1];
        if constexpr (KEEP) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                sc[ct] = scores(ct);
                best = fmaxf(fmaxf(fmaxf(fmaxf(best, sc[ct].x), sc[ct].y), sc[ct].z), sc[ct].w);
            }
        } else {
            for (int ct = 0; ct < K / 16; ct += 2) {
                const f32x4 a0 = scores(ct), a1 = scores(ct + 1);
                best = fmaxf(fmaxf(fmaxf(fmaxf(best, a0.x), a0.y), a0.z), a0.w);   // (chains of three fold into v_max3_f32)
                best = fmaxf(fmaxf(fmaxf(fmaxf(best, a1.x), a1.y), a1.z), a1.w);
            }
        }
        best = fmaxf(best, __shfl_xor(best, 16, 64));   // the four lanes that share a row
        best = fmaxf(best, __shfl_xor(best, 32, 64));
        // |zn| <= 1 (+ rounding) after l2 normalisation, whatever the row held
        constexpr float zs = 1.0001f;
        const float T = 2.0f * kScrE * sqrtf(zs * cmax2) + 0.5e-5f * (zs + cmax2) + 2e-6f;
        const float thr = best - T;
        // a row that holds a NaN or an Inf normalises to NaN scores, which fmaxf skips: best stays -inf and
        // the row ends with no survivor, i.e. index 0 (what arg-min over all-NaN distances returns)
        const bool row_finite = best > -__builtin_inff();

        PQ_STAMP(2);
        // ---- pass 2: recompute the scores; one bit per (codeword of this lane) that survives --------------
        // Branch-free: two instructions per score (subtract, shift the sign in), one 32-bit word per 8
        // codeword tiles (32 codewords of this lane).  Value n of a word (n = 4 (ct - ct0) + e) ends at bit
        // nv - 1 - n.
        const int nct = K / 16;
        uint32_t mk[MW];
#pragma unroll
        for (int w = 0; w < MW; ++w) mk[w] = ~0u;   // survivors are shifted in as zero bits; inverted below
#pragma unroll
        for (int w = 0; w < MW; ++w) {
            if constexpr (KEEP) {
#pragma unroll
                for (int ct = 8 * w; ct < 8 * w + 8; ++ct)
#pragma unroll
                    for (int e = 0; e < 4; ++e) mk[w] = shift_in_sign(mk[w], sc[ct][e] - thr);
                mk[w] = row_finite ? ~mk[w] : 0u;
                continue;
            }
            const int ct_end = min(nct, 8 * w + 8);
            for (int ct = 8 * w; ct < ct_end; ct += 2) {
                const f32x4 a0 = scores(ct), a1 = scores(ct + 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) mk[w] = shift_in_sign(mk[w], a0[e] - thr);
#pragma unroll
                for (int e = 0; e < 4; ++e) mk[w] = shift_in_sign(mk[w], a1[e] - thr);
            }
            mk[w] = row_finite ? ~mk[w] : 0u;
        }
        // The next tile's rows are requested only now: they then occupy registers while the survivor
        // lists, the occasional exact re-scoring and the gather run (several microseconds, enough for the
        // loads to land), not during the two MFMA passes, whose fragments need the room.
        if constexpr (!KEEP) fetch(tile + nwaves);
        PQ_STAMP(3);
        auto codeword = [&](int w, int bit) {   // bit of word w -> codeword index
            const int n = 4 * min(nct - 8 * w, 8) - 1 - bit;
            return 16 * (8 * w + (n >> 2)) + 4 * q + (n & 3);
        };
        int mine = 0;
#pragma unroll
        for (int w = 0; w < MW; ++w) mine += __popc(mk[w]);
        int total = mine + __shfl_xor(mine, 16, 64);
        total += __shfl_xor(total, 32, 64);
        int hit = -1;
        if (mine == 1) {
#pragma unroll
            for (int w = 0; w < MW; ++w)
                if (mk[w]) hit = codeword(w, 31 - __clz(mk[w]));
        }
        hit = max(hit, __shfl_xor(hit, 16, 64));
        hit = max(hit, __shfl_xor(hit, 32, 64));
        if (lane < kScrRows) fin_s[lane] = max(hit, 0);   // right when exactly one codeword survived
        if (total >= 2 && mine > 0) {                      // several survivors: list them for the exact pass
#pragma unroll
            for (int w = 0; w < MW; ++w) {
                uint32_t m = mk[w];
                while (m) {
                    const int bit = 31 - __clz(m);
                    m &= ~(1u << bit);
                    const int slot = atomicAdd(&cnt_s[j], 1);
                    if (slot < kScrCap) cand_s[j * kScrCap + slot] = codeword(w, bit);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        PQ_STAMP(4);
        // ---- exact fp32 re-scoring of rows with several survivors: a half-wave per row ---------------
        // dist = (|zn|^2 + |cn|^2) - 2 <zn, cn>, smaller distance wins, then smaller index.
        if (__builtin_amdgcn_ballot_w64(total >= 2) != 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int r = i * RPI + rsub;
                const int n = tile * kScrRows + r < a.rows ? cnt_s[r] : 0;   // padding rows of the last tile are skipped
                if (n >= 2) {
                    const f32x4 z = cur[i];
                    const float zq = row32_sum(z.x * z.x + z.y * z.y + z.z * z.z + z.w * z.w);
                    const int count = n > kScrCap ? K : n;   // list overflowed: exact scan of the whole codebook
                    float bd = __builtin_inff();
                    int bk = 0;
                    for (int ci = 0; ci < count; ci += 2) {   // two candidates per step: their loads overlap
                        const int cj = ci + 1 < count ? ci + 1 : ci;
                        const int k0 = n > kScrCap ? ci : cand_s[r * kScrCap + ci];
                        const int k1 = n > kScrCap ? cj : cand_s[r * kScrCap + cj];
                        const f32x4 c0 = *reinterpret_cast<const f32x4*>(a.cn + ((size_t)p * K + k0) * D + 4 * c4);
                        const f32x4 c1 = *reinterpret_cast<const f32x4*>(a.cn + ((size_t)p * K + k1) * D + 4 * c4);
                        const float dot0 = row32_sum(fmaf(z.x, c0.x, fmaf(z.y, c0.y, fmaf(z.z, c0.z, z.w * c0.w))));
                        const float dot1 = row32_sum(fmaf(z.x, c1.x, fmaf(z.y, c1.y, fmaf(z.z, c1.z, z.w * c1.w))));
                        const float d0 = (zq + -2.0f * csq_s[k0]) - 2.0f * dot0;
                        const float d1 = (zq + -2.0f * csq_s[k1]) - 2.0f * dot1;
                        if (d0 < bd || (d0 == bd && k0 < bk)) { bd = d0; bk = k0; }
                        if (d1 < bd || (d1 == bd && k1 < bk)) { bd = d1; bk = k1; }
                    }
                    if (c4 == 0) fin_s[r] = bk;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < kScrRows && tile * kScrRows + lane < a.rows)
            a.idx[(size_t)p * a.rows + tile * kScrRows + lane] = (int64_t)fin_s[lane];

        PQ_STAMP(5);
        // ---- gather + straight-through output + mse numerator, from the row registers -----------------
        // All codeword loads of the tile are issued before the first is used: one L2 round trip per tile
        // instead of one per row pair.  (A 16-wave, 128-register build with half-tile batches was tried:
        // it spills and runs 15 % slower than 12 waves at 168 registers.)
#pragma unroll
        for (int h = 0; h < NI; h += GB) {
            f32x4 emb[GB];
#pragma unroll
            for (int i = 0; i < GB; ++i) {
                const int k = fin_s[(h + i) * RPI + rsub];
                emb[i] = *reinterpret_cast<const f32x4*>(a.gather + ((size_t)p * K + k) * D + 4 * c4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GB; ++i) {
                const int r = (h + i) * RPI + rsub;
                const int grow = tile * kScrRows + r;
                if (grow < a.rows) {
                    const f32x4 zn = cur[h + i];
                    const f32x4 e = emb[i];
                    f32x4 df; df.x = e.x - zn.x; df.y = e.y - zn.y; df.z = e.z - zn.z; df.w = e.w - zn.w;
                    loss += df.x * df.x + df.y * df.y + df.z * df.z + df.w * df.w;
                    if (a.zq) {
                        f32x4 o; o.x = zn.x + df.x; o.y = zn.y + df.y; o.z = zn.z + df.z; o.w = zn.w + df.w;
                        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(a.zq + (size_t)grow * a.ld_zq + (size_t)p * D + 4 * c4));
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" ::: "memory");
        PQ_STAMP(6);
    }
    // ---- per-workgroup loss partial (fixed order, no atomics) ----------------------------------
    loss = group_sum<64>(loss);
    if (lane == 0) red_s[wave] = loss;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < NWAVES; ++w) s += red_s[w];
        a.loss_partial[(size_t)p * kMaxGridX + blockIdx.x] = s;
    }
}

__global__ void pq_loss_reduce(const float* __restrict__ partial, float* __restrict__ loss_sum, int gx) {
    const int p = blockIdx.x;
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < gx; ++i) s += partial[(size_t)p * kMaxGridX + i];
        loss_sum[p] = s;
    }
}

// softmax(-dist * inv_tau) over K for every (row, sub-space).  One workgroup per tile of 16 rows and sub-space: the
// normalised rows wait in LDS, a thread owns codewords (tid, tid + 256, ...) and reads each of them once for all 16 rows
// (a workgroup per row read the whole codebook of the sub-space for that one row), the distances of the tile go to LDS and
// a wave per row turns them into probabilities.  Per (row, codeword) the dot product is the same fma chain as before.
// Not on the timed path (the eval loop never consumes it), so plain fp32 VALU code.
constexpr int kProbRows = 16;
template <int D>
__global__ __launch_bounds__(256) void pq_prob_kernel(const float* __restrict__ code, int64_t ld_code,
                                                      const float* __restrict__ cn,
                                                      const float* __restrict__ csq, float* __restrict__ prob,
                                                      int rows, int num_pq, int K, int mode, float inv_tau, int rt) {
    // (rt <= kProbRows rows per tile: fewer when K is large, so that the tile's distances fit 64 KiB of LDS)
    constexpr int LPR = D / 4, RT = kProbRows;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* zn = reinterpret_cast<float*>(smem);  // RT x D
    float* arow = zn + RT * D;                   // RT: sum of squares of the normalised rows
    float* dist = arow + RT;                     // rt x K
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * rt, p = blockIdx.y;
    for (int r = wave; r < RT; r += 4) {   // one wave normalises a row (LPR <= 32 lanes carry data)
        const int row = row0 + r < rows ? row0 + r : rows - 1;   // rows past the end repeat the last row; nothing of them is stored
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (lane < LPR) v = *reinterpret_cast<const f32x4*>(code + (size_t)row * ld_code + (size_t)p * D + 4 * lane);
        const float sa = normalize_row<D>(v, mode);
        if (lane < LPR) *reinterpret_cast<f32x4*>(zn + r * D + 4 * lane) = v;
        if (lane == 0) arow[r] = sa;
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
        const float* c = cn + ((size_t)p * K + k) * D;
        float dot[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) dot[r] = 0.f;
        for (int j = 0; j < D; j += 4) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(c + j);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const f32x4 z4 = *reinterpret_cast<const f32x4*>(zn + r * D + j);   // (the same address in every lane: a broadcast)
                dot[r] = fmaf(z4.x, c4.x, dot[r]);
                dot[r] = fmaf(z4.y, c4.y, dot[r]);
                dot[r] = fmaf(z4.z, c4.z, dot[r]);
                dot[r] = fmaf(z4.w, c4.w, dot[r]);
            }
        }
        const float ck = csq[(size_t)p * K + k];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const float t0 = arow[r] + ck;
            if (r < rt) dist[r * K + k] = -(t0 - 2.0f * dot[r]) * inv_tau;
        }
    }
    __syncthreads();
    for (int r = wave; r < rt; r += 4) {
        if (row0 + r >= rows) break;
        float* dr = dist + r * K;
        float m = -__builtin_inff();
        for (int k = lane; k < K; k += 64) m = fmaxf(m, dr[k]);
        m = equss::group_max<64>(m);
        float lsum = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float e = expf(dr[k] - m);
            dr[k] = e;
            lsum += e;
        }
        const float s = group_sum<64>(lsum);
        float* out = prob + (size_t)(row0 + r) * num_pq * K + (size_t)p * K;
        for (int k = lane; k < K; k += 64) out[k] = dr[k] / s;
    }
}

struct PqPlan { int NT, WK, WR, RT, R; };

// Split of the 8 waves of a workgroup: WK waves share K (each pins NT tiles of 32 codewords),
// WR groups share the rows of a tile, every group walks RT row tiles of 32.
bool make_plan(int K, int d, PqPlan* pl) {
    if (d != 16 && d != 32 && d != 64 && d != 128) return false;
    if (K <= 0 || K % 32 != 0) return false;
    const int tiles = K / 32;
    int nt = 128 / d;  // pinned codebook <= 64 registers per lane
    if (nt > 8) nt = 8;
    while (nt > 1 && (tiles % nt != 0)) nt >>= 1;
    int wk = tiles / nt;
    if (wk > 8 && (wk % 2 == 0) && nt * 2 * (d / 2) <= 128 && nt * 2 <= 8) { nt *= 2; wk /= 2; }  // <= 128 registers
    if (wk > 8 || (wk & (wk - 1)) != 0) return false;
    const int wr = 8 / wk;
    int rt = wr == 1 ? 2 : 1;
    int r = 32 * wr * rt;
    while (r * (d / 4) < kPqThreads) { rt *= 2; r *= 2; }  // at least one float4 per thread per tile
    if ((r * (d / 4)) / kPqThreads > kPqMaxItems) return false;
    pl->NT = nt; pl->WK = wk; pl->WR = wr; pl->RT = rt; pl->R = r;
    return true;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int g_pq_mode = 0;  // 0: screened search where it applies; 1: exact fp32 search everywhere; 2: as 0 without the kept-score variants

struct PqWorkspace { float* cn; float* csq; float* loss_partial; size_t bytes; };

PqWorkspace carve(void* ws, int num_pq, int K, int d) {
    PqWorkspace w;
    size_t off = 0;
    char* base = reinterpret_cast<char*>(ws);
    w.cn = reinterpret_cast<float*>(base + off); off += align256((size_t)num_pq * K * d * 4);
    w.csq = reinterpret_cast<float*>(base + off); off += align256((size_t)num_pq * K * 4);
    w.loss_partial = reinterpret_cast<float*>(base + off); off += align256((size_t)num_pq * kMaxGridX * 4);
    w.bytes = off;
    return w;
}

template <int D>
int launch_prep(const float* cb, const PqWorkspace& w, int total_rows, int mode, hipStream_t s) {
    const int threads = 256;
    const long items = (long)total_rows * (D / 4);
    const int blocks = (int)((items + threads - 1) / threads);
    hipLaunchKernelGGL(pq_prep_codebook<D>, dim3(blocks), dim3(threads), 0, s, cb, w.cn, w.csq, total_rows, mode);
    return equss::check_launch("pq_prep_codebook");
}

template <int D, int NT, int ITEMS>
int launch_assign_items(const PqArgs& a, const PqPlan& pl, int gx, hipStream_t s) {
    const size_t lds = ((size_t)pl.R * D + pl.R + a.K + 2 * (size_t)pl.WK * pl.R + pl.R) * 4;
    auto kern = pq_assign_kernel<D, NT, ITEMS>;
    if (lds > 65536 && !equss::lds_opt_in(reinterpret_cast<const void*>(kern), 163840))
        return equss::fail(EQUSS_E_UNSUPPORTED, "pq_assign: cannot reserve %zu bytes of LDS", lds);
    hipLaunchKernelGGL(kern, dim3(gx, a.num_pq), dim3(kPqThreads), lds, s, a);
    return equss::check_launch("pq_assign_kernel");
}

template <int D, int NT>
int launch_assign(const PqArgs& a, const PqPlan& pl, int gx, hipStream_t s) {
    switch ((pl.R * (D / 4)) / kPqThreads) {
        case 1: return launch_assign_items<D, NT, 1>(a, pl, gx, s);
        case 2: return launch_assign_items<D, NT, 2>(a, pl, gx, s);
        case 4: return launch_assign_items<D, NT, 4>(a, pl, gx, s);
        case 8: return launch_assign_items<D, NT, 8>(a, pl, gx, s);
        default: break;
    }
    return equss::fail(EQUSS_E_UNSUPPORTED, "pq_assign: no kernel for a %d-row tile at d=%d", pl.R, D);
}

template <int D>
int dispatch_nt(const PqArgs& a, const PqPlan& pl, int gx, hipStream_t s) {
    switch (pl.NT) {
        case 1: return launch_assign<D, 1>(a, pl, gx, s);
        case 2: return launch_assign<D, 2>(a, pl, gx, s);
        case 4: if constexpr (D <= 64) return launch_assign<D, 4>(a, pl, gx, s); break;
        case 8: if constexpr (D <= 32) return launch_assign<D, 8>(a, pl, gx, s); break;
        default: break;
    }
    return equss::fail(EQUSS_E_UNSUPPORTED, "pq_assign: no kernel for d=%d NT=%d", D, pl.NT);
}

int check_common(const float* code, int64_t ld_code, const float* codebooks, void* workspace,
                 size_t workspace_bytes, int rows, int num_pq, int K, int d, int normalize, PqPlan* pl) {
    if (!code || !codebooks || !workspace) return equss::fail(EQUSS_E_NULL, "pq: null pointer argument");
    if (rows <= 0 || num_pq <= 0 || K <= 0 || d <= 0)
        return equss::fail(EQUSS_E_SHAPE, "pq: rows=%d num_pq=%d K=%d d=%d must be positive", rows, num_pq, K, d);
    if (normalize < EQUSS_NORM_NONE || normalize > EQUSS_NORM_ZSCORE)
        return equss::fail(EQUSS_E_UNSUPPORTED, "pq: unsupported normalize mode %d", normalize);
    if (!make_plan(K, d, pl))
        return equss::fail(EQUSS_E_UNSUPPORTED,
                           "pq: unsupported shape K=%d d=%d (d in {16,32,64,128}; K/32 a power-of-two multiple "
                           "of the per-wave tile count, at most 16 waves)", K, d);
    if (ld_code < (int64_t)num_pq * d || (ld_code % 4) != 0 || ((uintptr_t)code % 16) != 0 ||
        ((uintptr_t)codebooks % 16) != 0 || ((uintptr_t)workspace % 256) != 0)
        return equss::fail(EQUSS_E_ALIGN, "pq: code/codebooks must be 16-byte aligned, workspace 256-byte aligned, "
                                          "ld_code a multiple of 4 and >= num_pq*d");
    if (workspace_bytes < equss_pq_workspace_bytes(rows, num_pq, K, d))
        return equss::fail(EQUSS_E_WORKSPACE, "pq: workspace of %zu bytes is smaller than the %zu required",
                           workspace_bytes, equss_pq_workspace_bytes(rows, num_pq, K, d));
    return EQUSS_OK;
}

}  // namespace

extern "C" void equss_dev_pq_mode(int m) { g_pq_mode = m; }  // test knob, not part of the ABI

extern "C" size_t equss_pq_workspace_bytes(int rows, int num_pq, int K, int d) {
    (void)rows;
    if (num_pq <= 0 || K <= 0 || d <= 0) return 0;
    return carve(nullptr, num_pq, K, d).bytes;
}

namespace {

// Scratch of equss_pq_assign_static: normalised codebooks, their squared norms and the per-workgroup
// loss partials of one call, i.e. what equss_pq_assign takes as its workspace argument.  One copy per
// GPU, held until the process ends.  Host code never uses this name as an address: the device address
// is looked up once per device with hipGetSymbolAddress (static_scratch below).
constexpr size_t kStaticScratchBytes = 5u << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices] = {nullptr};

// Device address of `scratch` on the current device (looked up on the first call per device).
int static_scratch(void** out) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) { equss::fail(0, "pq_assign_static: hipGetDevice: %s", hipGetErrorString(e)); return -(int)e; }
    if (device < 0 || device >= kMaxDevices)
        return equss::fail(EQUSS_E_UNSUPPORTED, "pq_assign_static: device ordinal %d is outside [0, %d)", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            equss::fail(0, "pq_assign_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
            return e != hipSuccess ? -(int)e : -(int)hipErrorInvalidSymbol;
        }
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return EQUSS_OK;
}

int pq_assign_impl(const float* code, int64_t ld_code, const float* codebooks, int64_t* idx,
                   float* zq, int64_t ld_zq, float* loss_sum, void* workspace,
                   size_t workspace_bytes, int rows, int num_pq, int K, int d, int normalize,
                   int gather_normalized, void* stream) {
    PqPlan pl;
    int rc = check_common(code, ld_code, codebooks, workspace, workspace_bytes, rows, num_pq, K, d, normalize, &pl);
    if (rc) return rc;
    if (!idx) return equss::fail(EQUSS_E_NULL, "pq_assign: idx is null");
    if (zq && (ld_zq < (int64_t)num_pq * d || (ld_zq % 4) != 0 || ((uintptr_t)zq % 16) != 0))
        return equss::fail(EQUSS_E_ALIGN, "pq_assign: zq must be 16-byte aligned with ld_zq a multiple of 4");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PqWorkspace w = carve(workspace, num_pq, K, d);
    {
        equss::ProfScope ps(EQUSS_K_PQ_PREP, s);
        switch (d) {
            case 16: rc = launch_prep<16>(codebooks, w, num_pq * K, normalize, s); break;
            case 32: rc = launch_prep<32>(codebooks, w, num_pq * K, normalize, s); break;
            case 64: rc = launch_prep<64>(codebooks, w, num_pq * K, normalize, s); break;
            default: rc = launch_prep<128>(codebooks, w, num_pq * K, normalize, s); break;
        }
    }
    if (rc) return rc;
    PqArgs a;
    a.code = code; a.ld_code = ld_code; a.cn = w.cn; a.csq = w.csq;
    a.gather = gather_normalized ? w.cn : codebooks;
    a.idx = idx; a.zq = zq; a.ld_zq = ld_zq; a.loss_partial = w.loss_partial;
    a.rows = rows; a.num_pq = num_pq; a.K = K; a.mode = normalize;
    a.WK = pl.WK; a.WR = pl.WR; a.RT = pl.RT;
    a.ntiles = (rows + pl.R - 1) / pl.R;
    int gx = (256 + num_pq - 1) / num_pq;  // one resident workgroup per CU
    const bool screened = g_pq_mode != 1 && d == 128 && normalize == EQUSS_NORM_L2 && (size_t)K * d * 2 <= 131072;
    if (screened) {
        a.ntiles = (rows + kScrRows - 1) / kScrRows;
        const bool small = (size_t)K * d * 2 <= 65536;
        const bool keep = g_pq_mode != 2 && K == 256;   // pass-1 scores kept in registers (mode 2: the twelve-wave kernel)
        const bool keep4 = g_pq_mode != 2 && K == 512;   // K = 512: kept scores need one wave per SIMD (360 -> 320 us at 16 x 512)
        const int waves = keep4 ? 4 : keep ? 8 : small ? 12 : 6;
        if (gx > (a.ntiles + waves - 1) / waves) gx = (a.ntiles + waves - 1) / waves;
        if (gx > kMaxGridX) gx = kMaxGridX;
        const size_t lds = (size_t)K * d * 2 + (K + 16) * 4 + (size_t)waves * kScrWaveLds;
        void (*kern)(const PqArgs) = keep4 ? pq_screen_kernel<128, 4, 4, true> : keep ? pq_screen_kernel<128, 8, 2, true> : !small ? pq_screen_kernel<128, 6, 4> : (K <= 128 ? pq_screen_kernel<128, 12, 1> : pq_screen_kernel<128, 12, 2>);
        if (!equss::lds_opt_in(reinterpret_cast<const void*>(kern), 163840))
            return equss::fail(EQUSS_E_UNSUPPORTED, "pq_assign: cannot reserve LDS for the screened search");
        {
            equss::ProfScope ps(EQUSS_K_PQ_ASSIGN, s);
            hipLaunchKernelGGL(kern, dim3(gx, num_pq), dim3(waves * 64), lds, s, a);
            rc = equss::check_launch("pq_screen_kernel");
        }
        if (rc) return rc;
        if (loss_sum) {
            hipLaunchKernelGGL(pq_loss_reduce, dim3(num_pq), dim3(64), 0, s, w.loss_partial, loss_sum, gx);
            rc = equss::check_launch("pq_loss_reduce");
        }
        return rc;
    }
    if (gx > a.ntiles) gx = a.ntiles;
    if (gx > kMaxGridX) gx = kMaxGridX;
    if (gx < 1) gx = 1;
    {
        equss::ProfScope ps(EQUSS_K_PQ_ASSIGN, s);
        switch (d) {
            case 16: rc = dispatch_nt<16>(a, pl, gx, s); break;
            case 32: rc = dispatch_nt<32>(a, pl, gx, s); break;
            case 64: rc = dispatch_nt<64>(a, pl, gx, s); break;
            default: rc = dispatch_nt<128>(a, pl, gx, s); break;
        }
    }
    if (rc) return rc;
    if (loss_sum) {
        hipLaunchKernelGGL(pq_loss_reduce, dim3(num_pq), dim3(64), 0, s, w.loss_partial, loss_sum, gx);
        rc = equss::check_launch("pq_loss_reduce");
    }
    return rc;
}

}  // namespace

extern "C" int equss_pq_assign(const float* code, int64_t ld_code, const float* codebooks, int64_t* idx,
                               float* zq, int64_t ld_zq, float* loss_sum, void* workspace,
                               size_t workspace_bytes, int rows, int num_pq, int K, int d, int normalize,
                               int gather_normalized, void* stream) {
    return pq_assign_impl(code, ld_code, codebooks, idx, zq, ld_zq, loss_sum, workspace, workspace_bytes, rows, num_pq,
                          K, d, normalize, gather_normalized, stream);
}

extern "C" size_t equss_pq_static_scratch_bytes(void) { return kStaticScratchBytes; }

extern "C" int equss_pq_assign_static(const float* code, int64_t ld_code, const float* codebooks, int64_t* idx,
                                      float* zq, int64_t ld_zq, float* loss_sum, int rows, int num_pq, int K, int d,
                                      int normalize, int gather_normalized, void* stream) {
    if (rows <= 0 || num_pq <= 0 || K <= 0 || d <= 0)
        return equss::fail(EQUSS_E_SHAPE, "pq: rows=%d num_pq=%d K=%d d=%d must be positive", rows, num_pq, K, d);
    const size_t need = equss_pq_workspace_bytes(rows, num_pq, K, d);
    if (need > kStaticScratchBytes)
        return equss::fail(EQUSS_E_WORKSPACE,
                           "pq_assign_static: num_pq=%d K=%d d=%d needs %zu bytes of scratch, the built-in buffer holds "
                           "%zu (use equss_pq_assign with a caller-owned workspace)", num_pq, K, d, need,
                           kStaticScratchBytes);
    void* ws = nullptr;
    const int rc = static_scratch(&ws);
    if (rc) return rc;
    // Every word of the scratch that a call reads is written earlier in the same call (prepared codebooks
    // by pq_prep_codebook, loss partials by every workgroup of the grid that pq_loss_reduce walks), so
    // nothing has to be cleared between calls.
    return pq_assign_impl(code, ld_code, codebooks, idx, zq, ld_zq, loss_sum, ws, kStaticScratchBytes, rows, num_pq, K, d,
                          normalize, gather_normalized, stream);
}

extern "C" int equss_pq_prob(const float* code, int64_t ld_code, const float* codebooks, float* prob,
                             void* workspace, size_t workspace_bytes, int rows, int num_pq, int K, int d,
                             int normalize, float inv_tau, void* stream) {
    PqPlan pl;
    int rc = check_common(code, ld_code, codebooks, workspace, workspace_bytes, rows, num_pq, K, d, normalize, &pl);
    if (rc) return rc;
    if (!prob) return equss::fail(EQUSS_E_NULL, "pq_prob: prob is null");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PqWorkspace w = carve(workspace, num_pq, K, d);
    int rt = (int)((60 * 1024 - ((size_t)kProbRows * d + kProbRows) * 4) / ((size_t)K * 4));
    rt = rt > kProbRows ? kProbRows : (rt < 1 ? 1 : rt);
    const size_t lds = ((size_t)kProbRows * d + kProbRows + (size_t)rt * K) * 4;
    const unsigned prob_tiles = (unsigned)((rows + rt - 1) / rt);
    switch (d) {
        case 16:
            rc = launch_prep<16>(codebooks, w, num_pq * K, normalize, s);
            if (!rc) hipLaunchKernelGGL(pq_prob_kernel<16>, dim3(prob_tiles, num_pq), dim3(256), lds, s, code, ld_code, w.cn, w.csq, prob, rows, num_pq, K, normalize, inv_tau, rt);
            break;
        case 32:
            rc = launch_prep<32>(codebooks, w, num_pq * K, normalize, s);
            if (!rc) hipLaunchKernelGGL(pq_prob_kernel<32>, dim3(prob_tiles, num_pq), dim3(256), lds, s, code, ld_code, w.cn, w.csq, prob, rows, num_pq, K, normalize, inv_tau, rt);
            break;
        case 64:
            rc = launch_prep<64>(codebooks, w, num_pq * K, normalize, s);
            if (!rc) hipLaunchKernelGGL(pq_prob_kernel<64>, dim3(prob_tiles, num_pq), dim3(256), lds, s, code, ld_code, w.cn, w.csq, prob, rows, num_pq, K, normalize, inv_tau, rt);
            break;
        default:
            rc = launch_prep<128>(codebooks, w, num_pq * K, normalize, s);
            if (!rc) hipLaunchKernelGGL(pq_prob_kernel<128>, dim3(prob_tiles, num_pq), dim3(256), lds, s, code, ld_code, w.cn, w.csq, prob, rows, num_pq, K, normalize, inv_tau, rt);
            break;
    }
    if (rc) return rc;
    return equss::check_launch("pq_prob_kernel");
}
