// Multi-head self-attention for head_dim 64 on gfx950, flash style: softmax(q k^T * scale) v with the
// (N x N) matrix never leaving registers.  Replaces Attention.forward's
// q @ k^T -> softmax -> @ v -> transpose/reshape (model/dino/vision_transformer.py:81-88); the
// attention matrix that forward also returns is discarded by its only caller on this path
// (model/dino/dino_featurizer.py:76-80), so it is never written.
//
// One workgroup = NW waves (4 by default) = NW * 32 queries of one (image, head); each wave owns 32
// queries and walks the keys in tiles of 64 (a K/V tile is staged once per workgroup).
//  * K and V tiles go HBM/L2 -> LDS by global_load_lds_dwordx4 (two stages; the next tile is in
//    flight while the current one is on the matrix pipe).  K is chunk-swizzled for conflict-free
//    ds_read_b128; V is swizzled for ds_read_b64_tr_b16, the transposing LDS read that turns
//    row-major V into the k-contiguous fragment the MFMA wants.
//  * S^T = K Q^T with v_mfma_f32_32x32x16_bf16, so a lane owns ONE query and 32 of the tile's 64
//    scores: the row max / row sum are in-lane scans plus one cross-half shuffle.
//  * The S^T accumulators, exponentiated and packed to bf16, are already the B operand of
//    O^T = V^T P^T (rows of S^T are the summed index), so P never touches LDS.
//  * Online softmax in fp32 (exp2 with the scale folded in) without a per-tile maximum: the reference
//    exponent is introduced or raised only when a tile's exponentials leave a safe range (see the kernel).
//  * Workgroup ids are remapped so the query blocks of one (image, head) run on one XCD and share
//    its K/V in that XCD's L2.
#include "equss_common.h"
#include <type_traits>

namespace {

constexpr int HD = 64;          // head dim
constexpr int QW = 32;          // queries per wave
constexpr int kPieces = 8;      // 1-KiB LDS-DMA pieces per K (or V) tile
constexpr int KV = 64;          // keys per tile
constexpr int kTile = KV * HD * 2;      // 8 KiB per K or V tile
constexpr int kStage = 2 * kTile;       // K + V

struct AttnArgs {
    const u16* qkv; u16* out;
    int B, N, H, nqb, ntiles;
    float c;  // scale * log2(e)
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const void __attribute__((address_space(1)))*)gsrc,
        (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x4 ds_read_tr16(const char* p) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}

// NW = waves per workgroup (NW * 32 queries of one (image, head)).
template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void attention_kernel(const AttnArgs a) {
    constexpr int QB = NW * QW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;

    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int bh = logical / a.nqb, qb = logical % a.nqb;
    const int b = bh / a.H, h = bh % a.H;
    const int C = a.H * HD;
    const size_t ld = (size_t)3 * C;
    const u16* base = a.qkv + (size_t)b * a.N * ld + h * HD;  // q of token 0 of this (image, head)

    // ---- Q fragments (B operand): lane (query lr, half lh) holds q[16 s + 8 lh + 0..7] -----------
    int qrow = qb * QB + wave * QW + lr;
    const bool q_valid = qrow < a.N;
    const bool wave_active = qb * QB + wave * QW < a.N;  // waves past the last query only help staging
    if (!q_valid) qrow = a.N - 1;
    // q is pre-multiplied by scale * log2(e), so the MFMA result is already the exponent of exp2.
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 raw = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow * ld + 16 * s + 8 * lh);
        union { bf16x8 v; uint32_t w[4]; } sc;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            sc.w[j] = equss::pack_bf16(equss::bf2f((u16)raw[2 * j]) * a.c, equss::bf2f((u16)raw[2 * j + 1]) * a.c);
        qf[s] = sc.v;
    }
    // A fifth k step carries the softmax reference: K side = 1 in its first slot, Q side = -m_run there
    // (both zero elsewhere), so S^T comes out of the matrix pipe as  q.k * scale * log2(e) - m_run  and
    // goes straight into exp2.  m_run is kept bf16-representable; any reference works as long as the
    // same one is used for every key of a query between two rescalings.
    union { bf16x8 v; uint32_t w[4]; } kx, qx;
    kx.w[0] = lh == 0 ? 0x00003F80u : 0u; kx.w[1] = 0u; kx.w[2] = 0u; kx.w[3] = 0u;
    qx.w[0] = 0u; qx.w[1] = 0u; qx.w[2] = 0u; qx.w[3] = 0u;

    // ---- K/V staging plan: a tile is 8 pieces of 8 rows per operand; piece p of K and of V is moved
    // by wave p % NW (with 8 waves: one piece each).  The per-lane source pointers of tile 0 are kept
    // and advanced by a scalar stride; only the ragged last tile needs its row index clamped (those
    // keys are masked in the softmax; clamping keeps the V values finite).
    constexpr int NP = kPieces / NW;   // pieces of each operand per wave
    static_assert(kPieces % NW == 0 && NW % 2 == 0, "piece i of a wave is piece 0 shifted by NW * 8 rows: same chunk swizzle");
    auto k_chunk = [&](int st_row) { return ((lane & 7) ^ ((st_row >> 1) & 7)) * 8; };
    auto v_chunk = [&](int st_row) { return ((lane & 7) ^ (((st_row >> 1) & 1) << 2)) * 8; };
    const int st_row0 = wave * 8 + (lane >> 3);
    const u16* ksrc = base + (size_t)st_row0 * ld + C + k_chunk(st_row0);
    const u16* vsrc = base + (size_t)st_row0 * ld + 2 * C + v_chunk(st_row0);
    auto stage = [&](int buf, int t) {
        char* sb = smem + buf * kStage;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = i * NW + wave;
            const size_t step = (size_t)(t * KV + i * NW * 8) * ld;   // scalar
            const u16* ks = ksrc + step;
            const u16* vs = vsrc + step;
            if ((t + 1) * KV > a.N) {
                const int st_row = p * 8 + (lane >> 3);
                int key = t * KV + st_row;
                if (key > a.N - 1) key = a.N - 1;
                ks = base + (size_t)key * ld + C + k_chunk(st_row);
                vs = base + (size_t)key * ld + 2 * C + v_chunk(st_row);
            }
#ifndef EQUSS_ABL_ATTNOSTAGE
            glds16(ks, sb + p * 1024);
            glds16(vs, sb + kTile + p * 1024);
#else
            if (t < 2) { glds16(ks, sb + p * 1024); glds16(vs, sb + kTile + p * 1024); }
#endif
        }
    };

    // ---- per-lane LDS read offsets -------------------------------------------------------------
    // K (A operand of S^T): row lr (+32 for the second key tile), chunk 2 s + lh, swizzle (row>>1)&7.
    int k_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) k_off[s] = lr * 128 + (((2 * s + lh) ^ ((lr >> 1) & 7)) * 16);
    // V through the transposing read: 16-lane group G = lane>>4 covers d columns 16 (G&1) .. +15 and
    // the keys of half lh = G>>1; lane 4 q + p of the group addresses key row q, columns 4 p .. 4 p + 3.
    const int vq = (lane >> 2) & 3, vp = lane & 3, vg = (lane >> 4) & 1;
    const int vsw = ((vq >> 1) & 1) << 2;
    int v_off[2];  // per 32-wide d tile
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
        v_off[dt] = (4 * lh + vq) * 128 + (((4 * dt + 2 * vg + (vp >> 1)) ^ vsw) * 16) + (vp & 1) * 8;

    const int xhalf = (lane ^ 32) << 2;   // ds_bpermute address of the lane holding the other 16 keys of the same query
    f32x16 o0 = {0.f}, o1 = {0.f};  // O^T: d tile 0 / 1, lane = query
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = 0.f;   // reference exponent (log2 domain), set from the first tile
    float l_run = 0.f;

    // One key tile: S^T, online softmax, O^T update.
    // `buf` (which staging buffer holds the tile) is a compile-time constant so that every LDS address
    // below is a per-lane base plus an immediate offset.
    //
    // Softmax without a per-tile maximum.  VALU instructions do not hide behind MFMAs of other waves on
    // this part (their times add), so the common path is kept to the arithmetic that cannot be avoided:
    // exp2 of the score, the running sum, the bf16 conversion.  Any per-query reference works as long
    // as every key of the query uses the same one between two rescalings, and fp32 / bf16 carry 2^+-100
    // as well as values near 1, so the reference starts at 0 and is only introduced or raised when a
    // tile's sum of exponentials leaves [2^-60, 2^60] (or is not finite).  Only then is the tile redone
    // the classical way: maximum, new reference, rescaling of O and l.  Once a reference exists it rides
    // on the fifth k step described above.
    bool have_ref = false;   // wave-uniform
    constexpr float kBig = 1.152921504606846976e18f, kTiny = 8.673617379884035e-19f;   // 2^60, 2^-60
    auto tile = [&](int t, bool masked, auto buf) {
        const char* kb = smem + decltype(buf)::value * kStage;
        const char* vb = kb + kTile;
        f32x16 s0, s1;
        // ---- S^T = K Q^T for the two 32-key tiles (+ the reference step), masked on the ragged tile ----
        auto scores = [&]() {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const bf16x8 ka0 = *reinterpret_cast<const bf16x8*>(kb + k_off[0]);
            const bf16x8 kc0 = *reinterpret_cast<const bf16x8*>(kb + 32 * 128 + k_off[0]);
            if (have_ref) {
                asm volatile("" ::: "memory");   // keep this a real (wave-uniform) branch
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kx.v, qx.v, zero, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kx.v, qx.v, zero, 0, 0, 0);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, qf[0], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc0, qf[0], s1, 0, 0, 0);
            } else {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, qf[0], zero, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc0, qf[0], zero, 0, 0, 0);
            }
#pragma unroll
            for (int s = 1; s < 4; ++s) {
                const bf16x8 ka = *reinterpret_cast<const bf16x8*>(kb + k_off[s]);
                const bf16x8 kc = *reinterpret_cast<const bf16x8*>(kb + 32 * 128 + k_off[s]);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[s], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc, qf[s], s1, 0, 0, 0);
            }
            // register r of tile u <-> key  t*64 + 32 u + (r&3) + 8 (r>>2) + 4 lh
            if (masked) {
                // Only the ragged last tile takes this branch.  The empty asm keeps it a real (wave-uniform)
                // branch: if-converted, the ~100 compare/select instructions would run for every tile.
                asm volatile("" ::: "memory");
                const int kbase = t * KV + 4 * lh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase + (r & 3) + 8 * (r >> 2);
                    if (key >= a.N) s0[r] = -__builtin_inff();
                    if (key + 32 >= a.N) s1[r] = -__builtin_inff();
                }
            }
        };
        float psum;
        auto exponentials = [&]() {
            psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#ifndef EQUSS_ABL_ATTNOEXP   // (EQUSS_ABL_*: developer builds that leave one cost out, tools/build_ablations.sh; wrong results by design)
                s0[r] = __builtin_amdgcn_exp2f(s0[r]);
                s1[r] = __builtin_amdgcn_exp2f(s1[r]);
#else
                s0[r] = fmaf(s0[r], 1e-9f, 0.01f); s1[r] = fmaf(s1[r], 1e-9f, 0.01f);
#endif
#ifndef EQUSS_ABL_ATTNOSUM
                psum += s0[r] + s1[r];
#else
                if (r == 0) psum += s0[r] + s1[r];
#endif
            }
        };
        scores();
        exponentials();
        const bool out_of_range = !(psum < kBig) || (t == 0 && !(psum > kTiny));
        if (__builtin_amdgcn_ballot_w64(out_of_range) != 0) {
            asm volatile("" ::: "memory");   // keep this a real branch: it is rare
            scores();
            float mloc = s0[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s0[r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s1[r]);
            mloc = fmaxf(mloc, __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(xhalf, __builtin_bit_cast(int, mloc))));
            // The first tile may set any reference (O and l are still zero); later ones only raise it.
            const float grow = fmaxf(mloc, t == 0 ? -__builtin_inff() : 0.f);
            // new reference = the maximum rounded UP to bf16: every score of this tile then comes out <= 0
            const float target = m_run + (grow > -__builtin_inff() ? grow : 0.f);
            u16 mb = equss::f2bf(target);
            if (equss::bf2f(mb) < target) mb = (u16)(target > 0.f ? mb + 1 : mb - 1);
            const float m_new = equss::bf2f(mb);
            const float shift = m_new - m_run;
            const float alpha = t == 0 ? 0.f : __builtin_amdgcn_exp2f(-shift);
            m_run = m_new;
            qx.w[0] = lh == 0 ? (uint32_t)(mb ^ 0x8000u) : 0u;   // -m_run as bf16 in the first slot
            have_ref = true;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] -= shift; s1[r] -= shift; }
            exponentials();
        }
        l_run += psum;

        // ---- O^T += V^T P^T: P fragments straight from the S^T registers ----------------------
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                union { bf16x8 v; uint32_t w[4]; } pf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = u == 0 ? s0[8 * sp + 2 * j] : s1[8 * sp + 2 * j];
                    const float hi = u == 0 ? s0[8 * sp + 2 * j + 1] : s1[8 * sp + 2 * j + 1];
#ifndef EQUSS_ABL_ATTNOCVT
                    pf.w[j] = equss::pack_bf16(lo, hi);
#else
                    pf.w[j] = __builtin_bit_cast(uint32_t, lo) ^ __builtin_bit_cast(uint32_t, hi);
#endif
                }
                const int krow = (32 * u + 16 * sp) * 128;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    union { bf16x8 v; bf16x4 hlf[2]; } vf;
                    vf.hlf[0] = ds_read_tr16(vb + krow + v_off[dt]);
                    vf.hlf[1] = ds_read_tr16(vb + krow + 8 * 128 + v_off[dt]);
#ifndef EQUSS_ABL_ATTNOPV
                    if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o0, 0, 0, 0);
                    else         o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o1, 0, 0, 0);
#else
                    asm volatile("" ::"v"(vf.v), "v"(pf.v));
#endif
                }
            }
        }
    };

    stage(0, 0);
    const int full_tiles = a.N / KV;  // tiles whose 64 keys are all valid
    for (int t = 0; t < a.ntiles; t += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < a.ntiles) stage(1, t + 1);
        if (wave_active) tile(t, t >= full_tiles, std::integral_constant<int, 0>());
        if (t + 1 >= a.ntiles) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 2 < a.ntiles) stage(0, t + 2);
        if (wave_active) tile(t + 1, t + 1 >= full_tiles, std::integral_constant<int, 1>());
    }

    // ---- normalise and store: lane (query, half) holds d = 32 dt + 8 g + 4 lh + 0..3 -----------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_valid) {
        u16* orow = a.out + ((size_t)b * a.N + qrow) * C + h * HD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float x0 = (dt == 0 ? o0[4 * g + 0] : o1[4 * g + 0]) * inv;
                const float x1 = (dt == 0 ? o0[4 * g + 1] : o1[4 * g + 1]) * inv;
                const float x2 = (dt == 0 ? o0[4 * g + 2] : o1[4 * g + 2]) * inv;
                const float x3 = (dt == 0 ? o0[4 * g + 3] : o1[4 * g + 3]) * inv;
                uint2 pk;
                pk.x = equss::pack_bf16(x0, x1);
                pk.y = equss::pack_bf16(x2, x3);
                *reinterpret_cast<uint2*>(orow + 32 * dt + 8 * g + 4 * lh) = pk;
            }
        }
    }
}

int g_waves = 0;  // developer override of the waves per workgroup (0 = default)

}  // namespace

extern "C" void equss_dev_attn_diag(int d) { g_waves = d; }  // tuning knob, not part of the ABI

namespace equss {

int attention_bf16(const void* qkv, void* out, int B, int N, int H, float scale, hipStream_t s) {
    if (!qkv || !out) return fail(EQUSS_E_NULL, "attention_bf16: null pointer argument");
    if (B <= 0 || N <= 0 || H <= 0) return fail(EQUSS_E_SHAPE, "attention_bf16: B=%d N=%d H=%d must be positive", B, N, H);
    if (((uintptr_t)qkv % 16) || ((uintptr_t)out % 16))
        return fail(EQUSS_E_ALIGN, "attention_bf16: qkv and out must be 16-byte aligned");
    AttnArgs a;
    a.qkv = reinterpret_cast<const u16*>(qkv);
    a.out = reinterpret_cast<u16*>(out);
    a.B = B; a.N = N; a.H = H;
    // 128-query workgroups (4 waves, 4 resident per CU): they stage K/V twice as often per query as
    // 256-query ones, but fewer waves idle past the last query and the last round of workgroups fills
    // the chip better (1601 tokens: 13 x 4 waves instead of 7 x 8); measured 5 % faster.
    const int nw = g_waves == 8 ? 8 : 4;
    const int qb = nw * QW;
    a.nqb = (N + qb - 1) / qb;
    a.ntiles = (N + KV - 1) / KV;
    a.c = scale * 1.4426950408889634f;
    const long grid = (long)a.nqb * B * H;
    if (grid > 0x7fffffffL) return fail(EQUSS_E_SHAPE, "attention_bf16: grid too large");
    switch (nw) {
        case 8: hipLaunchKernelGGL(attention_kernel<8>, dim3((unsigned)grid), dim3(512), 2 * kStage, s, a); break;
        default: hipLaunchKernelGGL(attention_kernel<4>, dim3((unsigned)grid), dim3(256), 2 * kStage, s, a); break;
    }
    return check_launch("attention_kernel");
}

}  // namespace equss

extern "C" int equss_attention_bf16(const void* qkv, void* out, int B, int N, int H, float scale, void* stream) {
    return equss::attention_bf16(qkv, out, B, N, H, scale, reinterpret_cast<hipStream_t>(stream));
}
