// Linear + cluster probes of the eval loop (model/evaluator.py:46-111, is_crf = False) without
// materialising the bilinearly up-sampled (B, D, H, W) feature map.
//
// The reference up-samples the D-channel map to the label size (420 MB per 320x320 image at
// D = 1024), then applies a 1x1 conv (linear probe) and a cosine-similarity lookup (cluster probe)
// per output pixel.  Both probes are linear in the feature, and bilinear interpolation is a convex
// combination of 4 low-resolution neighbours, so
//     W f_up + b              = interp(W f + b)
//     <f_up, c^>              = interp(<f, c^>)
//     |f_up|^2                = sum_ab w_a w_b <f_a, f_b>      (a, b over the 4 neighbours)
// Three kernels:
//   probe_table_kernel  exact-fp32 MFMA GEMM of every low-res row against the 64 stacked probe
//                       vectors [linear weights ; normalised clusters] -> table[row][0..63]
//   probe_gram_kernel   the 5 neighbour inner products per low-res row -> table[row][64..68]
//   probe_pixel_kernel  one wave per 8x8 block of output pixels: interpolates the 27+27 table
//                       values and the norm, arg-max, cross-entropy, per-wave loss partials
// followed by a fixed-order reduction of the partials (deterministic losses).
//
// Training the probes (train.py:80-97 with the probes' parameters in the optimiser): the same three kernels, the pixel
// kernel in its GRAD form.  With T the table (low-resolution logits / inner products) the losses are functions of T only, and
//     d(sum of cross-entropies) / dT[t][c]   = sum over pixels of  w_t(pixel) * (softmax(logit)[c] - [c == label])
//     d(-sum of max cosines) / dT[t][32 + c] = sum over pixels of -w_t(pixel) * [c == arg-max] / |f_up|
// (the arg-max and the norm are constants of the reference's graph too: one_hot(argmax), detached features), so the
// kernel scatters those terms into a (rows, 64) gradient table: per wave through a small LDS window of the low-resolution
// cells its 8x8 pixels touch, then one 256-byte atomic row add per cell.  The parameter gradients are table^T-shaped
// products of that gradient table with the features, done by the caller.
#include "equss_common.h"

namespace {

constexpr int kTableStride = 72;   // floats per low-res row: 32 linear | 32 cluster | 5 gram | 3 pad
constexpr int kProbeSlots = 64;
constexpr int kGramOff = 64;
constexpr float kUnusedScore = -1e30f;

// ---------------------------------------------------------------------------------------------
// table[r][c] = <feat[r], probes[c]> + bias[c] in exact fp32 (v_mfma_f32_32x32x2_f32).
// The probe matrix sits on the MFMA A side, so a lane ends up with 4 consecutive columns of one
// row and stores them as one 16-byte piece.  A wave owns 64 rows x 64 columns; operands come
// straight from global memory as 16-byte pieces (4 k-steps per load) with the next k-block
// prefetched into registers while the current one is multiplied.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void probe_table_kernel(const float* __restrict__ feat, int64_t ld_feat,
                                                            const float* __restrict__ probes,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ table, int rows, int D,
                                                            int n_lin, int n_clu) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wave) * 64;
    if (r0 >= rows) return;
    const int l32 = lane & 31, kh = lane >> 5;
    const float* fa[2];
    const float* pb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int r = min(r0 + h * 32 + l32, rows - 1);
        fa[h] = feat + (int64_t)r * ld_feat + kh * 4;
        pb[h] = probes + (int64_t)(h * 32 + l32) * D + kh * 4;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    f32x4 f_cur[2], p_cur[2], f_nxt[2], p_nxt[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f_cur[h] = *reinterpret_cast<const f32x4*>(fa[h]);
        p_cur[h] = *reinterpret_cast<const f32x4*>(pb[h]);
    }
    for (int kb = 0; kb < D; kb += 8) {
        const int kn = (kb + 8 < D) ? kb + 8 : kb;   // last iteration re-reads its own block
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f_nxt[h] = *reinterpret_cast<const f32x4*>(fa[h] + kn);
            p_nxt[h] = *reinterpret_cast<const f32x4*>(pb[h] + kn);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int rh = 0; rh < 2; ++rh)
                    acc[ch][rh] = __builtin_amdgcn_mfma_f32_32x32x2f32(p_cur[ch][s], f_cur[rh][s], acc[ch][rh], 0, 0, 0);
#pragma unroll
        for (int h = 0; h < 2; ++h) { f_cur[h] = f_nxt[h]; p_cur[h] = p_nxt[h]; }
    }
    // acc[ch][rh][v]: column ch*32 + (v/4)*8 + kh*4 + v%4 of row r0 + rh*32 + l32
#pragma unroll
    for (int rh = 0; rh < 2; ++rh) {
        const int r = r0 + rh * 32 + l32;
        if (r >= rows) continue;
        float* trow = table + (int64_t)r * kTableStride;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = ch * 32 + g * 8 + kh * 4;
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias + c);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // unused class slots get a score no arg-max picks and whose exp() is 0
                    const bool used = (c + e - ch * 32) < (ch ? n_clu : n_lin);
                    o[e] = used ? acc[ch][rh][g * 4 + e] + b[e] : kUnusedScore;
                }
                *reinterpret_cast<f32x4*>(trow + c) = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Per low-res cell (b, y, x), with xr = min(x+1, w-1), yd = min(y+1, h-1):
//   g0 = <f(y,x), f(y,x)>   g1 = <f(y,x), f(y,xr)>   g2 = <f(y,x), f(yd,x)>
//   g3 = <f(y,x), f(yd,xr)> g4 = <f(y,xr), f(yd,x)>
// One wave per cell, 16-byte loads, fp32 accumulation, shuffle reduction.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void probe_gram_kernel(const float* __restrict__ feat, int64_t ld_feat,
                                                        float* __restrict__ table, int B, int h, int w, int D) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * h * w) return;
    const int x = r % w, y = (r / w) % h;
    const int dx = (x + 1 < w) ? 1 : 0, dy = (y + 1 < h) ? w : 0;
    const float* f0 = feat + (int64_t)r * ld_feat;
    const float* f1 = feat + (int64_t)(r + dx) * ld_feat;
    const float* f2 = feat + (int64_t)(r + dy) * ld_feat;
    const float* f3 = feat + (int64_t)(r + dy + dx) * ld_feat;
    float g[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(f0 + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(f1 + c);
        const f32x4 d = *reinterpret_cast<const f32x4*>(f2 + c);
        const f32x4 e = *reinterpret_cast<const f32x4*>(f3 + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            g[0] = fmaf(a[i], a[i], g[0]);
            g[1] = fmaf(a[i], b[i], g[1]);
            g[2] = fmaf(a[i], d[i], g[2]);
            g[3] = fmaf(a[i], e[i], g[3]);
            g[4] = fmaf(b[i], d[i], g[4]);
        }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) g[i] = equss::group_sum<64>(g[i]);
    if (lane < 5) {
        float v = g[0];
        v = lane == 1 ? g[1] : v; v = lane == 2 ? g[2] : v; v = lane == 3 ? g[3] : v; v = lane == 4 ? g[4] : v;
        table[(int64_t)r * kTableStride + kGramOff + lane] = v;
    }
}

// Sum over the 16 lanes of a DPP row (quad swaps, half-row and row mirrors), in every lane of the row.
__device__ __forceinline__ float row16_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, false));
    return x;
}

// Source index and weights of torch's bilinear up-sampling with align_corners = False
// (F.interpolate, evaluator.py:54): src = scale * (dst + 0.5) - 0.5 clamped at 0, in fp32.
__device__ __forceinline__ void bilinear_tap(int dst, float scale, int in_size, int& i0, int& step, float& w0, float& w1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = min((int)src, in_size - 1);
    step = (i0 < in_size - 1) ? 1 : 0;
    w1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
    w0 = 1.f - w1;
}

struct ProbePixelArgs {
    const float* table;
    const int64_t* label;
    int64_t* lin_pred;
    int64_t* clu_pred;
    float* norm_out;         // (B, H, W) |f_up| clamped at 1e-12, or null
    float* grad;             // GRAD form: (B*h*w, 64) gradient table, zero on entry
    float* partial;          // (blocks, 4): ce sum, valid count, max-inner sum, unused
    int B, h, w, H, W, n_lin, n_clu;
    float scale_y, scale_x;
};

constexpr int kTilesPerWave = 2;
constexpr int kWinCells = 16;      // GRAD form: low-resolution cells in a wave's LDS window (integer factor 8: 3 x 3)

// One wave handles kTilesPerWave consecutive 8x8 blocks of output pixels (lane -> (lane/8, lane%8)).
// With the usual integer up-sampling factor 8 the lanes of a wave share at most 2x2 low-res cells, so
// the 16-byte table reads below are broadcasts served by a handful of cache lines.  Loss terms are
// summed per lane over the tiles, then over the wave and the 4 waves of the block in a fixed order.
//
// GRAD: the gradient terms of the header go into win[cell][class] with LDS atomics and the window is added to a.grad with one
// atomic per cell and lane: 256 consecutive bytes per instruction.  LDS float atomics retire about one lane per three
// cycles per CU whatever the addresses, so where the 16 pixels of a quadrant share their taps (integer factor 8) the
// cross-entropy terms are first summed over the quadrant's DPP row: 4 lanes instead of 64 add each (class, tap).  A tile whose pixels touch more than kWinCells cells (down-sampling, factors
// below 4) adds straight to memory.  The order of the atomic adds is not fixed: gradients agree to rounding, not bit for bit.
template <bool GRAD>
__global__ __launch_bounds__(256, GRAD ? 2 : 4) void probe_pixel_kernel(ProbePixelArgs a) {
    __shared__ float red[4][4];
    __shared__ float win_s[GRAD ? 4 * kWinCells * kProbeSlots : 1];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tiles_x = (a.W + 7) >> 3, tiles_y = (a.H + 7) >> 3;
    const int64_t n_tiles = (int64_t)a.B * tiles_y * tiles_x;
    const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * kTilesPerWave;
    float acc_ce = 0.f, acc_n = 0.f, acc_in = 0.f;
    for (int it = 0; it < kTilesPerWave; ++it) {
        const int64_t tile = tile0 + it;
        if (tile >= n_tiles) break;
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y), b = (int)(tile / ((int64_t)tiles_x * tiles_y));
        // lane -> pixel of the 8x8 tile: row-major; GRAD: quadrant-major (16 consecutive lanes = a 4x4 quadrant), because with
        // the usual up-sampling factor 8 a quadrant's pixels share their four taps and their gradient terms can be summed
        // over a 16-lane DPP row before they go to LDS
        const int X = tx * 8 + (GRAD ? ((lane >> 4) & 1) * 4 + (lane & 3) : (lane & 7));
        const int Y = ty * 8 + (GRAD ? (lane >> 5) * 4 + ((lane >> 2) & 3) : (lane >> 3));
        const bool live = X < a.W && Y < a.H;
        const int Xc = min(X, a.W - 1), Yc = min(Y, a.H - 1);

        int y0, sy, x0, sx;
        float wy0, wy1, wx0, wx1;
        bilinear_tap(Yc, a.scale_y, a.h, y0, sy, wy0, wy1);
        bilinear_tap(Xc, a.scale_x, a.w, x0, sx, wx0, wx1);
        const float* t00 = a.table + ((int64_t)(b * a.h + y0) * a.w + x0) * kTableStride;
        const float* t01 = t00 + sx * kTableStride;
        const float* t10 = t00 + (int64_t)sy * a.w * kTableStride;
        const float* t11 = t10 + sx * kTableStride;
        const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
        const int64_t lab = live ? a.label[((int64_t)b * a.H + Yc) * a.W + Xc] : -1;

        // GRAD: the window of cells this tile touches (taps grow with the pixel index: corners give the range)
        float* win = win_s + wave * kWinCells * kProbeSlots;
        int ymin = 0, xmin = 0, nx = 1, ncell = 0, j00 = 0, j01 = 0, j10 = 0, j11 = 0;
        bool windowed = false;
        if constexpr (GRAD) {
            ymin = __shfl(y0, 0, 64); xmin = __shfl(x0, 0, 64);
            const int ymax = __shfl(y0 + sy, 63, 64), xmax = __shfl(x0 + sx, 63, 64);
            nx = xmax - xmin + 1;
            ncell = nx * (ymax - ymin + 1);
            windowed = ncell <= kWinCells;
            if (windowed) {
                for (int j = 0; j < ncell; ++j) win[j * kProbeSlots + lane] = 0.f;
                j00 = ((y0 - ymin) * nx + (x0 - xmin)) * kProbeSlots;
                j01 = j00 + sx * kProbeSlots; j10 = j00 + sy * nx * kProbeSlots; j11 = j10 + sx * kProbeSlots;
            }
        }
        // do the 16 pixels of every quadrant share their taps?  (wave-uniform)
        bool quads = false;
        if constexpr (GRAD) {
            if (windowed) {
                const int lead = lane & 48;
                const bool same = j00 == __shfl(j00, lead, 64) && sx == __shfl(sx, lead, 64) && sy == __shfl(sy, lead, 64);
                quads = __builtin_amdgcn_ballot_w64(same) == ~0ull;
            }
        }
        // one gradient term of this pixel, spread over its four taps
        auto scatter = [&](int c, float v) {
            if (windowed) {
                atomicAdd(win + j00 + c, w00 * v); atomicAdd(win + j01 + c, w01 * v);
                atomicAdd(win + j10 + c, w10 * v); atomicAdd(win + j11 + c, w11 * v);
            } else {
                float* g00 = a.grad + (t00 - a.table) / kTableStride * kProbeSlots;
                float* g01 = a.grad + (t01 - a.table) / kTableStride * kProbeSlots;
                float* g10 = a.grad + (t10 - a.table) / kTableStride * kProbeSlots;
                float* g11 = a.grad + (t11 - a.table) / kTableStride * kProbeSlots;
                atomicAdd(g00 + c, w00 * v); atomicAdd(g01 + c, w01 * v);
                atomicAdd(g10 + c, w10 * v); atomicAdd(g11 + c, w11 * v);
            }
        };

        // ---- linear probe: interpolated logits, first arg-max, log-sum-exp ----------------------
        // (class slots beyond n_lin / n_clu hold kUnusedScore, so no masking is needed here)
        float logit[32];
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            const f32x4 v00 = *reinterpret_cast<const f32x4*>(t00 + c), v01 = *reinterpret_cast<const f32x4*>(t01 + c);
            const f32x4 v10 = *reinterpret_cast<const f32x4*>(t10 + c), v11 = *reinterpret_cast<const f32x4*>(t11 + c);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                logit[c + e] = fmaf(w11, v11[e], fmaf(w10, v10[e], fmaf(w01, v01[e], w00 * v00[e])));
            if ((c & 4) != 0) __builtin_amdgcn_sched_barrier(0);   // at most two chunks of loads in flight
        }
        float best = logit[0];
        int best_i = 0;
#pragma unroll
        for (int c = 1; c < 32; ++c)
            if (logit[c] > best) { best = logit[c]; best_i = c; }
        const bool valid = live && lab >= 0 && lab < a.n_lin;
        float se = 0.f, picked = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            se += __expf(logit[c] - best);
            picked = (c == (int)lab) ? logit[c] : picked;
        }
        acc_ce += valid ? (best + __logf(se)) - picked : 0.f;
        acc_n += valid ? 1.f : 0.f;
        if constexpr (GRAD) {
            const float rse = 1.0f / se;
            if (quads) {
#pragma unroll
                for (int c = 0; c < 32; ++c) {
                    if (c >= a.n_lin) break;
                    const float gterm = valid ? __expf(logit[c] - best) * rse - (c == (int)lab ? 1.f : 0.f) : 0.f;
                    const float r00 = row16_sum(w00 * gterm), r01 = row16_sum(w01 * gterm);
                    const float r10 = row16_sum(w10 * gterm), r11 = row16_sum(w11 * gterm);
                    if ((lane & 15) == 0) {
                        atomicAdd(win + j00 + c, r00); atomicAdd(win + j01 + c, r01);
                        atomicAdd(win + j10 + c, r10); atomicAdd(win + j11 + c, r11);
                    }
                }
            } else if (valid) {
#pragma unroll
                for (int c = 0; c < 32; ++c)
                    if (c < a.n_lin) scatter(c, __expf(logit[c] - best) * rse - (c == (int)lab ? 1.f : 0.f));
            }
        }

        // ---- cluster probe: interpolated inner products, first arg-max, cosine of the maximum ----
        float cbest = kUnusedScore;
        int cbest_i = 0;
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            const f32x4 v00 = *reinterpret_cast<const f32x4*>(t00 + 32 + c), v01 = *reinterpret_cast<const f32x4*>(t01 + 32 + c);
            const f32x4 v10 = *reinterpret_cast<const f32x4*>(t10 + 32 + c), v11 = *reinterpret_cast<const f32x4*>(t11 + 32 + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = fmaf(w11, v11[e], fmaf(w10, v10[e], fmaf(w01, v01[e], w00 * v00[e])));
                if (c + e == 0 || v > cbest) { cbest = v; cbest_i = c + e; }
            }
            if ((c & 4) != 0) __builtin_amdgcn_sched_barrier(0);
        }
        // |f_up|^2 from the neighbour Gram entries.  A clamped tap (sx or sy = 0) makes t01 / t10 / t11
        // alias their neighbour, and the Gram kernel clamps the same way, so these are the right inner
        // products in every case.
        const f32x4 q00 = *reinterpret_cast<const f32x4*>(t00 + kGramOff);   // self, right, down, diagonal
        const float a00 = t00[kGramOff + 4];                                  // <f01, f10>
        const float g01 = t01[kGramOff], d01 = t01[kGramOff + 2];             // <f01,f01>, <f01,f11>
        const float g10 = t10[kGramOff], r10 = t10[kGramOff + 1];             // <f10,f10>, <f10,f11>
        const float g11 = t11[kGramOff];
        float n2 = w00 * w00 * q00[0] + w01 * w01 * g01 + w10 * w10 * g10 + w11 * w11 * g11;
        n2 += 2.f * (w00 * w01 * q00[1] + w10 * w11 * r10 + w00 * w10 * q00[2] + w01 * w11 * d01
                     + w00 * w11 * q00[3] + w01 * w10 * a00);
        const float nrm = fmaxf(sqrtf(fmaxf(n2, 0.f)), 1e-12f);   // F.normalize eps (evaluator.py:96)
        acc_in += live ? cbest / nrm : 0.f;
        if constexpr (GRAD) {
            if (live) scatter(32 + cbest_i, -1.0f / nrm);
            if (windowed) {
                for (int j = 0; j < ncell; ++j) {
                    const float v = win[j * kProbeSlots + lane];
                    const int64_t cell = ((int64_t)b * a.h + ymin + j / nx) * a.w + xmin + j % nx;
                    if (v != 0.f) atomicAdd(a.grad + cell * kProbeSlots + lane, v);
                }
            }
        }

        if (live) {
            const int64_t o = ((int64_t)b * a.H + Y) * a.W + X;
            a.lin_pred[o] = best_i;
            a.clu_pred[o] = cbest_i;
            if (a.norm_out) a.norm_out[o] = nrm;
        }
    }
    const float s_ce = equss::group_sum<64>(acc_ce);
    const float s_n = equss::group_sum<64>(acc_n);
    const float s_in = equss::group_sum<64>(acc_in);
    if (lane == 0) { red[wave][0] = s_ce; red[wave][1] = s_n; red[wave][2] = s_in; }
    __syncthreads();
    if (threadIdx.x == 0) {
        f32x4 p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) { p[0] += red[k][0]; p[1] += red[k][1]; p[2] += red[k][2]; }
        *reinterpret_cast<f32x4*>(a.partial + (int64_t)blockIdx.x * 4) = p;
    }
}

// losses[0] = mean cross-entropy over valid pixels (nn.CrossEntropyLoss, evaluator.py:72-80; 0/0 =
// NaN when no pixel is labelled, as the reference's mean over an empty selection);
// losses[1] = -mean over all pixels of the maximal cosine (evaluator.py:101-106 with alpha None).
__global__ __launch_bounds__(256) void probe_loss_reduce(const float* __restrict__ partial, int64_t n, double pixels,
                                                        float* __restrict__ losses, int n_losses) {
    __shared__ double red[3][256];
    double s[3] = {0.0, 0.0, 0.0};
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(partial + i * 4);
        s[0] += p[0]; s[1] += p[1]; s[2] += p[2];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
#pragma unroll
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        losses[0] = (float)(red[0][0] / red[1][0]);
        losses[1] = (float)(-red[2][0] / pixels);
        if (n_losses > 2) losses[2] = (float)red[1][0];   // labelled pixels: what the cross-entropy gradients are divided by
    }
}

// confusion[pred, label] += 1 over pixels whose label and prediction both lie in [0, num_classes)
// (UnSegMetrics.update, model/metric.py:45-58).  Per-workgroup histogram in LDS, then one 64-bit
// integer atomic per non-empty bin: integer sums, so the result does not depend on the order.
__global__ __launch_bounds__(256) void confusion_kernel(const int64_t* __restrict__ preds,
                                                       const int64_t* __restrict__ label,
                                                       unsigned long long* __restrict__ confusion, int64_t n,
                                                       int num_classes, int bins) {
    extern __shared__ unsigned int hist[];
    for (int i = threadIdx.x; i < bins; i += 256) hist[i] = 0u;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int64_t p = preds[i], l = label[i];
        if (l >= 0 && l < num_classes && p >= 0 && p < num_classes) atomicAdd(&hist[(int)p * num_classes + (int)l], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256)
        if (hist[i]) atomicAdd(&confusion[i], (unsigned long long)hist[i]);
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t equss_probe_workspace_bytes(int B, int h, int w, int H, int W) {
    if (B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return 0;
    const size_t table = align256((size_t)B * h * w * kTableStride * sizeof(float));
    const size_t tiles = (size_t)B * ((H + 7) / 8) * ((W + 7) / 8);
    const size_t blocks = (tiles + 4 * kTilesPerWave - 1) / (4 * kTilesPerWave);
    return table + align256(blocks * 4 * sizeof(float));
}

namespace {
int probe_impl(const float* feat, int64_t ld_feat, const float* probes, const float* bias,
               const int64_t* label, int64_t* linear_preds, int64_t* cluster_preds,
               float* losses, float* feature_norm, float* grad_table, void* workspace, size_t workspace_bytes,
               int B, int h, int w, int D, int H, int W, int n_linear, int n_cluster,
               void* stream) {
    if (!feat || !probes || !bias || !label || !linear_preds || !cluster_preds || !losses || !workspace)
        return equss::fail(EQUSS_E_NULL, "equss_probe_eval: null pointer argument");
    if (B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || D <= 0 || (D % 8) != 0)
        return equss::fail(EQUSS_E_SHAPE, "equss_probe_eval: bad shape B=%d h=%d w=%d D=%d H=%d W=%d (D %% 8 == 0)", B, h, w, D, H, W);
    if (n_linear < 1 || n_linear > 32 || n_cluster < 1 || n_cluster > 32)
        return equss::fail(EQUSS_E_UNSUPPORTED, "equss_probe_eval: %d linear / %d cluster classes (1..32 each)", n_linear, n_cluster);
    if ((int64_t)B * h * w > 0x7fffffff || (int64_t)B * H * W > ((int64_t)1 << 40))
        return equss::fail(EQUSS_E_SHAPE, "equss_probe_eval: too many low-resolution cells");
    if (ld_feat < D || (ld_feat % 4) != 0 || ((uintptr_t)feat % 16) != 0 || ((uintptr_t)probes % 16) != 0 ||
        ((uintptr_t)bias % 16) != 0 || ((uintptr_t)workspace % 256) != 0)
        return equss::fail(EQUSS_E_ALIGN, "equss_probe_eval: feat/probes/bias need 16-byte alignment, ld_feat %% 4 == 0, workspace 256-byte aligned");
    if (workspace_bytes < equss_probe_workspace_bytes(B, h, w, H, W))
        return equss::fail(EQUSS_E_WORKSPACE, "equss_probe_eval: workspace %zu < %zu bytes", workspace_bytes,
                           equss_probe_workspace_bytes(B, h, w, H, W));
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * h * w;
    float* table = (float*)workspace;
    float* partial = (float*)((char*)workspace + align256((size_t)rows * kTableStride * sizeof(float)));
    int rc;
    {
        equss::ProfScope ps(EQUSS_K_PROBE_TABLE, s);
        hipLaunchKernelGGL(probe_table_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, feat, ld_feat, probes, bias, table, rows, D,
                           n_linear, n_cluster);
        rc = equss::check_launch("probe_table_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(probe_gram_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, feat, ld_feat, table, B, h, w, D);
        rc = equss::check_launch("probe_gram_kernel");
        if (rc) return rc;
    }
    ProbePixelArgs a;
    a.table = table; a.label = label; a.lin_pred = linear_preds; a.clu_pred = cluster_preds; a.norm_out = feature_norm; a.grad = grad_table; a.partial = partial;
    a.B = B; a.h = h; a.w = w; a.H = H; a.W = W; a.n_lin = n_linear; a.n_clu = n_cluster;
    a.scale_y = (float)h / (float)H;   // area_pixel_compute_scale of F.interpolate without scale_factor
    a.scale_x = (float)w / (float)W;
    const int64_t tiles = (int64_t)B * ((H + 7) / 8) * ((W + 7) / 8);
    {
        equss::ProfScope ps(EQUSS_K_PROBE_PIXELS, s);
        const int64_t blocks = (tiles + 4 * kTilesPerWave - 1) / (4 * kTilesPerWave);
        if (grad_table) {
            if (hipMemsetAsync(grad_table, 0, (size_t)rows * kProbeSlots * sizeof(float), s) != hipSuccess)
                return equss::fail(EQUSS_E_UNSUPPORTED, "equss_probe_train: cannot clear the gradient table");
            hipLaunchKernelGGL(probe_pixel_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL(probe_pixel_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
        }
        rc = equss::check_launch("probe_pixel_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(probe_loss_reduce, dim3(1), dim3(256), 0, s, partial, blocks, (double)B * H * W, losses, grad_table ? 3 : 2);
        rc = equss::check_launch("probe_loss_reduce");
    }
    return rc;
}
}  // namespace

extern "C" int equss_probe_eval(const float* feat, int64_t ld_feat, const float* probes, const float* bias,
                                const int64_t* label, int64_t* linear_preds, int64_t* cluster_preds,
                                float* losses, float* feature_norm, void* workspace, size_t workspace_bytes,
                                int B, int h, int w, int D, int H, int W, int n_linear, int n_cluster,
                                void* stream) {
    return probe_impl(feat, ld_feat, probes, bias, label, linear_preds, cluster_preds, losses, feature_norm, nullptr, workspace,
                      workspace_bytes, B, h, w, D, H, W, n_linear, n_cluster, stream);
}

extern "C" int equss_probe_train(const float* feat, int64_t ld_feat, const float* probes, const float* bias,
                                 const int64_t* label, int64_t* linear_preds, int64_t* cluster_preds,
                                 float* losses, float* grad_table, void* workspace, size_t workspace_bytes,
                                 int B, int h, int w, int D, int H, int W, int n_linear, int n_cluster,
                                 void* stream) {
    if (!grad_table || ((uintptr_t)grad_table % 256) != 0)
        return equss::fail(EQUSS_E_ALIGN, "equss_probe_train: grad_table must be a 256-byte aligned (B*h*w, 64) fp32 buffer");
    return probe_impl(feat, ld_feat, probes, bias, label, linear_preds, cluster_preds, losses, nullptr, grad_table, workspace,
                      workspace_bytes, B, h, w, D, H, W, n_linear, n_cluster, stream);
}

extern "C" int equss_confusion_update(const int64_t* preds, const int64_t* label, int64_t* confusion, int64_t n,
                                      int num_classes, int num_pred_classes, void* stream) {
    if (!preds || !label || !confusion) return equss::fail(EQUSS_E_NULL, "equss_confusion_update: null pointer argument");
    if (n < 0 || num_classes < 1 || num_pred_classes < num_classes || (int64_t)num_classes * num_pred_classes > 8192)
        return equss::fail(EQUSS_E_SHAPE, "equss_confusion_update: n=%lld, %d classes, %d predicted classes (at most 8192 bins)",
                           (long long)n, num_classes, num_pred_classes);
    if (n == 0) return EQUSS_OK;
    const int bins = num_classes * num_pred_classes;
    int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)blocks), dim3(256), bins * sizeof(unsigned int), s, preds, label,
                       reinterpret_cast<unsigned long long*>(confusion), n, num_classes, bins);
    return equss::check_launch("confusion_kernel");
}
