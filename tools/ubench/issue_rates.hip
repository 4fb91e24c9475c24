// Developer micro-benchmark (not part of the product): issue rates of v_exp_f32, v_fma_f32,
// v_pk_fma_f32 and the 32x32x16 bf16 MFMA on gfx950, alone and mixed across the waves of one SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef KITERS
#define KITERS 2000
#endif
constexpr int kIters = KITERS;

template <int MODE_EVEN, int MODE_ODD>   // 0 exp, 1 fma, 2 pk_fma, 3 mfma, 4 mfma+exp same wave, 5 idle, 6 max3, 7 cvt_pk
__global__ __launch_bounds__(1024) void k(float* out, float seed, const bf16x8* rnd = nullptr) {
    const int wave = threadIdx.x >> 6;
    const int mode = ((wave >> 2) & 1) ? MODE_ODD : MODE_EVEN;   // waves 0-3 / 4-7 / ...: both kinds land on every SIMD
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + i * 0.001f + threadIdx.x * 1e-6f;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    if (mode == 0) {
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        }
    } else if (mode == 1) {
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
        }
    } else if (mode == 2) {
        f32x2 p[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = f32x2{v[2 * i], v[2 * i + 1]};
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[2 * i] = p[i].x; v[2 * i + 1] = p[i].y; }
    } else if (mode == 3) {
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            }
        }
    } else if (mode == 4) {
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                asm volatile("v_exp_f32 %0, %0" : "+v"(v[2 * i]));
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                asm volatile("v_exp_f32 %0, %0" : "+v"(v[2 * i + 1]));
            }
        }
    } else if (mode == 8) {
        // operands with realistic bit activity: 8 distinct random fragments per lane, rotated
        bf16x8 ra[4], rb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra[i] = rnd[(threadIdx.x * 8 + i) & 4095]; rb[i] = rnd[(threadIdx.x * 8 + 4 + i) & 4095]; }
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[i & 3], rb[(i + 1) & 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[(i + 2) & 3], rb[i & 3], acc1, 0, 0, 0);
            }
        }
    } else if (mode == 6) {
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(v[i]));
        }
    } else if (mode == 7) {
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[i]));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + acc0[i] + acc1[i];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int E, int O>
void run(const char* name, float* out, int instr_even, int instr_odd, const bf16x8* rnd = nullptr) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<E, O>), dim3(256), dim3(1024), 0, 0, out, 1.0f, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<E, O>), dim3(256), dim3(1024), 0, 0, out, 1.0f, rnd);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 5;
    // per SIMD: 4 waves, 2 of each kind (waves 0-3 even kind, 4-7 odd kind, ... 16 waves per CU)
    const double n_even = 2.0 * kIters * instr_even, n_odd = 2.0 * kIters * instr_odd;
    printf("%-28s %8.1f us   ns per instr (per SIMD): even-kind %.3f  odd-kind %.3f  [cycles at 2.4 GHz: %.1f / %.1f]\n", name, us,
           n_even ? us * 1e3 / n_even : 0.0, n_odd ? us * 1e3 / n_odd : 0.0, n_even ? us * 2.4e3 / n_even : 0.0, n_odd ? us * 2.4e3 / n_odd : 0.0);
}

int main() {
    float* out; hipMalloc(&out, 4096);
    unsigned short* h = (unsigned short*)malloc(4096 * 16);
    unsigned st = 12345u;
    for (int i = 0; i < 4096 * 8; ++i) {   // random sign, exponent near 1, random mantissa
        st = st * 1664525u + 1013904223u;
        h[i] = (unsigned short)(((st >> 16) & 0x80ffu) | (0x3f00u + ((st >> 9) & 0x0100u)));
    }
    bf16x8* rnd; hipMalloc(&rnd, 4096 * 16); hipMemcpy(rnd, h, 4096 * 16, hipMemcpyHostToDevice);
    run<8, 8>("mfma random operands (all)", out, 16, 16, rnd);
    run<8, 5>("mfma random operands | idle", out, 16, 0, rnd);
    run<0, 0>("exp | exp", out, 16, 16);
    run<1, 1>("fma | fma", out, 16, 16);
    run<2, 2>("pk_fma | pk_fma", out, 16, 16);
    run<6, 6>("max3 | max3", out, 16, 16);
    run<7, 7>("cvt_pk | cvt_pk", out, 16, 16);
    run<3, 3>("mfma | mfma", out, 16, 16);
    run<3, 5>("mfma | idle", out, 16, 0);
    run<0, 5>("exp | idle", out, 16, 0);
    run<1, 5>("fma | idle", out, 16, 0);
    run<3, 0>("mfma | exp", out, 16, 16);
    run<3, 1>("mfma | fma", out, 16, 16);
    run<3, 2>("mfma | pk_fma", out, 16, 16);
    run<4, 4>("mfma+exp same wave (all)", out, 32, 32);
    run<4, 5>("mfma+exp same wave | idle", out, 32, 0);
    run<0, 1>("exp | fma", out, 16, 16);
    return 0;
}
