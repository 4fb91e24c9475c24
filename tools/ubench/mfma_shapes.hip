// Developer micro-benchmark (not part of the product): sustained bf16 MFMA rate with random operands,
// 16x16x32 against 32x32x16, at the register blocking of a 128 x 64 wave tile (128 accumulator
// registers), 2 waves per SIMD, with and without a workgroup barrier every 64 / 32 MFMAs.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shapes mfma_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef KITERS
#define KITERS 20000
#endif

template <int SHAPE, bool BARRIER>
__global__ __launch_bounds__(512, 2) void k(float* out, const bf16x8* rnd) {
    bf16x8 ra[8], rb[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) ra[i] = rnd[(threadIdx.x * 12 + i) & 4095];
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = rnd[(threadIdx.x * 12 + 8 + i) & 4095];
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < KITERS; ++it) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[i], ra[(j + ks) & 7], acc[i][j], 0, 0, 0);
            if (BARRIER) __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
    } else {
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < KITERS; ++it) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rb[(i + ks) & 3], ra[(j + 2 * ks) & 7], acc[i][j], 0, 0, 0);
            if (BARRIER) __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    }
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int SHAPE, bool BARRIER>
void run(const char* name, float* out, const bf16x8* rnd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, BARRIER>), dim3(256), dim3(512), 0, 0, out, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<SHAPE, BARRIER>), dim3(256), dim3(512), 0, 0, out, rnd);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 3;
    // per iteration and wave: 128 x 64 x 64 x 2 flop
    const double flop = 256.0 * 8 * KITERS * 128.0 * 64 * 64 * 2;
    printf("%-34s %9.1f us  %7.1f TFLOP/s\n", name, us, flop / us / 1e6);
}

int main() {
    float* out; hipMalloc(&out, 4096);
    unsigned short* h = (unsigned short*)malloc(4096 * 16);
    unsigned st = 12345u;
    for (int i = 0; i < 4096 * 8; ++i) {
        st = st * 1664525u + 1013904223u;
        h[i] = (unsigned short)(((st >> 16) & 0x80ffu) | (0x3f00u + ((st >> 9) & 0x0100u)));
    }
    bf16x8* rnd; hipMalloc(&rnd, 4096 * 16); hipMemcpy(rnd, h, 4096 * 16, hipMemcpyHostToDevice);
    run<16, false>("16x16x32, no barrier", out, rnd);
    run<32, false>("32x32x16, no barrier", out, rnd);
    run<16, true>("16x16x32, barrier per 64 MFMA", out, rnd);
    run<32, true>("32x32x16, barrier per 32 MFMA", out, rnd);
    run<16, false>("16x16x32, no barrier (again)", out, rnd);
    run<32, false>("32x32x16, no barrier (again)", out, rnd);
    return 0;
}
