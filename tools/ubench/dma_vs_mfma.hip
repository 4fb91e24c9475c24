// Developer micro-benchmark (not part of the product): do LDS-DMA transfers and MFMAs of DIFFERENT waves of a CU
// overlap?  One workgroup per CU: 8 "compute" waves (two per SIMD) run back-to-back MFMAs [+ ds_read_b128 fragment
// reads], NP "producer" waves (one per SIMD) only request global_load_lds_dwordx4 pieces of an L2-resident buffer
// into a private LDS region, with a bounded number in flight.  No barriers: each group runs free, the kernel
// reports the time each group took and the launch takes max of the two.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_vs_mfma dma_vs_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* gsrc, char* lds) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds, 16, 0, 0);
}

// MODE bit 0: compute waves run, bit 1: producer waves run, bit 2: compute waves also read NR fragments per 24 MFMAs
template <int MODE, int NP>
__global__ __launch_bounds__(768, 1) void k(const char* buf, int iters_c, int iters_p, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t0 = wall_clock64();
    if (wave < 8) {
        if (!(MODE & 1)) return;
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 f[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) f[i][j] = (short)(0x3f80 + ((lane * 7 + i * 13 + j * 29) & 0x7f));
        for (int it = 0; it < iters_c; ++it) {
            if (MODE & 4) {
#pragma unroll
                for (int i = 0; i < 6; ++i) f[i] = *reinterpret_cast<const bf16x8*>(smem + 65536 + ((i * 1024 + lane * 16 + wave * 2048 + it * 512) & 32767));
            }
#pragma unroll
            for (int m = 0; m < 24; ++m)
                acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[m % 6], f[(m + 1) % 6], acc[m & 7], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        if (s == 123.456f) out[100 + threadIdx.x] = 1;
        if (lane == 0 && blockIdx.x == 7 && wave == 0) out[0] = wall_clock64() - t0;
    } else {
        if (!(MODE & 2) || wave >= 8 + NP) return;
        const int pw = wave - 8;
        // each iteration: 12 pieces of 1 KiB (12 KiB per producer wave), at most 24 pieces in flight per wave
        for (int it = 0; it < iters_p; ++it) {
            const char* src = buf + ((size_t)((it * NP + pw) * 12) * 1024) % (1536 * 1024 - 12 * 1024) + lane * 16;
            char* dst = smem + pw * 16384 + (it & 1) * 0;   // 12 KiB private region (data are never read)
#pragma unroll
            for (int p = 0; p < 12; ++p) glds16(src + p * 1024, dst + p * 1024);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && blockIdx.x == 7 && pw == 0) out[1] = wall_clock64() - t0;
    }
}

template <int MODE, int NP>
void run(const char* buf, unsigned long long* out, const char* what) {
    const int iters_c = 4000, iters_p = 4000;
    const int lds = 65536 + 32768;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipMemset(out, 0, 64);
    hipLaunchKernelGGL((k<MODE, NP>), dim3(256), dim3(768), lds, 0, buf, 100, 100, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NP>), dim3(256), dim3(768), lds, 0, buf, iters_c, iters_p, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    const double tc = h[0] * 0.01, tp = h[1] * 0.01;   // us
    printf("%-58s launch %7.1f us", what, ms * 1e3);
    if (MODE & 1) printf(" | compute waves %7.1f us = %5.1f ns per 24 MFMAs (alone at 2.4 GHz: 320)", tc, tc * 1e3 / iters_c);
    if (MODE & 2) printf(" | producers %7.1f us = %5.1f GB/s per CU", tp, (double)iters_p * NP * 12 * 1024 / (tp * 1e3));
    printf("\n");
}

int main() {
    char* buf; hipMalloc(&buf, 1536 * 1024); hipMemset(buf, 0x3c, 1536 * 1024);
    unsigned long long* out; hipMalloc(&out, 1 << 16);
    run<1, 4>(buf, out, "MFMAs only");
    run<5, 4>(buf, out, "MFMAs + 6 fragment reads per 24");
    run<2, 4>(buf, out, "4 producer waves only");
    run<2, 2>(buf, out, "2 producer waves only");
    run<3, 4>(buf, out, "MFMAs and 4 producer waves");
    run<3, 2>(buf, out, "MFMAs and 2 producer waves");
    run<7, 4>(buf, out, "MFMAs + fragment reads and 4 producer waves");
    run<7, 2>(buf, out, "MFMAs + fragment reads and 2 producer waves");
    return 0;
}
