// Developer micro-benchmark (not part of the product): how fast can one CU pull L2-resident operand
// slices, (a) with LDS-DMA (global_load_lds_dwordx4), (b) with global_load_dwordx4 into registers
// followed by ds_write_b128, (c) registers only.  8 waves per workgroup, one workgroup per CU, every
// workgroup re-reads the same 1.5 MiB buffer (a weight panel shared by all tiles) `iters` times,
// `depth` 1-KiB pieces in flight per wave.
// build: hipcc --offload-arch=gfx950 -O3 -o fill_rates fill_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, char* lds) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds, 16, 0, 0);
}

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void k(const char* buf, size_t bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t pieces = bytes / 1024;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        for (size_t p = wave * DEPTH + (blockIdx.x % 7) * 64; p + DEPTH <= pieces; p += 8 * DEPTH) {
            if (MODE == 0) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) glds16(buf + (p + d) * 1024 + lane * 16, smem + (wave * DEPTH + d) * 1024);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                u32x4 v[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const u32x4*>(buf + (p + d) * 1024 + lane * 16);
                if (MODE == 1) {
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) *reinterpret_cast<u32x4*>(smem + (wave * DEPTH + d) * 1024 + lane * 16) = v[d];
                } else {
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) acc += v[d];
                }
            }
        }
    }
    if (MODE == 0 || MODE == 1) {
        __syncthreads();
        acc = *reinterpret_cast<u32x4*>(smem + threadIdx.x * 16);
    }
    if (acc.x == 0x12345678u) sink[threadIdx.x] = acc.y;
}

template <int MODE, int DEPTH>
void run(const char* name, const char* buf, size_t bytes, unsigned* sink) {
    const int iters = 40;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEPTH * 1024);
    hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(512), 8 * DEPTH * 1024, 0, buf, bytes, 2, sink);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(512), 8 * DEPTH * 1024, 0, buf, bytes, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = (double)bytes * iters;   // roughly (start offsets differ slightly)
    printf("%-44s %8.1f us   %6.1f GB/s per CU   %5.2f TB/s chip\n", name, ms * 1e3, per_cu / (ms * 1e-3) / 1e9,
           per_cu * 256 / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = 1536 * 1024;
    char* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    unsigned* sink; hipMalloc(&sink, 4096);
    run<0, 2>("LDS-DMA, 2 pieces in flight per wave", buf, bytes, sink);
    run<0, 4>("LDS-DMA, 4 pieces in flight per wave", buf, bytes, sink);
    run<0, 8>("LDS-DMA, 8 pieces in flight per wave", buf, bytes, sink);
    run<0, 16>("LDS-DMA, 16 pieces in flight per wave", buf, bytes, sink);
    run<1, 2>("global_load + ds_write, 2 per wave", buf, bytes, sink);
    run<1, 4>("global_load + ds_write, 4 per wave", buf, bytes, sink);
    run<1, 8>("global_load + ds_write, 8 per wave", buf, bytes, sink);
    run<2, 4>("global_load to registers only, 4 per wave", buf, bytes, sink);
    run<2, 8>("global_load to registers only, 8 per wave", buf, bytes, sink);
    run<2, 16>("global_load to registers only, 16 per wave", buf, bytes, sink);
    return 0;
}
