// Developer micro-benchmark (not part of the product): cost of one step of the LDS-DMA ring protocol
// used by the GEMM kernels -- [s_waitcnt vmcnt(n); s_barrier; request the next stage; MFMAs] -- as a
// function of stages in flight, stage size, MFMAs per step and LDS fragment reads per step.
// 8 waves per workgroup, one workgroup per CU, operands from a 1.5 MiB L2-resident buffer.
// build: hipcc --offload-arch=gfx950 -O3 -o ring_step ring_step.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* gsrc, char* lds) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds, 16, 0, 0);
}

// INFLIGHT stages in flight (ring of INFLIGHT + 1 slots), PW pieces (1 KiB) per wave per stage,
// NM MFMAs per wave per step, NR ds_read_b128 per wave per step
template <int INFLIGHT, int PW, int NM, int NR>
__global__ __launch_bounds__(512, 2) void k(const char* buf, int steps, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kStage = 8 * PW * 1024, S = INFLIGHT + 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int issued = 0, ibuf = 0;
    auto issue = [&]() {
        const char* src = buf + ((size_t)(issued * 8 + wave) * PW * 1024) % (1536 * 1024 - PW * 1024) + lane * 16;
#pragma unroll
        for (int p = 0; p < PW; ++p) glds16(src + p * 1024, smem + ibuf * kStage + (wave * PW + p) * 1024);
        ++issued;
        ibuf = ibuf + 1 == S ? 0 : ibuf + 1;
    };
#pragma unroll
    for (int d = 0; d < INFLIGHT; ++d) issue();
    int cbuf = 0;
    for (int st = 0; st < steps; ++st) {
        if (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue();
        const char* base = smem + cbuf * kStage;
        cbuf = cbuf + 1 == S ? 0 : cbuf + 1;
        bf16x8 f[NR > 0 ? NR : 1];
#pragma unroll
        for (int i = 0; i < NR; ++i) f[i] = *reinterpret_cast<const bf16x8*>(base + ((i * 1024 + lane * 16 + wave * 2048) % kStage));
        if (NR == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[0][j] = (short)(0x3f80 + lane + j);
        }
#pragma unroll
        for (int m = 0; m < NM; ++m)
            acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[m % (NR > 0 ? NR : 1)], f[(m + 1) % (NR > 0 ? NR : 1)], acc[m & 7], 0, 0, 0);
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int INFLIGHT, int PW, int NM, int NR>
void run(const char* buf, float* out) {
    const int steps = 4000;
    const int lds = (INFLIGHT + 1) * 8 * PW * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<INFLIGHT, PW, NM, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<INFLIGHT, PW, NM, NR>), dim3(256), dim3(512), lds, 0, buf, 100, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<INFLIGHT, PW, NM, NR>), dim3(256), dim3(512), lds, 0, buf, steps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / steps;
    printf("in flight %d, stage %2d KiB, %2d MFMA + %2d ds_read per wave-step: %7.1f ns per step  (MFMA alone %5.1f ns at 2.4 GHz; %5.1f GB/s per CU)\n",
           INFLIGHT, 8 * PW, NM, NR, ns, NM * 2 * 16 / 2.4, 8.0 * PW * 1024 / ns);
}

int main() {
    char* buf; hipMalloc(&buf, 1536 * 1024); hipMemset(buf, 0x3c, 1536 * 1024);
    float* out; hipMalloc(&out, 4096);
    run<1, 4, 0, 0>(buf, out);
    run<1, 4, 16, 0>(buf, out);
    run<1, 4, 48, 0>(buf, out);
    run<1, 6, 48, 0>(buf, out);
    run<2, 4, 0, 0>(buf, out);
    run<2, 4, 16, 0>(buf, out);
    run<2, 4, 48, 0>(buf, out);
    run<3, 4, 16, 0>(buf, out);
    run<1, 4, 16, 12>(buf, out);
    run<2, 4, 16, 12>(buf, out);
    run<1, 6, 48, 20>(buf, out);
    run<2, 3, 24, 10>(buf, out);
    run<1, 8, 64, 24>(buf, out);
    run<2, 4, 32, 12>(buf, out);
    return 0;
}
