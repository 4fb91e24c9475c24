// Developer micro-benchmark (not part of the product): the LDS-DMA ring step of ring_step.hip with the
// two wave groups of a workgroup (waves 0-3 and 4-7: one wave of each per SIMD) run one barrier apart,
// so that one group's ds_read phase coincides with the other group's MFMA phase ("ping-pong").
// Every phase is [NR ds_read_b128] barrier [NM MFMAs] barrier; a step is P phases on one stage.
// build: hipcc --offload-arch=gfx950 -O3 -o pingpong pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* gsrc, char* lds) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds, 16, 0, 0);
}

// MODE 0: lock step (all waves read, then all waves MFMA, one barrier per step, as ring_step.hip)
// MODE 1: ping-pong, two barriers per phase, group 1 one barrier behind group 0
// MODE 2: as MODE 0 but the same two barriers per phase (cost of the barriers alone)
template <int MODE, int PW, int P, int NM, int NR>
__global__ __launch_bounds__(512, 2) void k(const char* buf, int steps, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kStage = 8 * PW * 1024, S = 3;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2;
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
    issue();
    issue();
    int cbuf = 0;
    if (MODE == 1 && grp == 1) __builtin_amdgcn_s_barrier();
    for (int st = 0; st < steps; ++st) {
        const char* base = smem + cbuf * kStage;
        cbuf = cbuf + 1 == S ? 0 : cbuf + 1;
        if (MODE == 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue();
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            bf16x8 f[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i)
                f[i] = *reinterpret_cast<const bf16x8*>(base + (((p * NR + i) * 1024 + lane * 16 + wave * 2048) % kStage));
            if (MODE != 0) {
                if (p == 1 || P == 1) issue();       // two barriers after the last read of the slot it refills
                if (p == P - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");   // next step's stage
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
            }
#pragma unroll
            for (int m = 0; m < NM; ++m)
                acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[m % NR], f[(m + 1) % NR], acc[m & 7], 0, 0, 0);
            if (MODE != 0) {
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            asm volatile("" ::: "memory");
        }
    }
    if (MODE == 1 && grp == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE, int PW, int P, int NM, int NR>
void run(const char* buf, float* out) {
    const int steps = 4000;
    const int lds = 3 * 8 * PW * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, PW, P, NM, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<MODE, PW, P, NM, NR>), dim3(256), dim3(512), lds, 0, buf, 100, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, PW, P, NM, NR>), dim3(256), dim3(512), lds, 0, buf, steps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / steps;
    const double mf = P * NM * 2 * 16 / 2.4;
    printf("mode %d, stage %2d KiB, %d phases x (%2d MFMA + %2d ds_read) per wave-step: %7.1f ns per step  (MFMA alone %6.1f ns: %4.1f %%; %5.1f GB/s per CU)\n",
           MODE, 8 * PW, P, NM, NR, ns, mf, 100.0 * mf / ns, 8.0 * PW * 1024 / ns);
}

int main() {
    char* buf; hipMalloc(&buf, 1536 * 1024); hipMemset(buf, 0x3c, 1536 * 1024);
    float* out; hipMalloc(&out, 4096);
    // G2-like: W2 stage 48 KiB, two k32 sub-steps of 24 MFMAs + 10 fragment reads
    run<0, 6, 2, 24, 10>(buf, out);
    run<2, 6, 2, 24, 10>(buf, out);
    run<1, 6, 2, 24, 10>(buf, out);
    run<1, 3, 1, 24, 10>(buf, out);
    // G1-like: 48 KiB stage, three k32 sub-steps of 8 MFMAs + 6 reads (today), regrouped as 16 + 12
    run<0, 6, 3, 8, 6>(buf, out);
    run<1, 6, 3, 8, 6>(buf, out);
    run<0, 4, 1, 16, 12>(buf, out);
    run<1, 4, 1, 16, 12>(buf, out);
    run<1, 4, 2, 16, 12>(buf, out);
    run<1, 6, 3, 16, 12>(buf, out);
    // fat phases
    run<0, 4, 1, 32, 12>(buf, out);
    run<1, 4, 1, 32, 12>(buf, out);
    run<1, 4, 2, 32, 12>(buf, out);
    run<1, 2, 1, 32, 12>(buf, out);
    return 0;
}
