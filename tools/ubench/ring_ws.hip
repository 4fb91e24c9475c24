// Developer micro-benchmark (not part of the product): the ring step of ring_real.hip in producer / consumer form.
// One workgroup per CU = 8 consumer waves (fragment reads + MFMAs) + 4 producer waves (LDS-DMA requests only), one
// raw s_barrier per stage for all 12 waves, S ring slots.  Sources: half of every stage from a tile of the CU's own
// (96 KiB, re-read all the time: L2 resident), half from a 2.25 MiB weight matrix shared by every CU.
// STAGE KiB per stage (multiple of 4), NM MFMAs and NR ds_read_b128 per consumer wave and stage.
// build: hipcc --offload-arch=gfx950 -O3 -o ring_ws ring_ws.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* gsrc, char* lds) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int STAGE, int S, int NM, int NR, int WIDE>
__global__ __launch_bounds__(768, 1) void k(const char* shared, const char* own, int steps, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kStage = STAGE * 1024, PW = STAGE / 4;   // pieces per producer wave and stage
    static_assert((S - 2) * PW <= 63, "vmcnt range");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= 8) {
        const int pw = wave - 8;
        const char* mine = own + (size_t)blockIdx.x * 98304;
        int issued = 0, ibuf = 0;
        auto issue = [&]() {
            if (issued >= steps) return;
#pragma unroll
            for (int p = 0; p < PW; ++p) {
                const int piece = p * 4 + pw;
                const bool second = piece >= STAGE / 2;
                const int q = piece % (STAGE / 2);
                const int kstep = issued & 3, chunk = (issued >> 2) % 12;
                int row, col;
                if (WIDE == 0) { row = (q & 7) * 16 + (lane >> 2); col = (kstep * 192 + (q >> 3) * 64 + (lane & 3) * 16) % 768; }
                else if (WIDE == 1) { row = (q * 8 + (lane >> 3)) & 127; col = ((kstep & 1) * 128 + (lane & 7) * 16 + (kstep >> 1) * 256 + (q >> 4) * 384) % 768; }
                else {   // WIDE = bytes of a row segment (256, 512) or 1024 = wholly contiguous piece
                    constexpr int LPS = WIDE / 16;           // lanes per segment
                    row = (q * (64 / LPS) + lane / LPS) & 127;
                    col = ((kstep % (768 / WIDE > 0 ? 768 / WIDE : 1)) * WIDE + (lane % LPS) * 16) % 768;
                    if (WIDE == 1024) { row = 0; col = ((q + kstep * 16) & 63) * 1024 + lane * 16; }
                }
                const char* base = second ? shared + 98304 + (size_t)chunk * 98304 : mine;
                glds16(base + row * 768 + col, smem + ibuf * kStage + piece * 1024);
            }
            ++issued;
            ibuf = ibuf + 1 == S ? 0 : ibuf + 1;
        };
        auto wait_stage = [&](int j) {
            const int younger = issued - 1 - j;
            if (younger >= S - 2) wait_vm<(S - 2) * PW>();
            else if (younger == 1) wait_vm<PW>();
            else wait_vm<0>();
        };
#pragma unroll
        for (int d = 0; d < S - 1; ++d) issue();
        wait_stage(0);
        __builtin_amdgcn_s_barrier();
        for (int st = 0; st < steps; ++st) {
            asm volatile("" ::: "memory");
            issue();
            if (st + 1 < steps) wait_stage(st + 1);
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cbuf = 0;
    __builtin_amdgcn_s_barrier();
    for (int st = 0; st < steps; ++st) {
        const char* base = smem + cbuf * kStage;
        cbuf = cbuf + 1 == S ? 0 : cbuf + 1;
        constexpr int NF = NR > 0 ? NR : 2;
        bf16x8 f[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            if (NR > 0) f[i] = *reinterpret_cast<const bf16x8*>(base + ((i * 1024 + lane * 16 + wave * 2048) % kStage));
            else
#pragma unroll
                for (int j = 0; j < 8; ++j) f[i][j] = (short)(0x3f80 + ((lane * 7 + i * 13 + j * 29 + st) & 0x7f));
        }
#pragma unroll
        for (int m = 0; m < NM; ++m)
            acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[m % NF], f[(m + 1) % NF], acc[m & 7], 0, 0, 0);
        if (NM == 0) {
#pragma unroll
            for (int i = 0; i < NF; ++i) asm volatile("" ::"v"(f[i]));
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int STAGE, int S, int NM, int NR, int WIDE>
void run(const char* shared, const char* own, float* out) {
    const int steps = 4000;
    const int lds = S * STAGE * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<STAGE, S, NM, NR, WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<STAGE, S, NM, NR, WIDE>), dim3(256), dim3(768), lds, 0, shared, own, 100, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<STAGE, S, NM, NR, WIDE>), dim3(256), dim3(768), lds, 0, shared, own, steps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / steps;
    printf("stage %2d KiB (%s pieces), %d slots, %2d MFMA + %2d ds_read per consumer wave: %7.1f ns per stage (MFMA alone %5.1f ns at 2.4 GHz), %5.1f GB/s per CU\n",
           STAGE, WIDE == 0 ? "64-B" : WIDE == 1 ? "128-B" : WIDE == 256 ? "256-B" : WIDE == 512 ? "512-B" : "1-KiB", S, NM, NR, ns, NM * 2 * 16 / 2.4, STAGE * 1024.0 / ns);
}

int main() {
    char *shared, *own; hipMalloc(&shared, 4 << 20); hipMalloc(&own, 256 * 98304);
    std::vector<unsigned short> h((256 * 98304) / 2);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 16) & 0x807f) | (0x3f00 + ((x >> 8) & 0x80)) | ((x >> 4) & 0x7f)); }
    hipMemcpy(own, h.data(), 256 * 98304, hipMemcpyHostToDevice);
    hipMemcpy(shared, h.data() + 777, 4 << 20, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, 4096);
    run<32, 4, 24, 10, 0>(shared, own, out);    // the qkv / projection step
    run<32, 4, 0, 10, 0>(shared, own, out);     //   without MFMAs
    run<32, 4, 24, 0, 0>(shared, own, out);     //   without fragment reads
    run<48, 3, 24, 18, 0>(shared, own, out);    // G1 step of the fused kernel
    run<48, 3, 48, 20, 1>(shared, own, out);    // G2 step
    run<36, 3, 24, 30, 0>(shared, own, out);    // G1 step with 64-wide hidden chunks: 3 x (8 MFMAs + 10 reads)
    run<24, 4, 24, 10, 1>(shared, own, out);    // G2 step, 32 wide
    run<24, 5, 24, 10, 1>(shared, own, out);
    run<32, 4, 24, 10, 1>(shared, own, out);
    run<32, 4, 24, 10, 256>(shared, own, out);
    run<32, 4, 24, 10, 512>(shared, own, out);
    run<32, 4, 24, 10, 1024>(shared, own, out);
    run<32, 4, 0, 0, 1024>(shared, own, out);
    run<32, 4, 0, 0, 0>(shared, own, out);      // which of the two slows the requests down: the 64-B pieces ...
    run<32, 4, 0, 0, 1>(shared, own, out);
    run<32, 4, 0, 10, 1024>(shared, own, out);  // ... or the consumers' LDS reads
    run<32, 4, 0, 10, 1>(shared, own, out);
    run<32, 4, 24, 0, 1024>(shared, own, out);  // ... or their MFMAs
    run<32, 4, 12, 8, 1024>(shared, own, out);  // 64-row tiles: half the MFMAs per stage
    run<32, 4, 12, 12, 1024>(shared, own, out);
    run<32, 4, 16, 10, 1024>(shared, own, out);
    run<24, 5, 12, 8, 1024>(shared, own, out);
    run<48, 3, 24, 16, 1024>(shared, own, out);
    return 0;
}
