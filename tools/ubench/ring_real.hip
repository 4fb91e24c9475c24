// Developer micro-benchmark (not part of the product): the ring step of ring_step.hip (one 32 KiB stage in
// flight, 24 MFMAs + 18 ds_read_b128 per wave and step: the G1 step of the fused block-tail kernel) with
// source address patterns that approach the real kernel's step by step.
// build: hipcc --offload-arch=gfx950 -O3 -o ring_real ring_real.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* gsrc, char* lds) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds, 16, 0, 0);
}

// MODE 0: 1-KiB contiguous pieces of a 1.5 MiB buffer shared by every CU
// MODE 1: pieces of 16 rows x 64 B (row stride 768 B) of a shared 96 KiB tile
// MODE 2: pieces of 8 rows x 128 B (row stride 768 B) of a shared 96 KiB tile
// MODE 3: as 1, half of the pieces from a tile of the CU's own (96 KiB each, 24 MiB in all)
// MODE 4: as 3, the shared half walks a 2.25 MiB weight matrix (12 chunks of 128 rows x 768 B x 2 matrices)
// MODE 5: as 4 with 128-B pieces for both halves
template <int MODE, int INFLIGHT, int SPREAD>
__global__ __launch_bounds__(512, 2) void k(const char* shared, const char* own, int steps, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PW = 4, NM = 24, NR = 18;
    constexpr int kStage = 8 * PW * 1024, S = INFLIGHT + 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int issued = 0, ibuf = 0;
    const char* mine = own + (size_t)blockIdx.x * 98304;
    auto issue_piece = [&](int p) {
        {
            const int piece = p * 8 + wave;     // 32 pieces per stage
            const char* src;
            if (MODE == 0) {
                src = shared + ((size_t)(issued * 32 + piece) * 1024) % (1536 * 1024 - 1024) + lane * 16;
            } else {
                // a stage covers 96 B.. (three 32-wide sub-tiles) of K: k offset advances by 192 B per step, 4 steps per chunk
                const int kstep = issued & 3, chunk = (issued >> 2) % 12;
                const bool second = piece >= 16;            // second half of the stage: the weight slice
                const int q = piece & 15;                   // 16 pieces per half
                const bool wide = MODE == 2 || MODE == 5;
                int row, col;
                if (!wide) { const int sub = q >> 3; row = (q & 7) * 16 + (lane >> 2); col = kstep * 192 + sub * 64 + (lane & 3) * 16; }
                else { row = q * 8 + (lane >> 3); col = (kstep & 1) * 128 + (lane & 7) * 16 + (kstep >> 1) * 256; }
                const char* base = shared;
                if (second) { if (MODE >= 4) base = shared + 98304 + (size_t)chunk * 98304; }
                else if (MODE >= 3) base = mine;
                src = base + row * 768 + col;
            }
            glds16(src, smem + ibuf * kStage + piece * 1024);
        }
    };
    auto issue_done = [&]() {
        ++issued;
        ibuf = ibuf + 1 == S ? 0 : ibuf + 1;
    };
    auto issue = [&]() {
#pragma unroll
        for (int p = 0; p < PW; ++p) issue_piece(p);
        issue_done();
    };
#pragma unroll
    for (int d = 0; d < INFLIGHT; ++d) issue();
    int cbuf = 0;
    for (int st = 0; st < steps; ++st) {
        if (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!SPREAD) issue();
        const char* base = smem + cbuf * kStage;
        cbuf = cbuf + 1 == S ? 0 : cbuf + 1;
        bf16x8 f[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) f[i] = *reinterpret_cast<const bf16x8*>(base + ((i * 1024 + lane * 16 + wave * 2048) % kStage));
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (SPREAD && m % (NM / PW) == 0) { issue_piece(m / (NM / PW)); __builtin_amdgcn_sched_barrier(0); }
            acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[m % NR], f[(m + 1) % NR], acc[m & 7], 0, 0, 0);
        }
        if (SPREAD) issue_done();
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE, int INFLIGHT, int SPREAD = 0>
void run(const char* shared, const char* own, float* out, const char* what) {
    const int steps = 4000;
    const int lds = (INFLIGHT + 1) * 32 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, INFLIGHT, SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<MODE, INFLIGHT, SPREAD>), dim3(256), dim3(512), lds, 0, shared, own, 100, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, INFLIGHT, SPREAD>), dim3(256), dim3(512), lds, 0, shared, own, steps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / steps;
    printf("%smode %d (%s), %d in flight: %7.1f ns per step (MFMA alone 320 ns), %5.1f GB/s per CU\n", SPREAD ? "spread issue, " : "", MODE, what, INFLIGHT, ns, 32768.0 / ns);
}

int main(int argc, char**) {
    char *shared, *own; hipMalloc(&shared, 4 << 20); hipMalloc(&own, 256 * 98304);
    if (argc > 1) {   // random bf16 values in (-2, 2) instead of a constant: the matrix pipe draws more power on them
        std::vector<unsigned short> h((256 * 98304) / 2);
        unsigned x = 12345u;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 16) & 0x807f) | (0x3f00 + ((x >> 8) & 0x80)) | ((x >> 4) & 0x7f)); }
        hipMemcpy(own, h.data(), 256 * 98304, hipMemcpyHostToDevice);
        hipMemcpy(shared, h.data() + 777, 4 << 20, hipMemcpyHostToDevice);
        printf("random operands\n");
    } else {
        hipMemset(shared, 0x3c, 4 << 20); hipMemset(own, 0x3c, 256 * 98304);
    }
    float* out; hipMalloc(&out, 4096);
    run<0, 1>(shared, own, out, "1-KiB contiguous pieces, shared");
    run<1, 1>(shared, own, out, "16 x 64-B pieces, shared tile");
    run<2, 1>(shared, own, out, "8 x 128-B pieces, shared tile");
    run<3, 1>(shared, own, out, "64-B pieces, half from the CU's own tile");
    run<4, 1>(shared, own, out, "64-B pieces, own tile + 2.25 MiB of weights");
    run<5, 1>(shared, own, out, "128-B pieces, own tile + 2.25 MiB of weights");
    run<0, 2>(shared, own, out, "1-KiB contiguous pieces, shared");
    run<3, 2>(shared, own, out, "64-B pieces, half from the CU's own tile");
    run<4, 2>(shared, own, out, "64-B pieces, own tile + 2.25 MiB of weights");
    run<5, 2>(shared, own, out, "128-B pieces, own tile + 2.25 MiB of weights");
    run<0, 1, 1>(shared, own, out, "1-KiB contiguous pieces, shared");
    run<0, 2, 1>(shared, own, out, "1-KiB contiguous pieces, shared");
    run<4, 1, 1>(shared, own, out, "64-B pieces, own tile + 2.25 MiB of weights");
    run<4, 2, 1>(shared, own, out, "64-B pieces, own tile + 2.25 MiB of weights");
    run<5, 2, 1>(shared, own, out, "128-B pieces, own tile + 2.25 MiB of weights");
    return 0;
}
