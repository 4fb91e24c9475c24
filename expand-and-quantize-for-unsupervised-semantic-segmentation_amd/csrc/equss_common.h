// Shared helpers for the gfx950 kernels: error plumbing for the C ABI, vector types, bf16
// conversion and wave-level reductions (wave = 64 lanes on CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/equss_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));   // 8 bf16 = one MFMA A/B fragment
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define EQUSS_WAVE 64

namespace equss {

// Per-thread message buffer behind equss_last_error().
char* error_buffer();
int fail(int code, const char* fmt, ...);

// Event-pair recording around launches (see equss_profile_enable in the public header).
bool prof_on();
void prof_begin(int cls, hipStream_t s);
void prof_end(hipStream_t s);
struct ProfScope {
    hipStream_t s; bool on;
    ProfScope(int cls, hipStream_t st) : s(st), on(prof_on()) { if (on) prof_begin(cls, s); }
    ~ProfScope() { if (on) prof_end(s); }
};

// Raise a kernel's dynamic-LDS limit above 64 KiB.  hipFuncSetAttribute acts on the current device, so
// the opt-in is remembered per (kernel, device) and issued once for each pair.
bool lds_opt_in(const void* kernel, int bytes);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fail(0, "%s: %s", what, hipGetErrorString(e));
        return -(int)e;
    }
    return EQUSS_OK;
}

// fp32 -> bf16, round to nearest even; hipcc lowers the cast to v_cvt_pk_bf16_f32 on gfx950,
// which keeps NaN a NaN.
__device__ __forceinline__ u16 f2bf(float x) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(u16, b);
}
__device__ __forceinline__ float bf2f(u16 b) {
    return __builtin_bit_cast(float, (uint32_t)b << 16);
}
// two fp32 -> one dword of two bf16 (lo in bits 0..15): a single v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// GELU (erf form, torch.nn.GELU's default) with one transcendental:
//   gelu(x) = x * Phi(x) = max(x, 0) - |x| * (erfc(|x| / sqrt 2) / 2),   erfc(|x| / sqrt 2) / 2 = 2^Q(|x|),
// Q a degree-5 polynomial fitted to log2 of that tail (smooth and nearly quadratic, unlike erf itself),
// weighted by the sensitivity |x| 2^Q of the result.  |error| <= 7e-7 absolute over all x (fp32
// evaluation included), far below the bf16 rounding of the stored result; 5 fma + v_exp + max + fma
// instead of the rcp + exp + 14 ops of Abramowitz-Stegun 7.1.26 or the ~30 branchy ops of erff.
// Q's leading coefficient is negative, so large |x| gives 2^Q = 0 without a clamp (x = +-inf gives NaN).
// Two values at a time: the packed fp32 fma issues once for both (|x| = 2 max(x, 0) - x is exact and
// saves the two sign-bit masks; there is no packed max or exp).
typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f32x2 gelu_erf2(gelu_f32x2 x) {
    const gelu_f32x2 relu = {fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
    const gelu_f32x2 a = __builtin_elementwise_fma(gelu_f32x2{2.f, 2.f}, relu, -x);
    gelu_f32x2 q = __builtin_elementwise_fma(gelu_f32x2{-0.00047328611253760755f, -0.00047328611253760755f}, a,
                                             gelu_f32x2{0.007084404584020376f, 0.007084404584020376f});
    q = __builtin_elementwise_fma(q, a, gelu_f32x2{-0.051827020943164825f, -0.051827020943164825f});
    q = __builtin_elementwise_fma(q, a, gelu_f32x2{-0.45999282598495483f, -0.45999282598495483f});
    q = __builtin_elementwise_fma(q, a, gelu_f32x2{-1.1507877111434937f, -1.1507877111434937f});
    q = __builtin_elementwise_fma(q, a, gelu_f32x2{-1.000037670135498f, -1.000037670135498f});
    const gelu_f32x2 e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    return __builtin_elementwise_fma(-a, e, relu);
}

// The same for four pairs at a time with the steps of the four evaluations interleaved (same operations per element, so
// the same bits): a lone evaluation is a chain of dependent packed fmas, each waiting for the one before.
__device__ __forceinline__ void gelu_erf2x4(gelu_f32x2 (&x)[4]) {
    gelu_f32x2 relu[4], a[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) relu[i] = gelu_f32x2{fmaxf(x[i].x, 0.f), fmaxf(x[i].y, 0.f)};
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = __builtin_elementwise_fma(gelu_f32x2{2.f, 2.f}, relu[i], -x[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        q[i] = __builtin_elementwise_fma(gelu_f32x2{-0.00047328611253760755f, -0.00047328611253760755f}, a[i],
                                         gelu_f32x2{0.007084404584020376f, 0.007084404584020376f});
    __builtin_amdgcn_sched_barrier(0);   // (the steps stay apart: the scheduler would put each chain back together)
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = __builtin_elementwise_fma(q[i], a[i], gelu_f32x2{-0.051827020943164825f, -0.051827020943164825f});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = __builtin_elementwise_fma(q[i], a[i], gelu_f32x2{-0.45999282598495483f, -0.45999282598495483f});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = __builtin_elementwise_fma(q[i], a[i], gelu_f32x2{-1.1507877111434937f, -1.1507877111434937f});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = __builtin_elementwise_fma(q[i], a[i], gelu_f32x2{-1.000037670135498f, -1.000037670135498f});
    __builtin_amdgcn_sched_barrier(0);
    gelu_f32x2 e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = gelu_f32x2{__builtin_amdgcn_exp2f(q[i].x), __builtin_amdgcn_exp2f(q[i].y)};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] = __builtin_elementwise_fma(-a[i], e[i], relu[i]);
    }
}

// LDS stores the compiler's wait-count pass does not see.  After a global_load_lds_* the compiler puts
// s_waitcnt vmcnt(0) in front of every LDS store it emits, because the DMA's destination might alias it; in the
// ring kernels that drains every stage in flight and every global store the wave has issued, at each epilogue.
// Callers use these only for rows that are no DMA target while stages are in flight (staging rows of a
// consumed slot, the hidden-activation tile, the row-statistics exchange).  LDS operations of one wave
// execute in order, so a later load of the same wave sees the data; other waves need lgkmcnt(0) + a barrier.
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void lds_store_b64(void* p, uint32_t lo, uint32_t hi) {
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t v = {lo, hi};
    asm volatile("ds_write_b64 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}
// 16-byte LDS load the wait-count pass does not see (the first LDS load it emits itself after a global_load_lds_*
// gets the same s_waitcnt vmcnt(0)).  The data is in the returned registers only after lds_wait(registers).
typedef uint32_t lds_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lds_u32x4 lds_load_b128(const void* p) {
    lds_u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_addr(p)) : "memory");
    return v;
}
// (the registers pass through the wait as in/out operands: arithmetic on them cannot be scheduled above it)
template <int N>
__device__ __forceinline__ void lds_wait(lds_u32x4 (&d)[N]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(d[i]));
}
// The same load at a 32-bit LDS address plus an immediate byte offset (< 65536), and the counted wait that goes with
// it: "all but the N youngest LDS operations of this wave have returned" (they return in order), after which d holds data.
template <int OFF>
__device__ __forceinline__ lds_u32x4 lds_load_b128_at(uint32_t addr) {
    lds_u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// (the transposing 8-byte load of gfx950: lane l of a 16-lane group gets column l of the 4 x 16 block of 16-bit elements
// whose four rows the group's lanes address, four lanes per row)
typedef uint32_t lds_u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ lds_u32x2 lds_load_tr16_b64_at(uint32_t addr) {
    lds_u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lds_tie(lds_u32x4& d) { asm volatile("" : "+v"(d)); }
__device__ __forceinline__ void lds_tie(lds_u32x2& d) { asm volatile("" : "+v"(d)); }
template <int N, typename... R>
__device__ __forceinline__ void lds_wait_for(R&... d) {
    __builtin_amdgcn_sched_barrier(0);   // (what was written before the wait stays before it: MFMAs would sink below it)
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    (lds_tie(d), ...);
}
__device__ __forceinline__ void lds_store_f32x2(void* p, float a, float b) {
    lds_store_b64(p, __builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b));
}

template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {  // sum over WIDTH consecutive lanes
#pragma unroll
    for (int off = WIDTH / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int off = WIDTH / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

}  // namespace equss
