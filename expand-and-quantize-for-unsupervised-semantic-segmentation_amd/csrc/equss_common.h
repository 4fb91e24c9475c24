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
