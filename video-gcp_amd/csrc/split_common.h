// What every split-f16 kernel shares (split_mfma.h: the decoder convs; split_tr.h: the weight-gradient kernels): the f16 vector types,
// the 16x16x32 f16 MFMA wrapper, the two-piece split of four scaled values, the |.|-maximum instruction and a compile-time loop.
// ONE copy: the inline-asm forms must not drift apart between the forward and the backward kernels.
#pragma once
#include "common.h"

#include <type_traits>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// D[16x16] += A[16x32] * B[32x16]: lane l holds A[i = l & 15][k = 8 (l >> 4) .. + 7], B[k = 8 (l >> 4) .. + 7][j = l & 15];
// D as in mfma16 (lane l, reg r: i = 4 (l >> 4) + r, j = l & 15).
__device__ __forceinline__ f32x4 mfma32h(h8 a, h8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// the two f16 pieces of four scaled values: p1 = rn16(v s), p2 = rn16(v s - p1) (v s is exact: s is a power of two), one fused
// multiply-add with an f16 result per piece and value (the same bits as the cvt / fma / cvt chain, 8 instructions instead of ~14)
__device__ __forceinline__ void split4(const float4 v, const float s, h4& p1, h4& p2) {
    unsigned a0, a1, b0, b1;
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(a0), "=&v"(b0) : "v"(v.x), "v"(v.y), "s"(s));
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(a1), "=&v"(b1) : "v"(v.z), "v"(v.w), "s"(s));
    const uint2 ua = make_uint2(a0, a1), ub = make_uint2(b0, b1);
    p1 = *reinterpret_cast<const h4*>(&ua);
    p2 = *reinterpret_cast<const h4*>(&ub);
}

// running maximum of |x|, |y| in one instruction
__device__ __forceinline__ float vmax3abs(float a, float x, float y) {
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(x), "v"(y));
    return r;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

}  // namespace
