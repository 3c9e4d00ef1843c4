// Helpers shared by the split-f16 weight-gradient kernels (wgrad_conv_split.hip, wgrad_rows_split.hip): both operands are data, staged
// row-major ("pixel-major") into f16 planes in LDS and read back through the transposing LDS read, so that the MFMA k index walks rows.
#pragma once
#include "split_common.h"

namespace {

typedef const f32x4 __attribute__((address_space(1)))* gptr4;

// explicitly global (a flat load would tie up both memory counters), uniform base + 32-bit lane offset (one address register per slot)
__device__ __forceinline__ float4 gload4(const char* base, const unsigned off_bytes) {
    const f32x4 t = *(gptr4)(base + off_bytes);
    return make_float4(t[0], t[1], t[2], t[3]);
}

// ds_read_b64_tr_b16 (measured with tools/tr_probe.hip): inside a 16-lane group, lane s supplies the address of an 8-byte chunk
// (4 halfs) and lane i receives half (i & 3) of the chunks supplied by lanes 4 j + (i >> 2), j = 0 .. 3.  With lane s pointing at
// [pixel s >> 2][channels 4 (s & 3) .. + 3] of a [pixel][16 channel] plane, lane i gets channel i of four pixels: the MFMA operand
// (k = pixels) straight from the pixel-major image.
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ h4 ds_tr(const _Float16* p) {
    return __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)p));
}
__device__ __forceinline__ h8 ds_tr8(const _Float16* p, const int second_off) {
    const h4 lo = ds_tr(p), hi = ds_tr(p + second_off);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int CTRL>
__device__ __forceinline__ float dppf(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

// largest value over the wavefront: DPP inside the 16-lane rows, v_readlane across them (wave-uniform result)
__device__ __forceinline__ float wave_max_nonneg(float v) {
    v = fmaxf(v, dppf<0xB1>(v));          // quad_perm [1, 0, 3, 2]
    v = fmaxf(v, dppf<0x4E>(v));          // quad_perm [2, 3, 0, 1]
    v = fmaxf(v, dppf<0x141>(v));         // row_half_mirror
    v = fmaxf(v, dppf<0x140>(v));         // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

}  // namespace
