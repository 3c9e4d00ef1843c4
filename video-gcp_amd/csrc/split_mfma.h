// Split-f16 arithmetic shared by the decoder kernels (conv3x3_split.hip, conv3x3_head_split.hip).
//
// gfx950 has no tf32 / xf32 matrix instruction and runs v_mfma_f32_16x16x4_f32 at 1/16 of the f16 / bf16 MFMA rate.  The exact f32
// head (conv3x3.hip: conv3x3_head_kernel) is bound by that rate (0.76 of the f32 MFMA peak).  This kernel computes the same 3x3 conv
// (DecoderModule.decode_seq -> gen_head, /root/reference/gcp/prediction/models/tree/tree_dense_rec.py:42) with both operands
// split into two f16 pieces and three f16 MFMAs per f32 product:
//
//     x 2^Ex = x1 + x2 + dx,  x1 = rn16(x 2^Ex),  x2 = rn16(x 2^Ex - x1)
//     w 2^Ew = w1 + w2 + dw,  likewise (one Ew per tensor; split on the device, gcpx_split_pack)
//     x w 2^(Ex+Ew) ~= x1 w1 + x1 w2 + x2 w1                                     (x2 w2 <= 2^-22 |x w| is dropped)
//
// Every partial product is exact in the f32 accumulator.  The bound is NORM-WISE per scaled unit, not element-wise: Ex is chosen PER
// ITEM from the largest staged activation (a power of two: scaling and unscaling are exact; max |x| 2^Ex lands in [2^14, 2^15), so
// nothing overflows), and
//
//     |dx| 2^-Ex <= max( 2^-22 |x| , 2^-39 max|x|_item )
//
// — two round-to-nearest f16 pieces carry 11 + 11 significant bits (2^-22 |x| worst case, about 2^-24 |x| on average) as long as the
// second piece is a normal f16, i.e. for |x| >= 2^-17 of the item's maximum; below that the f16 subnormal quantum (2^-24 under the
// scale) is the floor, and a value 2^-k of the maximum keeps about 39 - k bits.  A result is therefore one f32 rounding of its largest
// terms — f32-equivalent wherever the terms near the item's maximum carry the output (everything behind a BatchNorm) — and NOT
// element-wise f32 when a large channel meets zero weights beside a small channel that carries the output
// (tests/test_gpu_kernels.py::test_split_bound_head_large_channel_with_zero_weights states the bound and reports that case).  The error
// against a float64 conv is measured next to the exact-f32 kernel's in tests/test_gpu_kernels.py.
//
//
// This header: the f16 MFMA wrapper, hardware exp / log forms, the LDS layout of the wave-autonomous head / 16-channel kernels and
// their MFMA pass over channel tiles.
#pragma once
#include "split_common.h"

namespace {

// exp / log straight on the hardware instructions (v_exp_f32 / v_log_f32 are base 2).  __expf / __logf wrap them in a range fix-up for
// denormal results / inputs — two compares, two selects and a multiply per call: 531 selects and 558 compares in the likelihood variant of
// the head kernel, whose time IS its VALU count (PMC: VALU 58 % busy, matrix pipe 36 %, no overlap).  Every use here has a result that
// may flush to zero (softmax / mixture weights, sigmoids) or an argument >= 1e-12 (logs).
__device__ __forceinline__ float exp_hw(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float log_hw(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }

__device__ __forceinline__ float fast_tanh_s(float x) {
    const float e = exp_hw(2.f * x);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

struct SplitHeadCfg {
    static constexpr int CT = 7, KS = 5;
    static constexpr int RW = 18, RH = 6;
    static constexpr int PLANE_BYTES = RH * RW * 32;                       // one f16 piece of the region: 16 channels x 2 B per pixel
    static constexpr int REGION_BYTES = 2 * PLANE_BYTES;                   // 6912 B per wavefront
    static constexpr int W_BYTES = KS * CT * 2 * 1024;                     // 71680 B
    static constexpr int BIAS_BYTES = CT * 16 * 4;                          // bias of the 112 channel slots
    static constexpr int LDS_BYTES = W_BYTES + 8 * REGION_BYTES + BIAS_BYTES;
    static constexpr int STASH_BYTES = 16 * 64 * 4;                        // fused likelihood: 16 floats per lane parked during pass A
    static constexpr int LDS_BYTES_NLL = LDS_BYTES + 8 * STASH_BYTES;     // 160192 of the 163840 B of a CU
    static constexpr int NS = (RH * RW * 4 + 63) / 64;                     // float4 staging slots per lane
};

// One pass of the MFMA phase over channel tiles C0 .. C0 + NC - 1: KS x NC blocks (k-step s, tile) of 12 MFMAs.  The two weight
// pieces of the next block and, during the last four tiles of a k-step, the activation pieces of k-step s + 1 are fetched from LDS
// while a block computes (12 x 16 cycles of the matrix pipe cover the LDS round trip): a wavefront alone keeps the pipe busy.
// wl: [KS][CTW][2][64] x 16 B packed pieces (CTW = channel tiles of the pack); INIT: the accumulators start from zero.
template <int C0, int NC, int CTW = SplitHeadCfg::CT, bool INIT = true>
__device__ __forceinline__ void mfma_tiles(const char* wl, const char* reg, const int (&tapoff)[SplitHeadCfg::KS], const int lane,
                                           f32x4 (&acc)[NC][4]) {
    using Cfg = SplitHeadCfg;
    constexpr int KS = Cfg::KS, CT = CTW, RW = Cfg::RW;
    constexpr int PB = NC < 4 ? NC : 4;                      // blocks of a k-step that carry the next k-step's activation loads
    constexpr int PPB = 4 / PB;                              // pixel groups fetched per such block
    h8 wq[2][2], bq[2][4][2];
    auto load_w = [&](const int s, const int ct, h8 (&w)[2]) __attribute__((always_inline)) {
        const char* wp = wl + ((s * CT + ct) * 2 * 64 + lane) * 16;                      // [KS][CT][2][64] x 16 B
        w[0] = *reinterpret_cast<const h8*>(wp);
        w[1] = *reinterpret_cast<const h8*>(wp + 1024);
    };
    auto load_b = [&](const int s, const int pt, h8 (&b)[2]) __attribute__((always_inline)) {
        b[0] = *reinterpret_cast<const h8*>(reg + tapoff[s] + pt * RW * 32);
        b[1] = *reinterpret_cast<const h8*>(reg + tapoff[s] + pt * RW * 32 + Cfg::PLANE_BYTES);
    };
    load_w(0, C0, wq[0]);
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) load_b(0, pt, bq[0][pt]);
    __builtin_amdgcn_s_setprio(1);
    static_for<0, KS * NC>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value, s = t / NC, c = t % NC;
        constexpr bool more = t + 1 < KS * NC;
        constexpr bool pre_b = c >= NC - PB && s + 1 < KS;
        if constexpr (more) load_w((t + 1) / NC, C0 + (t + 1) % NC, wq[(t + 1) & 1]);
        if constexpr (pre_b) {
#pragma unroll
            for (int k = 0; k < PPB; ++k) load_b(s + 1, (c - (NC - PB)) * PPB + k, bq[(s + 1) & 1][(c - (NC - PB)) * PPB + k]);
        }
        const h8 w1 = wq[t & 1][0], w2 = wq[t & 1][1];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            // small terms first: they are added to the accumulator while it is still small
            if constexpr (s == 0 && INIT) acc[c][pt] = mfma32h(w2, bq[0][pt][0], f32x4{0, 0, 0, 0});
            else acc[c][pt] = mfma32h(w2, bq[s & 1][pt][0], acc[c][pt]);
        }
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) acc[c][pt] = mfma32h(w1, bq[s & 1][pt][1], acc[c][pt]);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) acc[c][pt] = mfma32h(w1, bq[s & 1][pt][0], acc[c][pt]);
        __builtin_amdgcn_sched_group_barrier(0x100, (more ? 2 : 0) + (pre_b ? 2 * PPB : 0), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    });
    __builtin_amdgcn_s_setprio(0);
}

// v_max_f32 without the canonicalising v_max the compiler puts in front of fmaxf on values that went through a lane swap (bit casts)
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// the value of the other lane of a pixel's lane pair (lane ^ 32) combined with this lane's: one swap instead of a ds_bpermute round trip
__device__ __forceinline__ float pair_sum(float v) {
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}
__device__ __forceinline__ float pair_max(float v) {
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax(__uint_as_float(s[0]), __uint_as_float(s[1]));
}
// largest value of a wavefront (v >= 0 in every lane, all lanes active): non-negative floats order like their bit patterns
__device__ __forceinline__ float wave_max_nonneg(float v) {
    int x = __float_as_int(v);
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false));     // row_half_mirror
    x = max(x, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, false));     // row_mirror: every lane of a 16-lane row holds the row's max
    const auto s16 = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
    x = max((int)s16[0], (int)s16[1]);
    const auto s32 = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    x = max((int)s32[0], (int)s32[1]);
    return __int_as_float(__builtin_amdgcn_readfirstlane(x));
}
// sum over the 16 lanes of a row by DPP row operations (every lane of the row ends with the sum)
__device__ __forceinline__ float row16_sum_dpp(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));     // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));     // row_mirror
    return v;
}


}  // namespace
