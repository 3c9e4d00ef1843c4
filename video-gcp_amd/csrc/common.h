// Shared device helpers for the gfx950 kernels (wave = 64 lanes, f32-input MFMA 16x16x4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gcpx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// D[16x16] += A[16x4] * B[4x16], exact f32 (v_mfma_f32_16x16x4_f32, 32-cycle issue).
//   A operand: lane l holds A[i = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][j = l & 15]
//   D: lane l, reg r holds D[i = (l >> 4) * 4 + r][j = l & 15]
// All kernels here put OUTPUT CHANNELS on the A/i side and PIXELS/ROWS on the B/j side, so a lane ends up
// with 4 consecutive output channels of one pixel/row -> one 16-byte store, and per-row epilogues
// (LSTM gates, GroupNorm groups, mixture parameters) stay inside a lane or a 4-lane column.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }

__device__ __forceinline__ float apply_act(float x, int act) {
    if (act == GCPX_ACT_LRELU) return lrelu(x, 0.2f);
    if (act == GCPX_ACT_TANH) return tanhf(x);
    return x;
}

__device__ __forceinline__ float4 affine_act4(float4 v, const float* __restrict__ scale,
                                              const float* __restrict__ shift, int c, int act) {
    if (scale) {
        const float4 s = *reinterpret_cast<const float4*>(scale + c);
        const float4 t = *reinterpret_cast<const float4*>(shift + c);
        v.x = fmaf(v.x, s.x, t.x); v.y = fmaf(v.y, s.y, t.y); v.z = fmaf(v.z, s.z, t.z); v.w = fmaf(v.w, s.w, t.w);
    }
    if (act == GCPX_ACT_LRELU) {
        v.x = lrelu(v.x, 0.2f); v.y = lrelu(v.y, 0.2f); v.z = lrelu(v.z, 0.2f); v.w = lrelu(v.w, 0.2f);
    }
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// LSTM cell backward of (row r, unit u): the body of gcpx_lstm_bwd, also run by the row GEMM's epilogue (gcpx_gemm_args.lstm_bwd) with
// `dh_above` = the accumulator instead of a.dh_dense
__device__ __forceinline__ void lstm_bwd_cell(const gcpx_lstm_bwd_args& a, const int r, const int u, const float dh_above) {
    const int b = r / a.rpb, j = r % a.rpb;
    const size_t pos = (size_t)b * a.pb + (size_t)j * a.prow + u;
    const float4 g = *reinterpret_cast<const float4*>(a.gates + ((size_t)r * a.H + u) * 4);   // i, f, g, o (activated)
    const float c = a.c_new[pos];
    const float cp = a.c_prev[(size_t)r * a.c_prev_stride + u];
    float dh = 0.f, dc = 0.f;
    dh += dh_above;
    if (a.dh_pos) dh += a.dh_pos[pos];
    if (a.dc_pos) dc += a.dc_pos[pos];
    const float tc = tanhf(c);
    dc += dh * g.w * (1.f - tc * tc);
    const float di = dc * g.z * g.x * (1.f - g.x);
    const float df = dc * cp * g.y * (1.f - g.y);
    const float dg = dc * g.x * (1.f - g.z * g.z);
    const float dout = dh * tc * g.w * (1.f - g.w);
    float* dgr = a.dgates + (size_t)r * 4 * a.H + u;
    dgr[0] = di;
    dgr[a.H] = df;
    dgr[2 * a.H] = dg;
    dgr[3 * a.H] = dout;
    a.dc_prev[(size_t)r * a.dcp_stride + u] = dc * g.y;
}

// sum over the 16 lanes that share (lane >> 4): butterflies inside a 16-lane row
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

// slot of log_scale_{c,k} in the 112-slot layout of the discrete-logistic-mixture head (video-gcp_amd/packing.py:dlm_log_scale_slot;
// c = 0: red, inside mixture k's 8-slot group; c = 1, 2: green / blue, slots 80..99 arranged for the head kernel's fused likelihood)
__host__ __device__ __forceinline__ constexpr int dlm_ls_slot(const int c, const int k) {
    return c == 0 ? 8 * k + 7
                  : (k >= 8 ? 96 + 2 * (k - 8) + (c - 1) : 80 + 4 * (2 * (k & 1) + (k >> 2)) + 2 * ((k >> 1) & 1) + (c - 1));
}

void gcpx_set_error(const char* fmt, ...);
#define GCPX_CHECK_ARG(cond, msg)                         \
    do {                                                  \
        if (!(cond)) {                                    \
            gcpx_set_error("%s: %s", __func__, msg);      \
            return GCPX_ERR_INVALID_ARG;                  \
        }                                                 \
    } while (0)
#define GCPX_CHECK_LAUNCH()                                                        \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            gcpx_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return GCPX_ERR_HIP;                                                   \
        }                                                                          \
    } while (0)
