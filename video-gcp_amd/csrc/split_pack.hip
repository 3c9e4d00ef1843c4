// Re-split of every split-f16 weight tensor after an optimizer step in ONE launch: workgroup p does what gcpx_split_pack
// (conv3x3_split.hip) does for tensor p — largest gathered magnitude -> power-of-two exponent -> the two f16 pieces in
// [..][2][512]-element blocks.  Eleven one-workgroup launches in a row took 0.45 ms of every training step (each is bound by what one
// CU pulls); side by side they take as long as the largest.
#include "common.h"

namespace {
__global__ void __launch_bounds__(1024) split_pack_group_kernel(const gcpx_split_pack_desc* __restrict__ tab) {
    __shared__ float red[16];
    const gcpx_split_pack_desc d = tab[blockIdx.x];
    const float* __restrict__ theta = d.src;
    const int* __restrict__ idx = d.idx;
    _Float16* __restrict__ out = reinterpret_cast<_Float16*>(d.out);
    const int n = d.n, tid = threadIdx.x;
    // (eight gathers in flight per thread: one workgroup owns a tensor of up to 10^5 elements, and a dependent index -> value load per
    //  loop iteration made the launch 0.2 ms)
    constexpr int U = 8;
    float m = 0.f;
    for (int i0 = tid; i0 < n; i0 += 1024 * U) {
        int k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) k[u] = i0 + 1024 * u < n ? idx[i0 + 1024 * u] : -1;
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = k[u] >= 0 ? theta[k[u]] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) m = fmaxf(m, fabsf(v[u]));
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    // largest magnitude times 2^e in [2^14, 2^15): the range the kernels' exponent bookkeeping covers (as gcpx_split_pack)
    int e = m > 0.f ? 14 + 127 - (int)((__float_as_uint(m) >> 23) & 0xff) : 0;
    e = max(-20, min(100, e));
    if (tid == 0) *d.log2_out = e;
    const float sc = __uint_as_float((unsigned)(127 + e) << 23);
    for (int i0 = tid; i0 < n; i0 += 1024 * U) {
        int k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) k[u] = i0 + 1024 * u < n ? idx[i0 + 1024 * u] : -1;
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = k[u] >= 0 ? theta[k[u]] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 1024 * u;
            if (i < n) {
                const float v = w[u] * sc;
                const _Float16 h1 = (_Float16)v;
                const _Float16 h2 = (_Float16)(v - (float)h1);
                const int o = (i >> 9) * 1024 + (i & 511);
                out[o] = h1;
                out[o + 512] = h2;
            }
        }
    }
}
// The same in two launches with SPLIT_NB workgroups per tensor (the one-workgroup form takes as long as the largest tensor: 0.14 ms for
// the 295 k weights of a 256 -> 128 block, behind every optimizer step): largest magnitudes first (atomic max of the float bits, which
// order like the non-negative floats they encode), then the pieces.  Same exponent and pieces, bit for bit.
constexpr int SPLIT_NB = 32;
__device__ __forceinline__ void split_chunk(const int n, int& lo, int& hi) {
    const int per = ((n / 512 + SPLIT_NB - 1) / SPLIT_NB) * 512;
    lo = min(n, (int)blockIdx.y * per);
    hi = min(n, lo + per);
}

__global__ void __launch_bounds__(256) split_pack_max_kernel(const gcpx_split_pack_desc* __restrict__ tab, unsigned* __restrict__ maxbits) {
    __shared__ float red[4];
    const gcpx_split_pack_desc d = tab[blockIdx.x];
    int lo, hi;
    split_chunk(d.n, lo, hi);
    if (lo >= hi) return;
    const int tid = threadIdx.x;
    constexpr int U = 8;
    float m = 0.f;
    for (int i0 = lo + tid; i0 < hi; i0 += 256 * U) {
        int k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) k[u] = i0 + 256 * u < hi ? d.idx[i0 + 256 * u] : -1;
#pragma unroll
        for (int u = 0; u < U; ++u) m = fmaxf(m, k[u] >= 0 ? fabsf(d.src[k[u]]) : 0.f);
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) atomicMax(maxbits + blockIdx.x, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

__global__ void __launch_bounds__(256) split_pack_write_kernel(const gcpx_split_pack_desc* __restrict__ tab, const unsigned* __restrict__ maxbits) {
    const gcpx_split_pack_desc d = tab[blockIdx.x];
    int lo, hi;
    split_chunk(d.n, lo, hi);
    const float m = __uint_as_float(maxbits[blockIdx.x]);
    int e = m > 0.f ? 14 + 127 - (int)((__float_as_uint(m) >> 23) & 0xff) : 0;
    e = max(-20, min(100, e));
    const int tid = threadIdx.x;
    if (blockIdx.y == 0 && tid == 0) *d.log2_out = e;
    const float sc = __uint_as_float((unsigned)(127 + e) << 23);
    _Float16* __restrict__ out = reinterpret_cast<_Float16*>(d.out);
    constexpr int U = 8;
    for (int i0 = lo + tid; i0 < hi; i0 += 256 * U) {
        int k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) k[u] = i0 + 256 * u < hi ? d.idx[i0 + 256 * u] : -1;
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = k[u] >= 0 ? d.src[k[u]] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 256 * u;
            if (i < hi) {
                const float v = w[u] * sc;
                const _Float16 h1 = (_Float16)v;
                const _Float16 h2 = (_Float16)(v - (float)h1);
                const int o = (i >> 9) * 1024 + (i & 511);
                out[o] = h1;
                out[o + 512] = h2;
            }
        }
    }
}
}  // namespace

// scratch: DEVICE [nprob] uint32 (contents irrelevant, overwritten)
extern "C" int gcpx_split_pack_group2(const gcpx_split_pack_desc* tab, int32_t nprob, uint32_t* scratch, void* stream_) {
    GCPX_CHECK_ARG(tab && scratch && nprob > 0, "bad arguments");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (hipMemsetAsync(scratch, 0, sizeof(uint32_t) * nprob, stream) != hipSuccess) {
        gcpx_set_error("%s: hipMemsetAsync failed", __func__);
        return GCPX_ERR_HIP;
    }
    hipLaunchKernelGGL(split_pack_max_kernel, dim3(nprob, SPLIT_NB), dim3(256), 0, stream, tab, scratch);
    hipLaunchKernelGGL(split_pack_write_kernel, dim3(nprob, SPLIT_NB), dim3(256), 0, stream, tab, scratch);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

// tab: DEVICE array of nprob descriptors (n a positive multiple of 512 each)
extern "C" int gcpx_split_pack_group(const gcpx_split_pack_desc* tab, int32_t nprob, void* stream_) {
    GCPX_CHECK_ARG(tab && nprob > 0, "bad arguments");
    hipLaunchKernelGGL(split_pack_group_kernel, dim3(nprob), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream_), tab);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
