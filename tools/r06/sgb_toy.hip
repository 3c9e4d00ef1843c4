#include <hip/hip_runtime.h>
#include <type_traits>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) { if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); } }
#ifndef NU
#define NU 36
#endif
#ifndef NV
#define NV 500
#endif
__global__ void k(const float4* W, float* out, const float* in) {
    extern __shared__ float4 smem[];
    for (int i = threadIdx.x; i < 4096; i += 64) smem[i] = W[i];
    __syncthreads();
    const char* wl = (const char*)smem + threadIdx.x * 16;
    const char* rl = (const char*)smem + 40000 + threadIdx.x * 16;
    f32x16 acc[4];
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + 64 * i];
    h8 w[2][2], b[2][2];
    auto load_w = [&](int u, h8 (&d)[2]) { d[0] = *(const h8*)(wl + u * 2048); d[1] = *(const h8*)(wl + u * 2048 + 1024); };
    auto load_b = [&](int t, h8 (&d)[2]) { d[0] = *(const h8*)(rl + t * 592); d[1] = *(const h8*)(rl + t * 592 + 3552); };
    load_b(0, b[0]); load_w(0, w[0]);
    static_for<0, NU>([&](auto i) { constexpr int u = decltype(i)::value, t = u / 4, c = u % 4;
        if constexpr (u + 1 < NU) load_w(u + 1, w[(u + 1) & 1]);
        if constexpr (c == 0 && u + 4 < NU) load_b(t + 1, b[(t + 1) & 1]);
        const h8 w1 = w[u & 1][0], w2 = w[u & 1][1], b1 = b[t & 1][0], b2 = b[t & 1][1];
        if constexpr (t == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, b1, f32x16{}, 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, b1, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b2, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b1, acc[c], 0, 0, 0);
    });
    static_for<0, NV>([&](auto i) { constexpr int u = decltype(i)::value;
        if constexpr (u > 150 && u % 3 == 0) v[u & 7] = __builtin_amdgcn_exp2f(v[u & 7]); else
        v[u & 7] = __builtin_fmaf(v[u & 7], 0.999f, 0.25f); });
    static_for<0, NU * 3>([&](auto i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#ifdef DSG
        if constexpr (decltype(i)::value % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#endif
#ifdef COMB
        __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
#else
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
#endif
    });
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[threadIdx.x] = s;
}
