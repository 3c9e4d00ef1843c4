// Microbenchmark (gfx950): do MFMAs of one wavefront overlap with VALU work of the OTHER wavefront of the same SIMD, and with VALU work of
// the same wavefront?  Background: PMC of conv3x3_head_split_kernel shows the matrix pipe 36 % busy and the VALU 58 % with their sum ~ 94 %.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap tools/mfma_valu_overlap.hip && ./mfma_valu_overlap
// One 512-thread workgroup per CU (LDS-limited), waves 0-3 = first wavefront of SIMD 0-3, waves 4-7 = second.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

typedef float f16v __attribute__((ext_vector_type(16)));
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0)
#define MF32(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)

template <int TRANS>
__device__ __forceinline__ void valu8(float (&v)[8], float c) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (TRANS) v[k] = __builtin_amdgcn_exp2f(v[k]) * c;
        else v[k] = __builtin_fmaf(v[k], c, 0.25f);
    }
}

// mode bits: what the first / second wavefront of every SIMD runs.  kind: 0 nothing, 1 MFMA stream (4 independent accumulators),
// 2 VALU stream (8 independent fma chains), 3 interleaved: 1 MFMA + NV VALU, 4 transcendental stream
template <int NV>
__global__ void __launch_bounds__(512) k(float* out, int kindA, int kindB, int iters) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6;
    const int kind = wave < 4 ? kindA : kindB;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.01f * (threadIdx.x + i);
    const float c = 0.999f;
    if (kind == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { MF(acc[0], a, b); MF(acc[1], a, b); MF(acc[2], a, b); MF(acc[3], a, b); }
        }
    } else if (kind == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2 * NV; ++u) valu8<0>(v, c);       // 16 NV VALU per iteration = NV per MFMA of kind 1
        }
    } else if (kind == 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2 * NV; ++u) valu8<1>(v, c);
        }
    } else if (kind == 5) {                                     // the same FLOPs per iteration in 32x32x16 MFMAs: 8 instead of 16
        f16v big[2];
        for (int i = 0; i < 16; ++i) { big[0][i] = 0.f; big[1][i] = 0.f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { MF32(big[0], a, b); MF32(big[1], a, b); }
        }
        for (int i = 0; i < 16; ++i) v[i & 7] += big[0][i] + big[1][i];
    } else if (kind == 6) {                                     // interleaved: 1 MFMA 32x32x16 + 2 NV VALU
        f16v big[2];
        for (int i = 0; i < 16; ++i) { big[0][i] = 0.f; big[1][i] = 0.f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                MF32(big[u & 1], a, b);
#pragma unroll
                for (int w = 0; w < 2 * NV; ++w) v[(u * 2 * NV + w) & 7] = __builtin_fmaf(v[(u * 2 * NV + w) & 7], c, 0.25f);
            }
        }
        for (int i = 0; i < 16; ++i) v[i & 7] += big[0][i] + big[1][i];
    } else if (kind == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                MF(acc[u & 3], a, b);
#pragma unroll
                for (int w = 0; w < NV; ++w) v[(u * NV + w) & 7] = __builtin_fmaf(v[(u * NV + w) & 7], c, 0.25f);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s + lds[threadIdx.x];
}

template <int NV>
static float run(int kindA, int kindB, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    k<NV><<<256, 512, 100 * 1024>>>(out, kindA, kindB, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<NV><<<256, 512, 100 * 1024>>>(out, kindA, kindB, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

template <int NV>
static void sweep(float* out, int iters) {
    const char* names[] = {"-", "mfma", "valu", "mfma+valu interleaved", "trans", "mfma32", "mfma32+valu interleaved"};
    const int cases[][2] = {{1, 0}, {0, 2}, {1, 2}, {2, 1}, {1, 1}, {2, 2}, {3, 0}, {3, 3}, {0, 4}, {1, 4}, {4, 4},
                            {5, 0}, {5, 5}, {5, 2}, {2, 5}, {6, 0}, {6, 6}, {5, 4}};
    printf("---- %d VALU per MFMA (per wavefront and iteration: 16 MFMAs / %d VALU) ----\n", NV, 16 * NV);
    for (auto& cs : cases) {
        const float ms = run<NV>(cs[0], cs[1], iters, out);
        // cycles per iteration at 2.4 GHz nominal (the clock under load is lower; compare rows, not absolute numbers)
        printf("first wave: %-24s second wave: %-24s  %8.3f ms  = %7.1f ns / iteration\n", names[cs[0]], names[cs[1]], ms, 1e6 * ms / iters);
    }
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const int iters = 20000;
    sweep<1>(out, iters);
    sweep<3>(out, iters);
    sweep<6>(out, iters);
    return 0;
}
