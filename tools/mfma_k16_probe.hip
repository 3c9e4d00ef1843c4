// Microbenchmark (gfx950): what does the legacy K = 16 f16 MFMA (v_mfma_f32_16x16x16_f16) cost next to the K = 32 form
// (v_mfma_f32_16x16x32_f16) — half (then the output head's ninth tap, K = 16, can leave the padded tenth behind), or the same?
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_k16_probe.bin tools/mfma_k16_probe.hip && tools/mfma_k16_probe.bin
// One wavefront per SIMD (256-thread workgroups, one per CU), 8 independent accumulators, the MFMAs as asm statements (hipcc rotates the
// accumulators of a builtin loop through v_accvgpr moves), s_memtime (shader cycles) around the loop.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define M32(u) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a8), "v"(b8))
#define M16(u) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a4), "v"(b4))

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int iters) {
    h8 a8, b8;
    h4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.001f * (threadIdx.x + i)); b8[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { M32(0); M32(1); M32(2); M32(3); M32(4); M32(5); M32(6); M32(7); }
        else if (KIND == 1) { M16(0); M16(1); M16(2); M16(3); M16(4); M16(5); M16(6); M16(7); }
        else { M32(0); M32(1); M32(2); M32(3); M16(4); M32(5); M32(6); M32(7); M32(0); M16(1); }      // 4 x K = 32 + 1 x K = 16, twice
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float* out;
    long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4);
    (void)hipMalloc(&cyc, 8);
    const int iters = 20000;
    const char* names[3] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x16_f16", "head pattern: (4 x K=32 + 1 x K=16) x 2"};
    const int per_iter[3] = {8, 8, 10};
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            if (kind == 0) k<0><<<256, 256>>>(out, cyc, iters);
            else if (kind == 1) k<1><<<256, 256>>>(out, cyc, iters);
            else k<2><<<256, 256>>>(out, cyc, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            long long c;
            (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            if (rep == 1)
                printf("%-44s %7.3f ms, %6.2f shader cycles per MFMA per SIMD (one wavefront), %5.2f ns\n", names[kind], ms,
                       (double)c / ((double)per_iter[kind] * iters), ms * 1e6 / ((double)per_iter[kind] * iters));
        }
    }
    return 0;
}
