// Second reproducer attempt for the round-4 corruption of the training head's gradient rows: the instruction neighbourhood of the failing
// build — packed-f32 multiplies (one with op_sel) feeding a 16-byte global store back to back, and the store's data registers overwritten
// by the next packed multiplies two instructions later — with matrix / transcendental work on the SIMD's other wavefront.  Every stored
// value is read back and compared.  Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/pk_store_hazard.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr int RING = 32;

template <int MODE>   // 0: pk -> store back to back + overwrite after; 1: scalar multiplies instead of pk; 2: pk, no overwrite after the store
__global__ void __launch_bounds__(512) probe(float4* out, unsigned long long* bad, unsigned* info, const int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t tid = (size_t)blockIdx.x * 512 + threadIdx.x;
    float4* ring = out + tid * RING;
    f32x2 x, y = {0.5f + wave, 3.0f}, z = {7.0f, -1.25f}, g = {0.75f, 1.5f};
    f32x4 acc = {0, 0, 0, 0};
    h8 ha, hb;
    for (int k = 0; k < 8; ++k) { ha[k] = (_Float16)(lane * 0.001f); hb[k] = (_Float16)1.0f; }
    float t = 0.3f + lane;
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        if (wave >= 4) {
#pragma unroll
            for (int k = 0; k < 6; ++k) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0); t = __builtin_amdgcn_exp2f(t * 0.5f) + acc[0] * 1e-30f; }
        }
        float4* p = ring + (it % RING);
        x.x = 1.0f + lane * 0.015625f + (it & 1023) * 0.0009765625f; x.y = 2.0f - lane * 0.03125f - (it & 1023) * 0.0009765625f;
        // a = x * y; b = z * (g.hi, g.hi) [op_sel]; store {a, b}; then (MODE != 2) the data registers are overwritten as in the kernel
        if (MODE == 1)
            asm volatile("v_mul_f32 v40, %5, %6\n\tv_mul_f32 v41, %7, %8\n\tv_mul_f32 v42, %9, %11\n\tv_mul_f32 v43, %10, %11\n\t"
                         "global_store_dwordx4 %0, v[40:43], off\n\t"
                         "v_pk_mul_f32 v[40:41], %3, v[42:43] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[42:43], %1, v[40:41]"
                         :: "v"(p), "v"(x), "v"(y), "v"(z), "v"(g), "v"(x.x), "v"(y.x), "v"(x.y), "v"(y.y), "v"(z.x), "v"(z.y), "v"(g.y)
                         : "v40", "v41", "v42", "v43", "memory");
        else if (MODE == 0)
            asm volatile("v_pk_mul_f32 v[40:41], %1, %2\n\tv_pk_mul_f32 v[42:43], %3, %4 op_sel:[0,1]\n\t"
                         "global_store_dwordx4 %0, v[40:43], off\n\t"
                         "v_pk_mul_f32 v[44:45], %3, v[42:43] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 v46, 0\n\t"
                         "v_pk_mul_f32 v[40:41], %3, v[42:43] op_sel_hi:[0,1]\n\tv_pk_mul_f32 v[42:43], %1, v[40:41]"
                         :: "v"(p), "v"(x), "v"(y), "v"(z), "v"(g) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "memory");
        else
            asm volatile("v_pk_mul_f32 v[40:41], %1, %2\n\tv_pk_mul_f32 v[42:43], %3, %4 op_sel:[0,1]\n\t"
                         "global_store_dwordx4 %0, v[40:43], off"
                         :: "v"(p), "v"(x), "v"(y), "v"(z), "v"(g) : "v40", "v41", "v42", "v43", "memory");
        if ((it % RING) == RING - 1) {            // read the ring back
            __builtin_amdgcn_s_waitcnt(0);
            for (int k = 0; k < RING; ++k) {
                const int itk = it - (RING - 1) + k;
                const f32x2 xx = {1.0f + lane * 0.015625f + (itk & 1023) * 0.0009765625f, 2.0f - lane * 0.03125f - (itk & 1023) * 0.0009765625f};
                const float4 v = ring[k];
                const float w0 = xx.x * y.x, w1 = xx.y * y.y, w2 = z.x * g.y, w3 = z.y * g.y;
                if (v.x != w0 || v.y != w1 || v.z != w2 || v.w != w3) {
                    ++nbad; atomicOr(&info[lane >> 5], 1u);
                    atomicOr(&info[2], (v.x != w0) | ((v.y != w1) << 1) | ((v.z != w2) << 2) | ((v.w != w3) << 3));
                    if (v.x == 0.f || v.y == 0.f || v.z == 0.f || v.w == 0.f) atomicOr(&info[3], 1u);
                }
            }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    if (t == 12345.f) bad[1] = (unsigned long long)acc[1];
}

template <int MODE>
void run(const char* name) {
    const int iters = 20000;
    float4* out; unsigned long long* bad; unsigned* info;
    (void)hipMalloc(&out, (size_t)256 * 512 * RING * 16); (void)hipMalloc(&bad, 16); (void)hipMalloc(&info, 16);
    (void)hipMemset(bad, 0, 16); (void)hipMemset(info, 0, 16);
    hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(512), 0, 0, out, bad, info, iters);
    unsigned long long h[2]; unsigned l[4];
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(l, info, 16, hipMemcpyDeviceToHost);
    printf("%-64s wrong stores %llu of %.3g  (lanes 0..31: %s, 32..63: %s, components %x, a zero seen: %s)\n", name, h[0], 256.0 * 512 * iters,
           l[0] ? "yes" : "no", l[1] ? "yes" : "no", l[2], l[3] ? "yes" : "no");
    (void)hipFree(out); (void)hipFree(bad); (void)hipFree(info);
}
int main() {
    run<0>("pk_mul, pk_mul op_sel -> store -> pk_mul over the data regs");
    run<1>("v_mul x 4 -> store -> pk_mul over the data regs");
    run<2>("pk_mul, pk_mul op_sel -> store, nothing behind it");
    return 0;
}
