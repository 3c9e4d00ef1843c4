// Third reproducer attempt: the failing build's instruction sequence of one gradient row, register numbers included (profiles/
// r05_head_store_hazard.txt), next to matrix + transcendental work of the SIMD's other wavefront.  hipcc --offload-arch=gfx950 -O2 ... && run
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(512) probe(unsigned long long* bad, unsigned* info, const int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
        const float r = 1.0f + (it & 255) * 0.00390625f + lane * 0.015625f;      // exact binary fractions: every product below is exact
        const float inv_se = 0.5f, coefn = -0.25f, gm1 = 3.0f, gm2 = 5.0f, pik = 0.75f, gm0 = 7.0f, coef = 0.25f;
        float o40, o41, o42, o43;
        asm volatile(
            "v_mov_b32 v114, %4\n\tv_mov_b32 v115, %4\n\tv_mov_b32 v56, %5\n\tv_mov_b32 v57, %5\n\t"      // (r, r) x (inv_se, inv_se)
            "v_mov_b32 v130, %6\n\tv_mov_b32 v131, %6\n\t"                                                // (-coef, -coef)
            "v_mov_b32 v156, %7\n\tv_mov_b32 v157, %8\n\tv_mov_b32 v110, %9\n\tv_mov_b32 v111, %10\n\t"     // (gm1, gm2), (coef, gm0)
            "v_mov_b32 v42, %11\n\tv_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v43, 0\n\t"
            "s_nop 7\n\t"
            "v_pk_mul_f32 v[40:41], v[114:115], v[56:57]\n\t"                      // wk pair
            "v_lshl_add_u64 v[78:79], v[38:39], 0, v[126:127]\n\t"
            "v_pk_mul_f32 v[108:109], v[130:131], v[40:41]\n\t"                    // gw pair = -coef wk
            "v_sub_f32 v42, v42, v41\n\t"                                          // pik - wk
            "v_mov_b32 v43, v109\n\t"
            "v_pk_mul_f32 v[40:41], v[110:111], v[42:43]\n\t"                      // (coef (pik - wk), gm0 gw)
            "v_pk_mul_f32 v[42:43], v[156:157], v[108:109] op_sel:[0,1]\n\t"       // (gm1 gw, gm2 gw): low lane reads the HIGH half of v[108:109]
            "s_nop 7\n\t"
            "v_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43"
            : "=v"(o40), "=v"(o41), "=v"(o42), "=v"(o43)
            : "v"(r), "v"(inv_se), "v"(coefn), "v"(gm1), "v"(gm2), "v"(coef), "v"(gm0), "v"(pik)
            : "v40", "v41", "v42", "v43", "v56", "v57", "v78", "v79", "v108", "v109", "v110", "v111", "v114", "v115", "v130", "v131", "v156", "v157");
        const float wk = r * inv_se, gw = coefn * wk;
        const float w40 = coef * (pik - wk), w41 = gm0 * gw, w42 = gm1 * gw, w43 = gm2 * gw;
        if (o40 != w40 || o41 != w41 || o42 != w42 || o43 != w43) {
            ++nbad; atomicOr(&info[lane >> 5], 1u);
            atomicOr(&info[2], (o40 != w40) | ((o41 != w41) << 1) | ((o42 != w42) << 2) | ((o43 != w43) << 3));
            if (o42 == 0.f) atomicOr(&info[3], 1u);
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    if (t == 12345.f) bad[1] = (unsigned long long)acc[1];
}
int main() {
    const int iters = 100000;
    unsigned long long* bad; unsigned* info;
    (void)hipMalloc(&bad, 16); (void)hipMalloc(&info, 16); (void)hipMemset(bad, 0, 16); (void)hipMemset(info, 0, 16);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, bad, info, iters);
    unsigned long long h[2]; unsigned l[4];
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); (void)hipMemcpy(l, info, 16, hipMemcpyDeviceToHost);
    printf("the failing build's row sequence: wrong %llu of %.3g (lanes 0..31: %s, 32..63: %s, components %x, a +-0 low result: %s)\n", h[0],
           256.0 * 512 * iters, l[0] ? "yes" : "no", l[1] ? "yes" : "no", l[2], l[3] ? "yes" : "no");
    return 0;
}
