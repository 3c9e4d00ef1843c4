// Do the MFMAs of ONE wavefront overlap with the VALU work of the OTHER wavefront of the same SIMD on gfx950, and does s_setprio change it?
// (tools/r06/issue_probe: within one wavefront, 5 plain VALU issue for free under every v_mfma_f32_32x32x16_f16, 1 under a 16x16x32.)
// 512-thread workgroups, one per CU: waves 0-3 = first wavefront of SIMD 0-3 run the MFMA stream, waves 4-7 the VALU stream.
// Both streams are hand-ordered asm; each runs a fixed number of iterations, the kernel's time is the longer of the two.
#include <hip/hip_runtime.h>
#include <cstdio>

#define MF16 "v_mfma_f32_16x16x32_f16 a[0:3], v[64:67], v[68:71], a[0:3]\n v_mfma_f32_16x16x32_f16 a[4:7], v[64:67], v[68:71], a[4:7]\n" \
             "v_mfma_f32_16x16x32_f16 a[8:11], v[64:67], v[68:71], a[8:11]\n v_mfma_f32_16x16x32_f16 a[12:15], v[64:67], v[68:71], a[12:15]\n"
#define MF32 "v_mfma_f32_32x32x16_f16 a[0:15], v[64:67], v[68:71], a[0:15]\n v_mfma_f32_32x32x16_f16 a[16:31], v[64:67], v[68:71], a[16:31]\n"
#define FMA8 "v_fma_f32 v80, v80, v72, v73\n v_fma_f32 v81, v81, v72, v73\n v_fma_f32 v82, v82, v72, v73\n v_fma_f32 v83, v83, v72, v73\n" \
             "v_fma_f32 v84, v84, v72, v73\n v_fma_f32 v85, v85, v72, v73\n v_fma_f32 v86, v86, v72, v73\n v_fma_f32 v87, v87, v72, v73\n"
// a dependent chain with transcendentals, as an epilogue has them: 8 instructions, 2 of them v_exp_f32, every one reading the previous result
#define CHAIN8 "v_fma_f32 v80, v80, v72, v73\n v_exp_f32 v80, v80\n v_fma_f32 v80, v80, v72, v73\n v_add_f32 v80, v80, v73\n" \
               "v_mul_f32 v80, v80, v72\n v_exp_f32 v80, v80\n v_fma_f32 v80, v80, v72, v73\n v_min_f32 v80, v80, v73\n"
#ifdef USE_CHAIN
#undef FMA8
#define FMA8 CHAIN8
#endif

// mode: bit 0 = MFMA waves run, bit 1 = VALU waves run; form: 0 = 16x16x32 (8 per iteration), 1 = 32x32x16 (4 per iteration: same FLOPs)
template <int FORM, int PRIO_M, int PRIO_V>
__global__ void __launch_bounds__(512) k(float* out, int mode, int it_m, int it_v) {
    const int wave = threadIdx.x >> 6;
    asm volatile("v_mov_b32 v72, 0x3f7fbe77\n v_mov_b32 v73, 0x3e800000\n v_mov_b32 v64, 0x2c002c00\n v_mov_b32 v65, 0x2c002c00\n"
                 "v_mov_b32 v66, 0x2c002c00\n v_mov_b32 v67, 0x2c002c00\n v_mov_b32 v68, 0x2c002c00\n v_mov_b32 v69, 0x2c002c00\n"
                 "v_mov_b32 v70, 0x2c002c00\n v_mov_b32 v71, 0x2c002c00\n v_mov_b32 v80, 0\n v_mov_b32 v81, 0\n v_mov_b32 v82, 0\n v_mov_b32 v83, 0\n"
                 "v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n v_mov_b32 v86, 0\n v_mov_b32 v87, 0\n" ::: "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71",
                 "v72", "v73", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
    if (wave < 4) {
        if (mode & 1) {
            if (PRIO_M) __builtin_amdgcn_s_setprio(PRIO_M);
            if (FORM == 0)
                asm volatile("1:\n" MF16 MF16 "s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n s_nop 7\n s_nop 7\n" : "+s"(it_m) ::
                             "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "scc");
            else
                asm volatile("1:\n" MF32 MF32 "s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n s_nop 7\n s_nop 7\n" : "+s"(it_m) ::
                             "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19",
                             "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "scc");
        }
    } else if (mode & 2) {
        if (PRIO_V) __builtin_amdgcn_s_setprio(PRIO_V);
        asm volatile("1:\n" FMA8 FMA8 "s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n" : "+s"(it_v) ::
                     "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "scc");
    }
    if (it_m == 12345) out[threadIdx.x] = 1.f;
}

template <int FORM, int PM, int PV>
static float run(int mode, int it_m, int it_v, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<FORM, PM, PV><<<256, 512, 0>>>(out, mode, it_m, it_v);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) k<FORM, PM, PV><<<256, 512, 0>>>(out, mode, it_m, it_v);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

template <int FORM, int PM, int PV>
static void sweep(float* out) {
    const int it_m = 40000;
    const float tm = run<FORM, PM, PV>(1, it_m, 0, out);
    printf("form %s  prio(mfma wave) %d  prio(valu wave) %d:  MFMA stream alone %.3f ms\n", FORM ? "32x32x16" : "16x16x32", PM, PV, tm);
    for (int ratio : {1, 2, 3, 5}) {                       // VALU per 16x16x32-equivalent MFMA (8 per iteration): 16 fma per VALU iteration
        const int it_v = it_m * ratio / 2;
        const float tv = run<FORM, PM, PV>(2, 0, it_v, out), tb = run<FORM, PM, PV>(3, it_m, it_v, out);
        printf("   %d VALU per 16x16x32-equivalent MFMA: VALU alone %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", ratio, tv, tb, tm + tv, tm > tv ? tm : tv);
    }
}

int main() {
    float* out; (void)hipMalloc(&out, 4096);
    sweep<0, 0, 0>(out); sweep<0, 0, 3>(out); sweep<0, 3, 0>(out);
    sweep<1, 0, 0>(out); sweep<1, 0, 3>(out); sweep<1, 3, 0>(out);
    return 0;
}
