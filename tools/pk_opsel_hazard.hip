// Reproducer for the wrong values the training head's gradient rows carried in round 4 (NOTEBOOK "a store-data hazard", re-read in
// round 5): two packed-f32 VALU instructions back to back, the second reading the HIGH half of the first one's 64-bit result into its
// LOW lane through op_sel.  With the other wavefront of the SIMD busy, lanes 32..63 of the consumer now and then see the register's
// PREVIOUS content.  Build: hipcc --offload-arch=gfx950 -O2 tools/pk_opsel_hazard.hip -o tools/pk_opsel_hazard.bin; run: ./pk_opsel_hazard.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int GAP, bool CROSS>      // GAP: s_nop states between producer and consumer; CROSS: consumer reads the high half into its low lane
__global__ void __launch_bounds__(512) probe(unsigned long long* bad, unsigned* lanes, const int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x2 x = {1.0f + lane * 0.01f, 2.0f + lane * 0.02f}, y = {0.5f + wave, 3.0f}, z = {7.0f, -1.25f};
    f32x4 acc = {0, 0, 0, 0};
    h8 ha, hb;
    for (int k = 0; k < 8; ++k) { ha[k] = (_Float16)(lane * 0.001f); hb[k] = (_Float16)1.0f; }
    float t = 0.3f + lane;
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        // the partner's kind of work: waves 4..7 issue matrix instructions and transcendentals, waves 0..3 mostly the probe
        if (wave >= 4) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0); t = __builtin_amdgcn_exp2f(t * 0.5f) + acc[0] * 1e-30f; }
        }
        f32x2 a = {0.f, 0.f}, b;
        asm volatile("v_pk_mul_f32 %0, %2, %3\n\t"
                     ".rept %5\n\ts_nop 0\n\t.endr\n\t"
                     ".if %6\n\tv_pk_mul_f32 %1, %4, %0 op_sel:[0,1]\n\t.else\n\tv_pk_mul_f32 %1, %4, %0\n\t.endif"
                     : "+v"(a), "=&v"(b) : "v"(x), "v"(y), "v"(z), "n"(GAP), "n"(CROSS ? 1 : 0));
        const float ahi = x.y * y.y, alo = x.x * y.x;
        const float want_lo = CROSS ? z.x * ahi : z.x * alo, want_hi = z.y * ahi;
        if (b.x != want_lo || b.y != want_hi) { ++nbad; atomicOr(&lanes[lane >> 5], 1u); if (b.x == 0.f) atomicOr(&lanes[2], 1u); }
        x.x += 1e-3f; x.y -= 1e-3f;
    }
    if (nbad) atomicAdd(bad, nbad);
    if (t == 12345.f) bad[1] = (unsigned long long)acc[1];       // keeps the partner's work alive
}

template <int GAP, bool CROSS>
void run(const char* name) {
    unsigned long long* bad; unsigned* lanes;
    hipMalloc(&bad, 16); hipMalloc(&lanes, 16); hipMemset(bad, 0, 16); hipMemset(lanes, 0, 16);
    const int iters = 200000;
    hipLaunchKernelGGL((probe<GAP, CROSS>), dim3(256), dim3(512), 0, 0, bad, lanes, iters);
    unsigned long long h[2]; unsigned l[4];
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(l, lanes, 16, hipMemcpyDeviceToHost);
    printf("%-58s wrong %llu of %.3g  (lanes 0..31: %s, lanes 32..63: %s, low result == 0 seen: %s)\n", name, h[0], 256.0 * 512 * iters,
           l[0] ? "yes" : "no", l[1] ? "yes" : "no", l[2] ? "yes" : "no");
    hipFree(bad); hipFree(lanes);
}
int main() {
    run<0, true>("v_pk_mul_f32 -> v_pk_mul_f32 op_sel:[0,1], back to back");
    run<1, true>("the same with s_nop 0 between");
    run<2, true>("the same with two s_nop 0 between");
    run<4, true>("the same with four s_nop 0 between");
    run<0, false>("v_pk_mul_f32 -> v_pk_mul_f32 (no op_sel), back to back");
    return 0;
}
