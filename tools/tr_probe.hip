// What ds_read_b64_tr_b16 returns: every lane supplies the address of an 8-byte chunk, the output names (source lane, half) of the four
// halfs each lane receives, for three address patterns.  hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/tr_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* chunk_of_lane, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    const short* p = lds + chunk_of_lane[lane] * 4;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    int h[64]; short o[256]; int* d; short* od;
    hipMalloc(&d, 256); hipMalloc(&od, 512);
    for (int mode = 0; mode < 3; ++mode) {
        for (int l = 0; l < 64; ++l) h[l] = mode == 0 ? l : mode == 1 ? (l * 7) % 64 + 64 : 3 * l + 5;
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d, od);
        hipMemcpy(o, od, 512, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d (chunk %3d):", l, h[l]);
            for (int j = 0; j < 4; ++j) {
                int e = o[l * 4 + j]; int c = e / 4, w = e % 4; int src = -1;
                for (int k = 0; k < 64; ++k) if (h[k] == c) src = k;
                printf("  e%4d=lane%2d.%d", e, src, w);
            }
            printf("\n");
        }
    }
    return 0;
}
