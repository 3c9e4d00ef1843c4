// Weight and bias gradient of the encoder's first layer (4x4 stride-2 pad-1 conv on the 3-channel NCHW image + LeakyReLU, no norm;
// backward of blox ConvEncoder's input block as called through /root/reference/gcp/prediction/models/base_gcp.py:188,208-209) in ONE
// launch.  The layer has no data gradient to pass on, so everything of its backward feeds these 16 x 48 + 16 numbers; before, that took an
// activation-backward pass (read 2, write 1 tensor of F x 32 x 32 x 16), an im2col of the image (write + read 48 floats per output
// pixel) and a [16 x 48] GEMM over 1.3 M rows that cannot fill the chip: 0.9 ms at the tail of the training step's side lanes.
//
// Here a persistent workgroup takes (frame, band of 8 output rows) items: the incoming gradient times the LeakyReLU slope goes to LDS
// as [pixel][16], the 18 image rows the band touches as [3][18][S + 8] with their zero border; k = output pixels (4 per
// v_mfma_f32_16x16x4_f32), i side = output channel, j side = (ky, kx) of one input channel — three tiles — plus a fourth tile whose B
// operand is the indicator of j = 0: its first column is the bias gradient.  Partials [grid][16 x 48 + 16] are summed by
// gcpx_reduce_partials in a fixed order (deterministic).
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) wgrad_image_kernel(const float* __restrict__ da, const float* __restrict__ add,
                                                          const float* __restrict__ r, const float* __restrict__ image,
                                                          float* __restrict__ partial, const int F, const int S) {
    extern __shared__ float4 smem4[];
    const int OW = S / 2, PITCH = S + 8, BH = 8;
    const int npx = BH * OW;                                   // output pixels of an item (256 at 64 x 64)
    float* sdu = reinterpret_cast<float*>(smem4);             // [npx][16]
    float* simg = sdu + npx * 16;                             // [3][18][PITCH]; image column c at index c + 4, zero border at 3 and S + 4
    float* sred = simg + 3 * 18 * PITCH;                      // [4 wavefronts][4 tiles][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ij = lane & 15, kk = lane >> 4;
    const int bands = OW / BH, nitems = F * bands;

    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0, 0, 0, 0};
    // zero borders of the image tile (never overwritten)
    for (int i = tid; i < 3 * 18; i += 256) { simg[i * PITCH + 3] = 0.f; simg[i * PITCH + S + 4] = 0.f; }
    const int boff = (ij >> 2) * PITCH + (ij & 3) + 3;        // (ky, kx) of this lane's column
    const float bias_b = ij == 0 ? 1.f : 0.f;

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int f = item / bands, oy0 = (item % bands) * BH;
        __syncthreads();                                       // the previous item's operand reads are done
        // ---- gradient tile: (da + add) * slope(r) ----
        const size_t gbase = ((size_t)f * OW + oy0) * OW * 16;
        for (int i = tid; i < npx * 4; i += 256) {
            float4 g = *reinterpret_cast<const float4*>(da + gbase + (size_t)i * 4);
            if (add) {
                const float4 v = *reinterpret_cast<const float4*>(add + gbase + (size_t)i * 4);
                g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
            }
            const float4 rv = *reinterpret_cast<const float4*>(r + gbase + (size_t)i * 4);
            g.x *= rv.x > 0.f ? 1.f : 0.2f; g.y *= rv.y > 0.f ? 1.f : 0.2f; g.z *= rv.z > 0.f ? 1.f : 0.2f; g.w *= rv.w > 0.f ? 1.f : 0.2f;
            *reinterpret_cast<float4*>(sdu + (size_t)i * 4) = g;
        }
        // ---- image rows 2 oy0 - 1 .. 2 oy0 + 16 of the three channels (rows outside the image: zero) ----
        const int S4 = S / 4;
        for (int i = tid; i < 3 * 18 * S4; i += 256) {
            const int c4 = i % S4, row = (i / S4) % 18, ci = i / (S4 * 18);
            const int iy = 2 * oy0 - 1 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < S) v = *reinterpret_cast<const float4*>(image + (((size_t)f * 3 + ci) * S + iy) * S + 4 * c4);
            *reinterpret_cast<float4*>(simg + (ci * 18 + row) * PITCH + 4 + 4 * c4) = v;
        }
        __syncthreads();
        // ---- k-steps of 4 output pixels, dealt round-robin over the wavefronts ----
        const int nsteps = npx / 4;
        for (int s = wave; s < nsteps; s += 4) {
            const int p = 4 * s + kk;
            const int oyl = p / OW, ox = p % OW;
            const float a = sdu[p * 16 + ij];
            const float* bp = simg + (2 * oyl) * PITCH + 2 * ox + boff;
            acc[0] = mfma16(a, bp[0], acc[0]);
            acc[1] = mfma16(a, bp[18 * PITCH], acc[1]);
            acc[2] = mfma16(a, bp[2 * 18 * PITCH], acc[2]);
            acc[3] = mfma16(a, bias_b, acc[3]);
        }
    }
    // ---- the four wavefronts' sums in a fixed order; lane holds rows co = 4 kk + reg, column ij of tile t ----
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) sred[(wave * 4 + t) * 256 + (4 * kk + rg) * 16 + ij] = acc[t][rg];
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * (16 * 48 + 16);
    for (int i = tid; i < 4 * 256; i += 256) {
        const int t = i >> 8, e = i & 255, co = e >> 4, j = e & 15;
        const float s = ((sred[(0 * 4 + t) * 256 + e] + sred[(1 * 4 + t) * 256 + e]) + sred[(2 * 4 + t) * 256 + e]) + sred[(3 * 4 + t) * 256 + e];
        if (t < 3) out[co * 48 + 16 * t + j] = s;
        else if (j == 0) out[16 * 48 + co] = s;
    }
}

}  // namespace

// da / add / r: NHWC [F][S/2][S/2][16] (incoming gradient, optional addend, activated forward output); image NCHW [F][3][S][S];
// partial [grid][16*48 + 16]: weight gradient [co][ci*16 + ky*4 + kx], then the bias gradient.  S in {32, 64, 128}.
extern "C" int gcpx_wgrad_image4x4s2(const float* da, const float* add, const float* r, const float* image, int32_t F, int32_t S,
                                     float* partial, int32_t grid, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(da && r && image && partial && F > 0 && grid > 0, "bad arguments");
    GCPX_CHECK_ARG(S == 32 || S == 64 || S == 128, "image size must be 32, 64 or 128");
    const int OW = S / 2, npx = 8 * OW;
    const size_t lds = ((size_t)npx * 16 + 3 * 18 * (S + 8) + 16 * 256) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_image_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_image_kernel, dim3(grid), dim3(256), lds, stream, da, add, r, image, partial, F, S);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
