// Encoder conv blocks on gfx950: 4x4 conv, stride 2, pad 1 as implicit GEMM on f32 MFMA.
//
// Replaces blox ConvEncoder blocks as called at /root/reference/gcp/prediction/models/base_gcp.py:188,208-209
// (this build's spec of the blocks: DESIGN.md "Model spec").
//
// The encoder is ~5 % of the forward FLOPs and every input element is used by only 4 taps, so these kernels
// skip LDS: each lane fetches its B fragment (one pixel, 4 consecutive input channels = 16 B) straight from
// L2/HBM and applies the producer's BatchNorm affine + LeakyReLU on the fly ("normalise on load").  Weights come
// pre-packed in fragment order (one coalesced 1 KiB load per 16 output channels per step).
#include "common.cuh"

namespace {

// generic NHWC layer: Cin % 16 == 0
template <int CT>
__global__ void __launch_bounds__(256) conv4x4s2_kernel(const gcpx_conv_args a, const int ngroups) {
    constexpr int PR = 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int Hin = a.Hin, Win = a.Win, Hout = a.Hout, Wout = a.Wout, Cin = a.Cin;
    const int ncg = Cin / 16;
    const int npix = a.F * Hout * Wout;
    const gcpx_conv_src s = a.src[0];
    const float4* wbase = reinterpret_cast<const float4*>(a.wpk) + lane;

    f32x4 st1[CT], st2[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { st1[ct] = f32x4{0, 0, 0, 0}; st2[ct] = f32x4{0, 0, 0, 0}; }

    const int nblk = (ngroups + 4 * PR - 1) / (4 * PR);
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        int pf[PR], poy[PR], pox[PR];
        bool pv[PR];
#pragma unroll
        for (int pt = 0; pt < PR; ++pt) {
            const int p = ((blk * 4 + wave) * PR + pt) * 16 + j;
            pv[pt] = p < npix;
            const int pp = pv[pt] ? p : 0;
            pox[pt] = pp % Wout;
            const int t = pp / Wout;
            poy[pt] = t % Hout;
            pf[pt] = t / Hout;
        }
        f32x4 acc[CT][PR];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PR; ++pt) acc[ct][pt] = f32x4{0, 0, 0, 0};

        for (int tap = 0; tap < 16; ++tap) {
            const int ky = tap >> 2, kx = tap & 3;
            const float* bp[PR];
            bool inb[PR];
#pragma unroll
            for (int pt = 0; pt < PR; ++pt) {
                const int iy = 2 * poy[pt] - 1 + ky, ix = 2 * pox[pt] - 1 + kx;
                inb[pt] = pv[pt] && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
                bp[pt] = s.ptr + (((size_t)pf[pt] * Hin + (inb[pt] ? iy : 0)) * Win + (inb[pt] ? ix : 0)) * Cin + q * 4;
            }
            // loads of UK channel groups go out together, then their MFMAs (latency-bound otherwise)
            constexpr int UK = (CT >= 8) ? 1 : (CT >= 4 ? 2 : 4);
            for (int cg = 0; cg < ncg; cg += UK) {
                float4 b[UK][PR], w[UK][CT];
#pragma unroll
                for (int u = 0; u < UK; ++u) {
                    const int c = (cg + u < ncg) ? cg + u : ncg - 1;
#pragma unroll
                    for (int pt = 0; pt < PR; ++pt) b[u][pt] = *reinterpret_cast<const float4*>(bp[pt] + c * 16);
                    const float4* wp = wbase + (size_t)(tap * ncg + c) * CT * 64;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) w[u][ct] = wp[ct * 64];
                }
#pragma unroll
                for (int u = 0; u < UK; ++u) {
                    if (cg + u < ncg) {
#pragma unroll
                        for (int pt = 0; pt < PR; ++pt) {
                            float4 bb = affine_act4(b[u][pt], s.scale, s.shift, (cg + u) * 16 + q * 4, s.act);
                            const float m = inb[pt] ? 1.f : 0.f;     // conv zero padding stays exactly zero
                            bb.x *= m; bb.y *= m; bb.z *= m; bb.w *= m;
                            b[u][pt] = bb;
                        }
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                            for (int pt = 0; pt < PR; ++pt) {
                                acc[ct][pt] = mfma16(w[u][ct].x, b[u][pt].x, acc[ct][pt]);
                                acc[ct][pt] = mfma16(w[u][ct].y, b[u][pt].y, acc[ct][pt]);
                                acc[ct][pt] = mfma16(w[u][ct].z, b[u][pt].z, acc[ct][pt]);
                                acc[ct][pt] = mfma16(w[u][ct].w, b[u][pt].w, acc[ct][pt]);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int pt = 0; pt < PR; ++pt) {
            if (!pv[pt]) continue;
            const size_t p = (size_t)((blk * 4 + wave) * PR + pt) * 16 + j;
            float* op = a.out + p * a.out_pitch;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + ct * 16 + q * 4);
                f32x4 v = acc[ct][pt];
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                if (a.out_act == GCPX_ACT_LRELU) {
                    v[0] = lrelu(v[0], 0.2f); v[1] = lrelu(v[1], 0.2f); v[2] = lrelu(v[2], 0.2f); v[3] = lrelu(v[3], 0.2f);
                }
                *reinterpret_cast<float4*>(op + ct * 16 + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
                if (a.stats_partial) { st1[ct] += v; st2[ct] += v * v; }
            }
        }
    }
    if (a.stats_partial) {
        __shared__ float red[4 * 2 * CT * 16];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s1 = row16_sum(st1[ct][r]);
                const float s2 = row16_sum(st2[ct][r]);
                if (j == 0) {
                    red[(wave * 2 + 0) * CT * 16 + ct * 16 + q * 4 + r] = s1;
                    red[(wave * 2 + 1) * CT * 16 + ct * 16 + q * 4 + r] = s2;
                }
            }
        }
        __syncthreads();
        if (tid < 2 * CT * 16) {
            const int which = tid / (CT * 16), c = tid % (CT * 16);
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) sum += red[(w * 2 + which) * CT * 16 + c];
            a.stats_partial[((size_t)blockIdx.x * 2 + which) * CT * 16 + c] = sum;
        }
    }
}

// first layer: NCHW 3-channel image, K ordered (ci, ky) x kx so that one MFMA consumes the 4 kx taps
template <int CT>
__global__ void __launch_bounds__(256) conv4x4s2_image_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ wpk,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              const int F, const int Hin, const int Win,
                                                              const int Cout, const int out_act) {
    constexpr int PR = 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int Hout = Hin / 2, Wout = Win / 2;
    const int npix = F * Hout * Wout;
    const int ngroups = (npix + 15) / 16;
    const int nblk = (ngroups + 4 * PR - 1) / (4 * PR);
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        int pf[PR], poy[PR], pox[PR];
        bool pv[PR];
#pragma unroll
        for (int pt = 0; pt < PR; ++pt) {
            const int p = ((blk * 4 + wave) * PR + pt) * 16 + j;
            pv[pt] = p < npix;
            const int pp = pv[pt] ? p : 0;
            pox[pt] = pp % Wout;
            const int t = pp / Wout;
            poy[pt] = t % Hout;
            pf[pt] = t / Hout;
        }
        f32x4 acc[CT][PR];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PR; ++pt) acc[ct][pt] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                float b[PR];
#pragma unroll
                for (int pt = 0; pt < PR; ++pt) {
                    const int iy = 2 * poy[pt] - 1 + ky, ix = 2 * pox[pt] - 1 + q;
                    const bool ok = pv[pt] && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
                    b[pt] = ok ? x[(((size_t)pf[pt] * 3 + ci) * Hin + iy) * Win + ix] : 0.f;
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float w = wpk[((ci * 4 + ky) * CT + ct) * 64 + lane];
#pragma unroll
                    for (int pt = 0; pt < PR; ++pt) acc[ct][pt] = mfma16(w, b[pt], acc[ct][pt]);
                }
            }
        }
#pragma unroll
        for (int pt = 0; pt < PR; ++pt) {
            if (!pv[pt]) continue;
            const size_t p = (size_t)((blk * 4 + wave) * PR + pt) * 16 + j;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = ct * 16 + q * 4;
                if (c >= Cout) continue;
                const float4 bv = *reinterpret_cast<const float4*>(bias + c);
                f32x4 v = acc[ct][pt];
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                if (out_act == GCPX_ACT_LRELU) {
                    v[0] = lrelu(v[0], 0.2f); v[1] = lrelu(v[1], 0.2f); v[2] = lrelu(v[2], 0.2f); v[3] = lrelu(v[3], 0.2f);
                }
                *reinterpret_cast<float4*>(out + p * Cout + c) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

}  // namespace

extern "C" int gcpx_conv4x4s2(const gcpx_conv_args* a, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(a != nullptr, "null args");
    GCPX_CHECK_ARG(a->nsrc == 1 && a->src[0].frame_div == 1, "encoder conv takes one per-frame source");
    GCPX_CHECK_ARG(a->Cin == a->src[0].C && a->Cin % 16 == 0, "Cin must equal src C and be a multiple of 16");
    GCPX_CHECK_ARG(a->Hout * 2 == a->Hin && a->Wout * 2 == a->Win, "Hout must be Hin/2");
    GCPX_CHECK_ARG(a->Cout % 16 == 0 && a->out_pitch == a->Cout, "Cout % 16 and dense output");
    GCPX_CHECK_ARG(a->wpk && a->bias && a->out && a->F > 0, "null pointer / F <= 0");
    const int npix = a->F * a->Hout * a->Wout;
    const int ngroups = (npix + 15) / 16;
    int grid = gcpx_conv_grid();
    const int nblk = (ngroups + 7) / 8;
    // stats_partial has gcpx_conv_grid() rows: all of them must be written
    if (!a->stats_partial && grid > nblk) grid = nblk;
    const int CT = a->Cout / 16;
    switch (CT) {
        case 2: hipLaunchKernelGGL(conv4x4s2_kernel<2>, dim3(grid), dim3(256), 0, stream, *a, ngroups); break;
        case 4: hipLaunchKernelGGL(conv4x4s2_kernel<4>, dim3(grid), dim3(256), 0, stream, *a, ngroups); break;
        case 8: hipLaunchKernelGGL(conv4x4s2_kernel<8>, dim3(grid), dim3(256), 0, stream, *a, ngroups); break;
        default:
            gcpx_set_error("conv4x4s2: unsupported Cout=%d", a->Cout);
            return GCPX_ERR_UNSUPPORTED;
    }
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_conv4x4s2_image(const float* x, const float* wpk, const float* bias, float* out, int32_t F,
                                    int32_t Hin, int32_t Win, int32_t Cout, int32_t out_act, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(x && wpk && bias && out && F > 0, "null pointer / F <= 0");
    GCPX_CHECK_ARG(Cout == 16, "first encoder layer: Cout must be 16 (ngf)");
    GCPX_CHECK_ARG(Hin % 2 == 0 && Win % 2 == 0, "even input size");
    const int npix = F * (Hin / 2) * (Win / 2);
    const int nblk = ((npix + 15) / 16 + 7) / 8;
    int grid = gcpx_conv_grid() * 2;
    if (grid > nblk) grid = nblk;
    hipLaunchKernelGGL(conv4x4s2_image_kernel<1>, dim3(grid), dim3(256), 0, stream, x, wpk, bias, out, F, Hin, Win,
                       Cout, out_act);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
