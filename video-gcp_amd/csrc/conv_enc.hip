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

// generic NHWC layer: Cin % 16 == 0.
// A 256-thread workgroup owns WP*PRW pixel groups (16 pixels each); its 4 wavefronts are arranged WC x WP:
// wave (wc, wp) computes CTW output-channel tiles [wc*CTW, (wc+1)*CTW) for PRW pixel groups.  Spreading the channel
// tiles over wavefronts (instead of giving every wavefront all channels of a few pixels) divides the L2->L1 weight
// traffic by WC: with all-channels-per-wave the layer was L2-bandwidth bound (each wave re-read the whole filter
// bank, 131-524 KB, for 32 pixels).
template <int CTW, int PRW, int WC, int WP>
__global__ void __launch_bounds__(256) conv4x4s2_kernel(const gcpx_conv_args a, const int ngroups) {
    static_assert(WC * WP == 4, "4 wavefronts");
    constexpr int CT = CTW * WC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave % WC, wp = wave / WC;
    const int j = lane & 15, q = lane >> 4;
    const int Hin = a.Hin, Win = a.Win, Hout = a.Hout, Wout = a.Wout, Cin = a.Cin;
    const int ncg = Cin / 16;
    const int npix = a.F * Hout * Wout;
    const gcpx_conv_src s = a.src[0];
    const float4* wbase = reinterpret_cast<const float4*>(a.wpk) + (size_t)wc * CTW * 64 + lane;

    f32x4 st1[CTW], st2[CTW];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) { st1[ct] = f32x4{0, 0, 0, 0}; st2[ct] = f32x4{0, 0, 0, 0}; }

    const int nblk = (ngroups + WP * PRW - 1) / (WP * PRW);
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        int pf[PRW], poy[PRW], pox[PRW];
        bool pv[PRW];
#pragma unroll
        for (int pt = 0; pt < PRW; ++pt) {
            const int p = ((blk * WP + wp) * PRW + pt) * 16 + j;
            pv[pt] = p < npix;
            const int pp = pv[pt] ? p : 0;
            pox[pt] = pp % Wout;
            const int t = pp / Wout;
            poy[pt] = t % Hout;
            pf[pt] = t / Hout;
        }
        f32x4 acc[CTW][PRW];
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
            for (int pt = 0; pt < PRW; ++pt) acc[ct][pt] = f32x4{0, 0, 0, 0};

        for (int tap = 0; tap < 16; ++tap) {
            const int ky = tap >> 2, kx = tap & 3;
            const float* bp[PRW];
            float msk[PRW];
#pragma unroll
            for (int pt = 0; pt < PRW; ++pt) {
                const int iy = 2 * poy[pt] - 1 + ky, ix = 2 * pox[pt] - 1 + kx;
                const bool inb = pv[pt] && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
                msk[pt] = inb ? 1.f : 0.f;                      // conv zero padding stays exactly zero
                bp[pt] = s.ptr + (((size_t)pf[pt] * Hin + (inb ? iy : 0)) * Win + (inb ? ix : 0)) * Cin + q * 4;
            }
            for (int cg = 0; cg < ncg; ++cg) {
                float4 b[PRW], w[CTW];
#pragma unroll
                for (int pt = 0; pt < PRW; ++pt) b[pt] = *reinterpret_cast<const float4*>(bp[pt] + cg * 16);
                const float4* wp_ = wbase + (size_t)(tap * ncg + cg) * CT * 64;
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct) w[ct] = wp_[ct * 64];
#pragma unroll
                for (int pt = 0; pt < PRW; ++pt) {
                    float4 bb = affine_act4(b[pt], s.scale, s.shift, cg * 16 + q * 4, s.act);
                    bb.x *= msk[pt]; bb.y *= msk[pt]; bb.z *= msk[pt]; bb.w *= msk[pt];
                    b[pt] = bb;
                }
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct) {
#pragma unroll
                    for (int pt = 0; pt < PRW; ++pt) {
                        acc[ct][pt] = mfma16(w[ct].x, b[pt].x, acc[ct][pt]);
                        acc[ct][pt] = mfma16(w[ct].y, b[pt].y, acc[ct][pt]);
                        acc[ct][pt] = mfma16(w[ct].z, b[pt].z, acc[ct][pt]);
                        acc[ct][pt] = mfma16(w[ct].w, b[pt].w, acc[ct][pt]);
                    }
                }
            }
        }
#pragma unroll
        for (int pt = 0; pt < PRW; ++pt) {
            if (!pv[pt]) continue;
            const size_t p = (size_t)((blk * WP + wp) * PRW + pt) * 16 + j;
            float* op = a.out + p * a.out_pitch;
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) {
                const int c = (wc * CTW + ct) * 16 + q * 4;
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + c);
                f32x4 v = acc[ct][pt];
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                if (a.out_act == GCPX_ACT_LRELU) {
                    v[0] = lrelu(v[0], 0.2f); v[1] = lrelu(v[1], 0.2f); v[2] = lrelu(v[2], 0.2f); v[3] = lrelu(v[3], 0.2f);
                }
                *reinterpret_cast<float4*>(op + c) = make_float4(v[0], v[1], v[2], v[3]);
                if (a.stats_partial) { st1[ct] += v; st2[ct] += v * v; }
            }
        }
    }
    if (a.stats_partial) {
        // deterministic per-workgroup partial sums: lanes of a row, then the WP wavefronts that share channels
        __shared__ float red[WP * 2 * CT * 16];
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s1 = row16_sum(st1[ct][r]);
                const float s2 = row16_sum(st2[ct][r]);
                if (j == 0) {
                    const int c = (wc * CTW + ct) * 16 + q * 4 + r;
                    red[(wp * 2 + 0) * CT * 16 + c] = s1;
                    red[(wp * 2 + 1) * CT * 16 + c] = s2;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * CT * 16; i += 256) {
            const int which = i / (CT * 16), c = i % (CT * 16);
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < WP; ++w) sum += red[(w * 2 + which) * CT * 16 + c];
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

extern "C" int gcpx_conv4x4s2_grid(void) { return gcpx_conv_grid() * 4; }

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
    // these kernels hide their global-load latency with occupancy, not with LDS staging: 8 workgroups per CU
    int grid = gcpx_conv4x4s2_grid();
    const int CT = a->Cout / 16;
    const int per_blk = (CT == 2) ? 16 : 8;      // pixel groups per workgroup (WP * PRW)
    const int nblk = (ngroups + per_blk - 1) / per_blk;
    // stats_partial has gcpx_conv4x4s2_grid() rows: all of them must be written
    if (!a->stats_partial && grid > nblk) grid = nblk;
    switch (CT) {
        case 2: hipLaunchKernelGGL((conv4x4s2_kernel<1, 8, 2, 2>), dim3(grid), dim3(256), 0, stream, *a, ngroups); break;
        // the deeper layers have few output pixels (20k at 4x4): keep every wavefront on all channels of 32 pixels so
        // that there are enough workgroups; measured 105 / 121 us vs 191 / 199 us with channel-split wavefronts (c2)
        case 4: hipLaunchKernelGGL((conv4x4s2_kernel<4, 2, 1, 4>), dim3(grid), dim3(256), 0, stream, *a, ngroups); break;
        case 8: hipLaunchKernelGGL((conv4x4s2_kernel<8, 2, 1, 4>), dim3(grid), dim3(256), 0, stream, *a, ngroups); break;
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
