// Encoder conv blocks on gfx950: 4x4 conv, stride 2, pad 1 as implicit GEMM on f32 MFMA.
//
// Replaces blox ConvEncoder blocks as called at /root/reference/gcp/prediction/models/base_gcp.py:188,208-209
// (this build's spec of the blocks: DESIGN.md "Model spec").
//
// The encoder is ~5 % of the forward FLOPs and every input element is used by only 4 taps, so these kernels
// skip LDS: each lane fetches its B fragment (one pixel, 4 consecutive input channels = 16 B) straight from
// L2/HBM and applies the producer's BatchNorm affine + LeakyReLU on the fly ("normalise on load").  Weights come
// pre-packed in fragment order (one coalesced 1 KiB load per 16 output channels per step).
#include "common.h"

#include <cstdlib>

int gcpx_launch_enc_split(const gcpx_conv_args* a, hipStream_t stream, int nrows, int cus);     // conv_enc_split.hip

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

// LDS-resident variant for the layers whose whole input frame(s) fit in LDS (every encoder layer after the first at 32 x 32
// and 64 x 64 images).  A workgroup stages FPB complete input frames once — coalesced 16-byte loads, the producer's BatchNorm
// affine + LeakyReLU applied ONCE per element instead of once per tap — and computes all their output pixels from LDS:
// NG = FPB * HOUT^2 / 16 pixel groups over WP wavefront rows, COUT / 16 channel tiles over WC wavefront columns, so a weight
// fragment fetched from L2 feeds PGW pixel groups (the direct-from-global kernel above re-fetched activations 4x through L1
// and was L2-bound on the filter bank).  Out-of-image taps read a zero pixel kept behind the frames.
// FPB = most frames a workgroup stages at once; the launch picks fpb <= FPB so that the workgroups fill the CUs in whole rounds
// (1280 frames over 256 CUs: 5 frames each = one round, instead of 160 workgroups of 8).  ZP: out-of-image taps read a zero
// pixel kept behind the frames; without it (layers whose frame + zero pixel would not let two workgroups share a CU's LDS)
// the loaded value is replaced by zero instead.
template <int CIN, int COUT, int HIN, int FPB, int WP, int WC, bool ZP>
struct EncLdsCfg {
    static constexpr int HOUT = HIN / 2, CT = COUT / 16, CTW = CT / WC, NCG = CIN / 16, CINP = CIN + 4;
    static constexpr int NG = FPB * HOUT * HOUT / 16, PGW = (NG + WP - 1) / WP;
    static constexpr int FRAME_PIX = HIN * HIN;
    static constexpr int LDS_BYTES = (FPB * FRAME_PIX * CINP + (ZP ? CINP : 0)) * 4;    // at the largest group
    static constexpr int lds_bytes(int fpb) { return (fpb * FRAME_PIX * CINP + (ZP ? CINP : 0)) * 4; }
    static_assert(WP * WC == 4 && CT % WC == 0 && (HOUT * HOUT) % 16 == 0, "tiling");
};

template <int CIN, int COUT, int HIN, int FPB, int WP, int WC, bool ZP>
__global__ void __launch_bounds__(256) conv4x4s2_lds_kernel(const gcpx_conv_args a, const int nblk, const int nrows, const int fpb) {
    using Cfg = EncLdsCfg<CIN, COUT, HIN, FPB, WP, WC, ZP>;
    constexpr int HOUT = Cfg::HOUT, CT = Cfg::CT, CTW = Cfg::CTW, NCG = Cfg::NCG, CINP = Cfg::CINP, PGW = Cfg::PGW;
    constexpr int FRAME_PIX = Cfg::FRAME_PIX, C4 = CIN / 4;
    const int tot4 = fpb * FRAME_PIX * C4;
    const int ngrp = fpb * HOUT * HOUT / 16;             // pixel groups of a full workgroup, dealt round-robin to the WP rows
    const int npg = __builtin_amdgcn_readfirstlane((ngrp - (int)((threadIdx.x >> 6) % WP) + WP - 1) / WP);   // groups of this wavefront
    extern __shared__ float4 enc_smem4[];
    float* lds = reinterpret_cast<float*>(enc_smem4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave % WP, wc = wave / WP;
    const int j = lane & 15, q = lane >> 4;
    const gcpx_conv_src s = a.src[0];
    const float4* wbase = reinterpret_cast<const float4*>(a.wpk) + (size_t)wc * CTW * 64 + lane;
    const int zero_off = fpb * FRAME_PIX * CINP;         // float offset of the zero pixel (behind the staged frames)
    if (ZP && tid < CINP) lds[zero_off + tid] = 0.f;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s.scale) {
        sc4 = *reinterpret_cast<const float4*>(s.scale + 4 * (tid % (CIN / 4)));
        sh4 = *reinterpret_cast<const float4*>(s.shift + 4 * (tid % (CIN / 4)));
    }

    f32x4 st1[CTW], st2[CTW];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) { st1[ct] = f32x4{0, 0, 0, 0}; st2[ct] = f32x4{0, 0, 0, 0}; }

    // per pixel group of this wavefront: LDS offset of input pixel (2 oy - 1, 2 ox - 1) + this lane's 4-channel slice, tap validity
    int base[PGW], fl_[PGW], pin_[PGW];
    unsigned vmask[PGW];                                   // bit ky (0..3): row valid, bit 4 + kx: column valid
#pragma unroll
    for (int pt = 0; pt < PGW; ++pt) {
        const int p = (wp + pt * WP) * 16 + j;             // pixel index inside the workgroup's frames (group wp + pt * WP)
        const int fl = p / (HOUT * HOUT), pin = p % (HOUT * HOUT);
        const int oy = pin / HOUT, ox = pin % HOUT;
        fl_[pt] = fl; pin_[pt] = pin;
        base[pt] = ((fl * HIN + 2 * oy - 1) * HIN + 2 * ox - 1) * CINP + 4 * q;
        unsigned m = 0;
        for (int k = 0; k < 4; ++k) {
            if (2 * oy - 1 + k >= 0 && 2 * oy - 1 + k < HIN) m |= 1u << k;
            if (2 * ox - 1 + k >= 0 && 2 * ox - 1 + k < HIN) m |= 16u << k;
        }
        vmask[pt] = m;
    }

    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int f0 = blk * fpb;
        __syncthreads();                                    // the previous iteration's operand reads are done
        const float4* src = reinterpret_cast<const float4*>(s.ptr) + (size_t)f0 * FRAME_PIX * C4;
        const int nvalid4 = min(fpb, a.F - f0) * FRAME_PIX * C4;
        // 8 independent 16-byte loads in flight per thread (a load -> affine -> store loop is one HBM latency per iteration);
        // 256 % C4 == 0, so a thread always handles the same 4 channels: its scale / shift live in registers
        static_assert(256 % C4 == 0, "staging batches");
#pragma unroll 1
        for (int i0 = 0; i0 < tot4; i0 += 256 * 8) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = i0 + k * 256 + tid;
                v[k] = idx < nvalid4 ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = i0 + k * 256 + tid;
                const int c4 = idx % C4, pix = idx / C4;
                if (idx < nvalid4) {
                    float4 x = v[k];
                    x.x = fmaf(x.x, sc4.x, sh4.x); x.y = fmaf(x.y, sc4.y, sh4.y); x.z = fmaf(x.z, sc4.z, sh4.z); x.w = fmaf(x.w, sc4.w, sh4.w);
                    if (s.act == GCPX_ACT_LRELU) { x.x = lrelu(x.x, 0.2f); x.y = lrelu(x.y, 0.2f); x.z = lrelu(x.z, 0.2f); x.w = lrelu(x.w, 0.2f); }
                    v[k] = x;
                }
                if (idx < tot4) *reinterpret_cast<float4*>(lds + pix * CINP + 4 * c4) = v[k];
            }
        }
        __syncthreads();

        f32x4 acc[CTW][PGW];
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
            for (int pt = 0; pt < PGW; ++pt) acc[ct][pt] = f32x4{0, 0, 0, 0};
        auto fetch = [&](const int step, float4 (&w)[CTW], float4 (&b)[PGW]) {
            const int tap = step / NCG, cg = step % NCG;
            const int ky = tap >> 2, kx = tap & 3;
            const int tapoff = (ky * HIN + kx) * CINP + cg * 16;
            const float4* wp_ = wbase + (size_t)step * CT * 64;
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) w[ct] = wp_[ct * 64];
#pragma unroll
            for (int pt = 0; pt < PGW; ++pt) {
                if (pt >= npg) break;                         // wave-uniform: this wavefront's pixel groups beyond the staged frames
                const bool ok = ((vmask[pt] >> ky) & 1u) && ((vmask[pt] >> (4 + kx)) & 1u);
                if constexpr (ZP) {
                    const int off = ok ? base[pt] + tapoff : zero_off + 4 * q;
                    b[pt] = *reinterpret_cast<const float4*>(lds + off);
                } else {
                    const float4 t = *reinterpret_cast<const float4*>(lds + (ok ? base[pt] + tapoff : 4 * q));
                    b[pt] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        };
        auto mm = [&](const float4 (&w)[CTW], const float4 (&b)[PGW]) {
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) {
#pragma unroll
                for (int pt = 0; pt < PGW; ++pt) {
                    if (pt >= npg) break;
                    acc[ct][pt] = mfma16(w[ct].x, b[pt].x, acc[ct][pt]);
                    acc[ct][pt] = mfma16(w[ct].y, b[pt].y, acc[ct][pt]);
                    acc[ct][pt] = mfma16(w[ct].z, b[pt].z, acc[ct][pt]);
                    acc[ct][pt] = mfma16(w[ct].w, b[pt].w, acc[ct][pt]);
                }
            }
        };
        // two register sets in ping-pong (no copies: a copy lets the scheduler pull the wait for the NEXT step's loads in front of
        // this step's MFMAs): the operands of step s + 1 are in flight during the 64 MFMAs of step s
        float4 wA[CTW], bA[PGW], wB[CTW], bB[PGW];
        constexpr int NSTEP = 16 * NCG;
        static_assert(NSTEP % 2 == 0, "ping-pong");
        fetch(0, wA, bA);
#pragma unroll 1
        for (int step = 0; step < NSTEP; step += 2) {
            fetch(step + 1, wB, bB);
            mm(wA, bA);
            if (step + 2 < NSTEP) fetch(step + 2, wA, bA);
            mm(wB, bB);
        }
#pragma unroll
        for (int pt = 0; pt < PGW; ++pt) {
            const int f = f0 + fl_[pt];
            if (f >= a.F || wp + pt * WP >= ngrp) continue;
            float* op = a.out + ((size_t)f * HOUT * HOUT + pin_[pt]) * COUT;
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
        __syncthreads();
        float* red = lds;                                   // [WP][2][CT*16], the frames are no longer needed
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
            for (int w_ = 0; w_ < WP; ++w_) sum += red[(w_ * 2 + which) * CT * 16 + c];
            a.stats_partial[((size_t)blockIdx.x * 2 + which) * CT * 16 + c] = sum;
            // rows of workgroups that were not launched (every launched one carries its frames in LDS): zero
            for (int r = blockIdx.x + gridDim.x; r < nrows; r += gridDim.x) a.stats_partial[((size_t)r * 2 + which) * CT * 16 + c] = 0.f;
        }
    }
}

template <int CIN, int COUT, int HIN, int FPB, int WP, int WC, bool ZP>
int launch_enc_lds(const gcpx_conv_args* a, hipStream_t stream) {
    using Cfg = EncLdsCfg<CIN, COUT, HIN, FPB, WP, WC, ZP>;
    auto kern = conv4x4s2_lds_kernel<CIN, COUT, HIN, FPB, WP, WC, ZP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) {
            gcpx_set_error("conv4x4s2 (LDS): hipFuncSetAttribute(%d B): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        attr_set = true;
    }
    // frames per workgroup: the most loaded CU should carry as few frames as possible (workgroups are dealt round-robin to the
    // CUs); among equals the larger group (weight fragments are fetched once per group; measured at 1280 frames: 4x4 layer
    // 62 us with 5 frames / workgroup, 97 us with 1)
    const int cus = gcpx_conv_grid() / 2;
    int fpb = 1;
    long best = -1;
    for (int c = 1; c <= FPB; ++c) {
        const int nb = (a->F + c - 1) / c;
        const long cost = (long)((nb + cus - 1) / cus) * c;
        if (best < 0 || cost <= best) { best = cost; fpb = c; }
    }
    if (const char* e = getenv("GCPX_ENC_FPB")) { const int v = atoi(e); if (v >= 1 && v <= FPB) fpb = v; }
    const int nblk = (a->F + fpb - 1) / fpb;
    // stats_partial has gcpx_conv4x4s2_grid() rows and all of them must be written: the launched workgroups zero the others
    const int nrows = gcpx_conv4x4s2_grid();
    const int grid = nblk < nrows ? nblk : nrows;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::lds_bytes(fpb), stream, *a, nblk, nrows, fpb);
    return GCPX_OK;
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
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, F * 3 * Hin * Win * 4, 0x00020000);
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
        // all 12 x PR taps are requested before the first MFMA: buffer loads against the image tensor, a tap outside the image
        // carries an offset beyond the buffer and reads 0 (with one branch + wait per tap the launch was a chain of 12 memory
        // latencies per 32 pixels: 92 us for the 63 MB trajectory batch)
        float b[12][PR];
#pragma unroll
        for (int pt = 0; pt < PR; ++pt) {
            const int ix = 2 * pox[pt] - 1 + q;
            const bool xok = pv[pt] && ix >= 0 && ix < Win;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) {
                    const int iy = 2 * poy[pt] - 1 + ky;
                    const bool ok = xok && iy >= 0 && iy < Hin;
                    const unsigned off = ok ? (unsigned)(((pf[pt] * 3 + ci) * Hin + iy) * Win + ix) * 4u : 0x80000000u;
                    b[ci * 4 + ky][pt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 12; ++t) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float w = wpk[(t * CT + ct) * 64 + lane];
#pragma unroll
                for (int pt = 0; pt < PR; ++pt) acc[ct][pt] = mfma16(w, b[t][pt], acc[ct][pt]);
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

// First layer with the frame resident in LDS.  conv4x4s2_image_kernel fetches every tap of every pixel from global memory with 4-byte
// loads: each input element travels through the vector memory path four times (16 taps, stride 2 x 2) and the launch runs at 2 TB/s
// of traffic (68 us for the 63 MB trajectory batch).  Here a workgroup stages one whole frame — coalesced 16-byte loads, three planes of
// (HIN + 2) rows x PITCH floats with the conv's zero border stored as zeros (no bounds checks on the operand reads) — and its four
// wavefronts compute the frame's HOUT x WOUT x 16 outputs from LDS: lane (j, q) of k-step (ci, ky) reads x[ci][2 oy - 1 + ky][2 ox_j - 1 + q].
// Same products in the same order as conv4x4s2_image_kernel: bit-identical results.
template <int HIN>
struct ImgLdsCfg {
    static constexpr int PITCH = HIN + 8;                      // floats per staged row: data at columns 4 .. HIN + 3 (16-byte aligned), zero at 3 and HIN + 4
    static constexpr int ROWS = HIN + 2;
    static constexpr int PLANE = ROWS * PITCH;
    static constexpr int LDS_BYTES = 3 * PLANE * 4;
};

template <int HIN>
__global__ void __launch_bounds__(256) conv4x4s2_image_lds_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                                  const float* __restrict__ bias, float* __restrict__ out, const int F,
                                                                  const int out_act) {
    using Cfg = ImgLdsCfg<HIN>;
    constexpr int PITCH = Cfg::PITCH, PLANE = Cfg::PLANE, HOUT = HIN / 2, NG = HOUT * HOUT / 16;
    extern __shared__ float4 img_smem4[];
    float* lds = reinterpret_cast<float*>(img_smem4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    // zero border: rows 0 and HIN + 1 of every plane, columns 3 and HIN + 4 of every row (written once: staging never touches them)
    for (int i = tid; i < 3 * 2 * PITCH; i += 256) {
        const int pl = i / (2 * PITCH), r = (i / PITCH) % 2, c = i % PITCH;
        lds[pl * PLANE + (r ? HIN + 1 : 0) * PITCH + c] = 0.f;
    }
    for (int i = tid; i < 3 * Cfg::ROWS * 2; i += 256) {
        const int pl = i / (Cfg::ROWS * 2), r = (i / 2) % Cfg::ROWS, c = (i & 1) ? HIN + 4 : 3;
        lds[pl * PLANE + r * PITCH + c] = 0.f;
    }
    float w[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) w[t] = wpk[t * 64 + lane];
    const float4 bv = *reinterpret_cast<const float4*>(bias + q * 4);
    constexpr int F4 = 3 * HIN * HIN / 4;                       // float4 of a frame
    for (int f = blockIdx.x; f < F; f += gridDim.x) {
        __syncthreads();                                        // the previous frame's operand reads are done
        const float4* src = reinterpret_cast<const float4*>(x) + (size_t)f * F4;
        for (int i = tid; i < F4; i += 256) {
            const int pl = i / (HIN * HIN / 4), rem = i % (HIN * HIN / 4);
            const int y = rem / (HIN / 4), x4 = rem % (HIN / 4);
            *reinterpret_cast<float4*>(lds + pl * PLANE + (y + 1) * PITCH + 4 + 4 * x4) = src[i];
        }
        __syncthreads();
        for (int g = wave; g < NG; g += 4) {
            const int p = g * 16 + j;
            const int oy = p / HOUT, ox = p % HOUT;
            const float* bp = lds + (2 * oy) * PITCH + 3 + 2 * ox + q;      // x[.][2 oy - 1 + ky][2 ox - 1 + q] = row 2 oy + ky, column 3 + 2 ox + q
            f32x4 acc = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) acc = mfma16(w[ci * 4 + ky], bp[ci * PLANE + ky * PITCH], acc);
            f32x4 v = acc;
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            if (out_act == GCPX_ACT_LRELU) {
                v[0] = lrelu(v[0], 0.2f); v[1] = lrelu(v[1], 0.2f); v[2] = lrelu(v[2], 0.2f); v[3] = lrelu(v[3], 0.2f);
            }
            *reinterpret_cast<float4*>(out + ((size_t)f * HOUT * HOUT + p) * 16 + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

template <int HIN>
int launch_image_lds(const float* x, const float* wpk, const float* bias, float* out, int F, int out_act, hipStream_t stream) {
    using Cfg = ImgLdsCfg<HIN>;
    auto kern = conv4x4s2_image_lds_kernel<HIN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) {
            gcpx_set_error("conv4x4s2 image (LDS): hipFuncSetAttribute(%d B): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        attr_set = true;
    }
    int grid = gcpx_conv_grid() / 2 * (160 * 1024 / Cfg::LDS_BYTES);      // as many workgroups as fit the CUs' LDS at once
    if (grid > F) grid = F;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES, stream, x, wpk, bias, out, F, out_act);
    return GCPX_OK;
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
    if (a->wpk_split && a->split_layout == GCPX_SPLIT_PLAIN && !getenv("GCPX_ENC_NOSPLIT")) {
        const int st = gcpx_launch_enc_split(a, stream, gcpx_conv4x4s2_grid(), gcpx_conv_grid() / 2);
        if (st < 0) return st;
        if (st == 0) {
            GCPX_CHECK_LAUNCH();
            return GCPX_OK;
        }
    }
    if (a->Hin == a->Win && !getenv("GCPX_ENC_DIRECT")) {
        int st = 1;
        if (a->Cin == 16 && a->Cout == 32 && a->Hin == 32) st = launch_enc_lds<16, 32, 32, 1, 4, 1, false>(a, stream);
        else if (a->Cin == 32 && a->Cout == 64 && a->Hin == 16) st = launch_enc_lds<32, 64, 16, 4, 4, 1, true>(a, stream);
        else if (a->Cin == 64 && a->Cout == 128 && a->Hin == 8) st = launch_enc_lds<64, 128, 8, 8, 1, 4, true>(a, stream);
        else if (a->Cin == 16 && a->Cout == 32 && a->Hin == 16) st = launch_enc_lds<16, 32, 16, 4, 4, 1, true>(a, stream);
        else if (a->Cin == 32 && a->Cout == 64 && a->Hin == 8) st = launch_enc_lds<32, 64, 8, 16, 4, 1, true>(a, stream);
        if (st <= 0) {
            if (st < 0) return st;
            GCPX_CHECK_LAUNCH();
            return GCPX_OK;
        }
    }
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
    GCPX_CHECK_ARG((long long)F * 3 * Hin * Win * 4 < (1LL << 31), "image tensor must stay below 2 GiB (32-bit buffer offsets)");
    if (Hin == Win && (Hin == 64 || Hin == 32) && !getenv("GCPX_ENC_IMAGE_DIRECT")) {
        const int st = Hin == 64 ? launch_image_lds<64>(x, wpk, bias, out, F, out_act, stream)
                                 : launch_image_lds<32>(x, wpk, bias, out, F, out_act, stream);
        if (st != GCPX_OK) return st;
        GCPX_CHECK_LAUNCH();
        return GCPX_OK;
    }
    const int npix = F * (Hin / 2) * (Win / 2);
    const int nblk = ((npix + 15) / 16 + 7) / 8;
    int grid = gcpx_conv_grid() * 2;
    if (grid > nblk) grid = nblk;
    hipLaunchKernelGGL(conv4x4s2_image_kernel<1>, dim3(grid), dim3(256), 0, stream, x, wpk, bias, out, F, Hin, Win,
                       Cout, out_act);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
