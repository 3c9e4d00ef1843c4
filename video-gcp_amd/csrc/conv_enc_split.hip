// Encoder conv blocks (4x4, stride 2, pad 1) on the f16 matrix pipes of gfx950 with f32-equivalent arithmetic ("split-f16").
//
// Replaces the same blox ConvEncoder blocks as conv_enc.hip (/root/reference/gcp/prediction/models/base_gcp.py:188,208-209), for
// the layers whose input frames fit in LDS.  conv4x4s2_lds_kernel is bound by the f32 MFMA rate (62-85 us per layer over 1280
// frames against a 37 us MFMA floor and 5-22 us of operand traffic); this kernel keeps its decomposition — a workgroup stages FPB
// whole input frames (producer's BatchNorm affine + LeakyReLU applied once per element), NG pixel groups over WP wavefront rows,
// COUT / 16 channel tiles over WC wavefront columns, out-of-image taps read a zero pixel — and computes with both operands as
// two f16 pieces, three v_mfma_f32_16x16x32_f16 per f32 product (arithmetic and error: conv3x3_split.hip):
//   * the frames are staged as f32, the workgroup's largest magnitude gives ONE power-of-two scale for the block (every output
//     pixel of the block accumulates over input pixels of the block only), and the frames are converted IN PLACE: the 32 bytes of
//     8 consecutive channels of a pixel become [8 first pieces | 8 second pieces] — the two ds_read_b128 operands of a lane;
//   * one k-step = 32 k: two taps x 16 channels (CIN = 16), one tap x 32 channels (CIN = 32) or half a tap (CIN = 64);
//     weights come from L2 in fragment order (packing.pack_conv4x4_split), one step ahead in ping-pong registers.
#include "common.h"

#include <cstdlib>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma32h(h8 a, h8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <int CIN, int COUT, int HIN, int FPB, int WP, int WC, bool ZP>
struct EncSplitCfg {
    static constexpr int HOUT = HIN / 2, CT = COUT / 16, CTW = CT / WC;
    static constexpr int PIXB = CIN * 4 + 16;                   // bytes per staged pixel (16 B pad: operand reads 2-way at worst)
    static constexpr int NG = FPB * HOUT * HOUT / 16, PGW = (NG + WP - 1) / WP;
    static constexpr int FRAME_PIX = HIN * HIN;
    static constexpr int NSTEP = CIN == 16 ? 8 : 16 * (CIN / 32);
    static constexpr int lds_bytes(int fpb) { return fpb * FRAME_PIX * PIXB + (ZP ? PIXB : 0); }
    static constexpr int LDS_BYTES = lds_bytes(FPB);
    static_assert(WP * WC == 4 && CT % WC == 0 && (HOUT * HOUT) % 16 == 0 && (CIN == 16 || CIN % 32 == 0), "tiling");
};

template <int CIN, int COUT, int HIN, int FPB, int WP, int WC, bool ZP>
__global__ void __launch_bounds__(256) conv4x4s2_split_kernel(const gcpx_conv_args a, const int nblk, const int nrows, const int fpb) {
    using Cfg = EncSplitCfg<CIN, COUT, HIN, FPB, WP, WC, ZP>;
    constexpr int HOUT = Cfg::HOUT, CT = Cfg::CT, CTW = Cfg::CTW, PIXB = Cfg::PIXB, PGW = Cfg::PGW, NSTEP = Cfg::NSTEP;
    constexpr int FRAME_PIX = Cfg::FRAME_PIX, C4 = CIN / 4, C8 = CIN / 8;
    const int tot4 = fpb * FRAME_PIX * C4, tot8 = fpb * FRAME_PIX * C8;
    const int ngrp = fpb * HOUT * HOUT / 16;
    const int npg = __builtin_amdgcn_readfirstlane((ngrp - (int)((threadIdx.x >> 6) % WP) + WP - 1) / WP);   // groups of this wavefront
    extern __shared__ float4 encs_smem4[];
    char* lds = reinterpret_cast<char*>(encs_smem4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave % WP, wc = wave / WP;
    const int j = lane & 15, q = lane >> 4;
    const gcpx_conv_src s = a.src[0];
    const int zero_off = fpb * FRAME_PIX * PIXB;            // byte offset of the zero pixel (behind the staged frames)
    float* wmax = reinterpret_cast<float*>(lds + CIN * 4);                          // the 4 per-wavefront maxima: pad bytes of pixel 0
    if (ZP && tid < PIXB / 4) reinterpret_cast<float*>(lds + zero_off)[tid] = 0.f;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s.scale) {
        sc4 = *reinterpret_cast<const float4*>(s.scale + 4 * (tid % C4));
        sh4 = *reinterpret_cast<const float4*>(s.shift + 4 * (tid % C4));
    }
    const int ew = a.w_split_log2_dev ? __builtin_amdgcn_readfirstlane(*a.w_split_log2_dev) : a.w_split_log2;

    f32x4 st1[CTW], st2[CTW];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) { st1[ct] = f32x4{0, 0, 0, 0}; st2[ct] = f32x4{0, 0, 0, 0}; }

    // k-step s of lane (j, q): tap and 8-channel group.  CIN = 16: taps 2 s + (q >> 1), group q & 1; CIN = 32: tap s, group q;
    // CIN = 64: tap s >> 1, group 4 (s & 1) + q.
    // per pixel group of this wavefront: byte offset of input pixel (2 oy - 1, 2 ox - 1), tap validity (bit ky: row, bit 4 + kx: column)
    int base[PGW], fl_[PGW], pin_[PGW];
    unsigned vmask[PGW];
#pragma unroll
    for (int pt = 0; pt < PGW; ++pt) {
        const int p = (wp + pt * WP) * 16 + j;
        const int fl = p / (HOUT * HOUT), pin = p % (HOUT * HOUT);
        const int oy = pin / HOUT, ox = pin % HOUT;
        fl_[pt] = fl; pin_[pt] = pin;
        base[pt] = ((fl * HIN + 2 * oy - 1) * HIN + 2 * ox - 1) * PIXB;
        unsigned m = 0;
        for (int k = 0; k < 4; ++k) {
            if (2 * oy - 1 + k >= 0 && 2 * oy - 1 + k < HIN) m |= 1u << k;
            if (2 * ox - 1 + k >= 0 && 2 * ox - 1 + k < HIN) m |= 16u << k;
        }
        vmask[pt] = m;
    }
    // packed pieces: [NSTEP][CT][2][64] x 16 B
    const char* wbase = reinterpret_cast<const char*>(a.wpk_split) + (size_t)(wc * CTW) * 2048 + lane * 16;

    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int f0 = blk * fpb;
        __syncthreads();                                    // the previous iteration's operand reads are done
        const float4* src = reinterpret_cast<const float4*>(s.ptr) + (size_t)f0 * FRAME_PIX * C4;
        const int nvalid4 = min(fpb, a.F - f0) * FRAME_PIX * C4;
        static_assert(256 % C4 == 0, "staging batches");
        float amax = 0.f;
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
                    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
                }
                if (idx < tot4) *reinterpret_cast<float4*>(lds + pix * PIXB + 16 * c4) = v[k];
            }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
        if (lane == 0) wmax[wave] = amax;
        __syncthreads();
        amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        int ex = 14 + 127 - (int)((__float_as_uint(amax) >> 23) & 0xff);            // amax 2^ex in [2^14, 2^15)
        ex = __builtin_amdgcn_readfirstlane(amax > 0.f ? max(-100, min(min(100, 126 - ew), ex)) : min(100, 126 - ew));
        const float sx2 = __uint_as_float((unsigned)(127 + ex) << 23);
        // in place: 8 consecutive channels of a pixel (32 B of f32) -> [8 first pieces | 8 second pieces]
        for (int i = tid; i < tot8; i += 256) {
            const int c8 = i % C8, pix = i / C8;
            char* p = lds + pix * PIXB + 32 * c8;
            const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 16);
            const float f[8] = {v0.x * sx2, v0.y * sx2, v0.z * sx2, v0.w * sx2, v1.x * sx2, v1.y * sx2, v1.z * sx2, v1.w * sx2};
            h8 p1, p2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                p1[e] = (_Float16)f[e];
                p2[e] = (_Float16)fmaf((float)p1[e], -1.f, f[e]);
            }
            *reinterpret_cast<h8*>(p) = p1;
            *reinterpret_cast<h8*>(p + 16) = p2;
        }
        __syncthreads();

        f32x4 acc[CTW][PGW];
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
            for (int pt = 0; pt < PGW; ++pt) acc[ct][pt] = f32x4{0, 0, 0, 0};
        auto fetch_w = [&](const int step, h8 (&w)[CTW][2]) __attribute__((always_inline)) {
            const char* wp_ = wbase + (size_t)step * CT * 2048;
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) {
                w[ct][0] = *reinterpret_cast<const h8*>(wp_ + ct * 2048);
                w[ct][1] = *reinterpret_cast<const h8*>(wp_ + ct * 2048 + 1024);
            }
        };
        auto mm = [&](const int step, const h8 (&w)[CTW][2]) __attribute__((always_inline)) {
            const int tap = CIN == 16 ? 2 * step + (q >> 1) : CIN == 32 ? step : step >> 1;
            const int grp = CIN == 16 ? (q & 1) : CIN == 32 ? q : 4 * (step & 1) + q;
            const int ky = tap >> 2, kx = tap & 3;
            const int tapoff = (ky * HIN + kx) * PIXB + grp * 32;
            h8 b[PGW][2];
#pragma unroll
            for (int pt = 0; pt < PGW; ++pt) {
                if (pt >= npg) break;                         // wave-uniform: pixel groups beyond the staged frames
                const bool ok = ((vmask[pt] >> ky) & 1u) && ((vmask[pt] >> (4 + kx)) & 1u);
                if constexpr (ZP) {
                    const char* p = lds + (ok ? base[pt] + tapoff : zero_off);
                    b[pt][0] = *reinterpret_cast<const h8*>(p);
                    b[pt][1] = *reinterpret_cast<const h8*>(p + 16);
                } else {
                    const char* p = lds + (ok ? base[pt] + tapoff : 0);
                    const h8 t0 = *reinterpret_cast<const h8*>(p), t1 = *reinterpret_cast<const h8*>(p + 16);
                    const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    b[pt][0] = ok ? t0 : z;
                    b[pt][1] = ok ? t1 : z;
                }
            }
            // small terms first: they are added to the accumulator while it is still small
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
                for (int pt = 0; pt < PGW; ++pt) {
                    if (pt >= npg) break;
                    acc[ct][pt] = mfma32h(w[ct][1], b[pt][0], acc[ct][pt]);
                }
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
                for (int pt = 0; pt < PGW; ++pt) {
                    if (pt >= npg) break;
                    acc[ct][pt] = mfma32h(w[ct][0], b[pt][1], acc[ct][pt]);
                }
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
                for (int pt = 0; pt < PGW; ++pt) {
                    if (pt >= npg) break;
                    acc[ct][pt] = mfma32h(w[ct][0], b[pt][0], acc[ct][pt]);
                }
        };
        h8 wA[CTW][2], wB[CTW][2];
        static_assert(NSTEP % 2 == 0, "ping-pong");
        fetch_w(0, wA);
#pragma unroll 1
        for (int step = 0; step < NSTEP; step += 2) {
            fetch_w(step + 1, wB);
            mm(step, wA);
            fetch_w(step + 2 < NSTEP ? step + 2 : step, wA);       // (unconditional: a static number of loads in flight)
            mm(step + 1, wB);
        }
        const float inv = __uint_as_float((unsigned)(127 - ex - ew) << 23);
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
                v[0] = fmaf(v[0], inv, bv.x); v[1] = fmaf(v[1], inv, bv.y); v[2] = fmaf(v[2], inv, bv.z); v[3] = fmaf(v[3], inv, bv.w);
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
        float* red = reinterpret_cast<float*>(lds);         // [WP][2][CT*16], the frames are no longer needed
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
int launch_enc_split(const gcpx_conv_args* a, hipStream_t stream, int nrows, int cus) {
    using Cfg = EncSplitCfg<CIN, COUT, HIN, FPB, WP, WC, ZP>;
    auto kern = conv4x4s2_split_kernel<CIN, COUT, HIN, FPB, WP, WC, ZP>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) {
            gcpx_set_error("conv4x4s2 (split): hipFuncSetAttribute(%d B): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        attr_set = true;
    }
    // Frames per workgroup.  With the MFMA time cut by 5x these layers are bound by the bytes a CU pulls (~10 B / cycle): per block
    // its frames in and out plus the filter bank once per wavefront COLUMN (wavefronts of a column read the same fragments at the
    // same time: they meet in L1) — so the cost of a choice is rounds x (frames x frame bytes + bank bytes), rounds = blocks on the
    // most loaded CU.  (The f32 kernel's rule — fewest frames on the most loaded CU — picks one-frame blocks for 1280 frames of
    // 16 x 16 x 32 and re-reads the 128 KB bank 5 times per CU: 76 us; three-frame blocks: see DESIGN.md.)
    int fpb = 1;
    double best = -1.0;
    const double frame_bytes = 4.0 * (HIN * HIN * CIN + Cfg::HOUT * Cfg::HOUT * COUT);
    const double bank_bytes = 4.0 * 16 * CIN * COUT * (4 / WC) / WP;
    for (int c = 1; c <= FPB; ++c) {
        const int nb = (a->F + c - 1) / c;
        const double cost = (double)((nb + cus - 1) / cus) * (c * frame_bytes + bank_bytes);
        if (best < 0 || cost <= best) { best = cost; fpb = c; }
    }
    if (const char* e = getenv("GCPX_ENC_FPB")) { const int v = atoi(e); if (v >= 1 && v <= FPB) fpb = v; }
    const int nblk = (a->F + fpb - 1) / fpb;
    const int grid = nblk < nrows ? nblk : nrows;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::lds_bytes(fpb), stream, *a, nblk, nrows, fpb);
    return GCPX_OK;
}

}  // namespace

// Called by gcpx_conv4x4s2 (conv_enc.hip) when the caller supplies split-f16 weights: 0 = launched, 1 = no split form for this shape
int gcpx_launch_enc_split(const gcpx_conv_args* a, hipStream_t stream, int nrows, int cus) {
    if (a->Hin != a->Win) return 1;
    if (a->Cin == 16 && a->Cout == 32 && a->Hin == 32) return launch_enc_split<16, 32, 32, 1, 2, 2, false>(a, stream, nrows, cus);
    if (a->Cin == 32 && a->Cout == 64 && a->Hin == 16) return launch_enc_split<32, 64, 16, 4, 1, 4, true>(a, stream, nrows, cus);
    if (a->Cin == 64 && a->Cout == 128 && a->Hin == 8) return launch_enc_split<64, 128, 8, 8, 1, 4, true>(a, stream, nrows, cus);
    if (a->Cin == 16 && a->Cout == 32 && a->Hin == 16) return launch_enc_split<16, 32, 16, 4, 2, 2, true>(a, stream, nrows, cus);
    if (a->Cin == 32 && a->Cout == 64 && a->Hin == 8) return launch_enc_split<32, 64, 8, 16, 1, 4, true>(a, stream, nrows, cus);
    return 1;
}
