// Decoder conv blocks on gfx950: (bilinear x2 upsample ->) 3x3 conv as an implicit GEMM on f32 MFMA.
//
// Replaces blox ConvDecoder blocks + gen_head as called through DecoderModule.decode_seq
// (/root/reference/gcp/prediction/models/tree/tree_dense_rec.py:42) — this build's spec of those blocks is in
// DESIGN.md "Model spec" (upsample = nn.Upsample(scale_factor=2, bilinear, align_corners=False), conv 3x3 pad 1).
//
// Work decomposition (one persistent 256-thread workgroup = 4 wavefronts):
//   * a tile is 256 output pixels (TH x TW pixels of TF frames); the (upsampled, channel-concatenated,
//     BatchNorm+LeakyReLU-applied) input region incl. the 1-pixel halo is staged ONCE in LDS per 32-channel chunk;
//   * each wavefront owns 4 pixel groups (16 pixels = the MFMA j side) x all CT output-channel groups
//     (16 channels = the MFMA i side); per (tap, 16 input channels) it issues one ds_read_b128 per pixel
//     group, one coalesced 1 KiB global load per channel group (weights pre-packed in fragment order, L2
//     resident) and 4*4*CT v_mfma_f32_16x16x4_f32;
//   * K is permuted so that lane (j, kk) consumes input channels 4*kk .. 4*kk+3 of its pixel — exactly one
//     16-byte LDS read — the packed weights use the same permutation.
//   * epilogue: bias, then either the raw NHWC store (+ per-workgroup BatchNorm partial sums) or, for the
//     output head, the discrete-logistic-mixture mean computed in registers (4-lane column shuffles).
#include "common.h"

#include <type_traits>

int gcpx_launch_up16_split(const gcpx_conv_args* a, hipStream_t stream, int grid);      // conv3x3_split.hip
int gcpx_launch_up32_split(const gcpx_conv_args* a, hipStream_t stream, int which, int grid);
int gcpx_launch_up16_fold(const gcpx_conv_args* a, hipStream_t stream, int grid);
int gcpx_launch_up16_fold16(const gcpx_conv_args* a, hipStream_t stream, int grid);
int gcpx_launch_wave_split(const gcpx_conv_args* a, hipStream_t stream, int ct, int depth);

namespace {

template <int TILE> struct TileShape;
template <> struct TileShape<0> { static constexpr int TH = 8, TW = 32, TF = 1; };
template <> struct TileShape<1> { static constexpr int TH = 16, TW = 16, TF = 1; };
template <> struct TileShape<2> { static constexpr int TH = 8, TW = 8, TF = 4; };
template <> struct TileShape<3> { static constexpr int TH = 8, TW = 16, TF = 1; };   // 128 pixels: more workgroups per CU

template <bool UP, int CC, int CT, int TILE>
struct ConvCfg {
    using TS = TileShape<TILE>;
    static constexpr int TH = TS::TH, TW = TS::TW, TF = TS::TF;
    static constexpr int RH = TH + 2, RW = TW + 2;          // staged hi-res region (halo 1)
    static constexpr int LH = TH / 2 + 2, LW = TW / 2 + 2;  // low-res patch feeding the upsample
    // padded channel pitch in LDS.  (A pitch of 40 floats makes the B-operand ds_read_b128 conflict-free — 36 is 2-way —
    // but measured no gain: these kernels are VALU+MFMA issue bound, not LDS bound.)
    static constexpr int CCP = (CC == 16) ? 24 : CC + 4;   // 16-channel chunks: pitch 24 (conflict-free b128 operand reads, as in the head kernel)
    static constexpr int C4 = CC / 4;
    static constexpr int HI_FLOATS = TF * RH * RW * CCP;
    static constexpr int RAW_FLOATS = UP ? TF * LH * LW * CC : 0;
    static constexpr int RED_FLOATS = 4 * 2 * CT * 16;      // cross-wave stats reduction
    static constexpr int LDS_BYTES = (HI_FLOATS + RAW_FLOATS) * 4;
    static constexpr int PR = TH * TW * TF / 64;            // pixel groups per wavefront (4 wavefronts)
    static constexpr int MIN_WAVES = (TILE == 3) ? 3 : 2;   // waves per SIMD the register allocation must allow
};

__device__ __forceinline__ float4 load_src4(const gcpx_conv_args& a, int f, int sy, int sx, int cglob) {
    const int c0 = a.src[0].C;
    const bool first = cglob < c0;
    const gcpx_conv_src& s = first ? a.src[0] : a.src[1];
    const int cl = first ? cglob : cglob - c0;
    const float* p = s.ptr + (((size_t)(f / s.frame_div) * a.Hin + sy) * a.Win + sx) * s.C + cl;
    float4 v = *reinterpret_cast<const float4*>(p);
    return affine_act4(v, s.scale, s.shift, cl, s.act);
}

// tanh via one v_exp + one v_rcp (coefficients of the mixture colour coupling; |err| ~1e-7)
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __expf(2.f * x);
    return 1.f - 2.f * __frcp_rn(e + 1.f);
}

template <bool UP, int CC, int CT, int TILE>
__global__ void __launch_bounds__(256, (ConvCfg<UP, CC, CT, TILE>::MIN_WAVES))
conv3x3_kernel(const gcpx_conv_args a, const int ntx, const int nty, const int ntiles) {
    using Cfg = ConvCfg<UP, CC, CT, TILE>;
    constexpr int PR = Cfg::PR;
    constexpr int TH = Cfg::TH, TW = Cfg::TW, TF = Cfg::TF, RH = Cfg::RH, RW = Cfg::RW;
    constexpr int LH = Cfg::LH, LW = Cfg::LW, CCP = Cfg::CCP, C4 = Cfg::C4;
    constexpr int NSTEP = 9 * (CC / 16);
    // staging slots (one float4 each) fetched from global per (tile, chunk): the low-res patch when upsampling,
    // the haloed region itself otherwise
    constexpr int SH = UP ? LH : RH, SW = UP ? LW : RW;
    constexpr int NSLOT = TF * SH * SW * C4;
    constexpr int NS = (NSLOT + 255) / 256;

    extern __shared__ float4 smem4[];
    float* hi = reinterpret_cast<float*>(smem4);
    float* raw = hi + Cfg::HI_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int nchunk = a.Cin / CC;
    const int Hout = a.Hout, Wout = a.Wout, F = a.F;

    // per-lane pixel of each of its 4 pixel groups, inside the tile
    int pixoff[PR], pfl[PR], py[PR], px[PR];
#pragma unroll
    for (int pt = 0; pt < PR; ++pt) {
        const int p = (wave * PR + pt) * 16 + j;
        pfl[pt] = p / (TH * TW);
        const int rem = p % (TH * TW);
        py[pt] = rem / TW;
        px[pt] = rem % TW;
        pixoff[pt] = ((pfl[pt] * RH + py[pt]) * RW + px[pt]) * CCP + q * 4;
    }
    // staging slot k of this thread: float4 index tid + 256*k of the staged region (decode = constant divisions)
    auto slot = [&](int k, int& c4, int& rx, int& ry, int& fl, int& lds) {
        const int idx = tid + 256 * k;
        c4 = idx % C4;
        int t = idx / C4;
        rx = t % SW; t /= SW;
        ry = t % SH;
        fl = (idx < NSLOT) ? t / SH : -1;               // -1: slot does not exist
        lds = UP ? ((fl * LH + ry) * LW + rx) * CC + c4 * 4 : ((fl * RH + ry) * RW + rx) * CCP + c4 * 4;
    };
    const float4* wbase = reinterpret_cast<const float4*>(a.wpk) + lane;

    // BatchNorm partial sums: only the upsampling blocks are followed by a norm
    f32x4 st1[UP ? CT : 1], st2[UP ? CT : 1];
#pragma unroll
    for (int ct = 0; ct < (UP ? CT : 1); ++ct) { st1[ct] = f32x4{0, 0, 0, 0}; st2[ct] = f32x4{0, 0, 0, 0}; }

    // ---- software pipeline over stages (tile, chunk): the global loads of stage s+1 are in flight while the
    //      MFMAs of stage s run; only the register -> LDS write (+ the LDS -> LDS upsample) is exposed ----
    float4 pre[NS];
    unsigned pre_ok = 0;
    auto tile_origin = [&](int tile, int& f0, int& y0, int& x0) {
        const int tx = tile % ntx;
        const int t2 = tile / ntx;
        f0 = (t2 / nty) * TF; y0 = (t2 % nty) * TH; x0 = tx * TW;
    };
    auto issue_loads = [&](int tile, int chunk) {
        int f0, y0, x0;
        tile_origin(tile, f0, y0, x0);
        pre_ok = 0;
        const int c0 = a.src[0].C;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            pre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            int s_c4, s_rx, s_ry, s_fl, s_lds;
            slot(k, s_c4, s_rx, s_ry, s_fl, s_lds);
            if (s_fl < 0) continue;
            int f = f0 + s_fl;
            int sy, sx;
            bool ok = f < F;
            if constexpr (!UP) {
                if (ok && a.src_row_map) { f = a.src_row_map[f]; ok = f >= 0; }
            }
            if constexpr (UP) {
                sy = min(max(y0 / 2 - 1 + s_ry, 0), a.Hin - 1);
                sx = min(max(x0 / 2 - 1 + s_rx, 0), a.Win - 1);
            } else {
                sy = y0 - 1 + s_ry;
                sx = x0 - 1 + s_rx;
                ok = ok && sy >= 0 && sy < Hout && sx >= 0 && sx < Wout;
            }
            if (ok) {
                const int cg = chunk * CC + s_c4 * 4;
                const bool first = cg < c0;
                const gcpx_conv_src& sr = first ? a.src[0] : a.src[1];
                const int cl = first ? cg : cg - c0;
                pre[k] = *reinterpret_cast<const float4*>(
                    sr.ptr + (((size_t)(f / sr.frame_div) * a.Hin + sy) * a.Win + sx) * sr.C + cl);
                pre_ok |= 1u << k;
            }
        }
    };
    auto write_stage = [&](int chunk) {
        const int c0 = a.src[0].C;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            int s_c4, s_rx, s_ry, s_fl, s_lds;
            slot(k, s_c4, s_rx, s_ry, s_fl, s_lds);
            if (s_fl < 0) continue;
            float4 v = pre[k];
            if (pre_ok & (1u << k)) {
                const int cg = chunk * CC + s_c4 * 4;
                const bool first = cg < c0;
                const gcpx_conv_src& sr = first ? a.src[0] : a.src[1];
                v = affine_act4(v, sr.scale, sr.shift, first ? cg : cg - c0, sr.act);
            }
            *reinterpret_cast<float4*>((UP ? raw : hi) + s_lds) = v;
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) issue_loads(tile, 0);
    for (; tile < ntiles; tile += gridDim.x) {
        int f0, y0, x0;
        tile_origin(tile, f0, y0, x0);

        f32x4 acc[CT][PR];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PR; ++pt) acc[ct][pt] = f32x4{0, 0, 0, 0};
        float4 wnext[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) wnext[ct] = wbase[ct * 64];

        // data-gradient launches: frames without a source row (tree nodes not matched to a ground-truth frame) are zero
        bool skip = false;
        if constexpr (!UP && TF == 1) skip = a.src_row_map != nullptr && a.src_row_map[f0] < 0;
        if (skip && tile + (int)gridDim.x < ntiles) issue_loads(tile + gridDim.x, 0);
        for (int chunk = 0; chunk < (skip ? 0 : nchunk); ++chunk) {
            // UP: the low-res patch `raw` is not read by the MFMA phase, so it can be written while other wavefronts
            // are still in the previous stage's MFMAs; one barrier then covers both "raw complete" and "hi free".
            if constexpr (!UP) __syncthreads();   // previous stage's LDS reads of `hi` are done
            write_stage(chunk);
            if constexpr (UP) {
                __syncthreads();
                // ---- bilinear x2 (align_corners=False) from the replicate-clamped low-res patch, LDS -> LDS ----
                // With even tile origins the source rows/weights depend only on the position inside the tile:
                //   hi row ry  <- raw rows (ry>>1, (ry>>1)+1) with weight (ry odd ? .75 : .25) on the lower one,
                // same along x.  (At the image border torch evaluates 1*in[0] + 0*in[1]; the clamped patch gives
                // .25*in[0] + .75*in[0]: identical up to one rounding.)  Thread map: c4 = tid & 7 (4 channels),
                // 32 positions per pass; no integer division or float index math in the loop.
                const int c4 = tid & (C4 - 1);
                const int p = tid >> 3;                                   // 0..31  (C4 == 8)
                static_assert(C4 == 8, "upsampling blocks stage 32-channel chunks");
                constexpr int RPP = 32 / (TW * TF);                       // hi rows covered per pass (1 or 2)
                static_assert(RPP == 1 || (RPP == 2 && TF == 1), "tile shapes: 8x32, 16x16, 8x8x4");
                const int fl = (TF > 1) ? p / TW : 0;
                const int rx = p % TW;                                    // TW is a power of two
                const int rsub = (RPP == 2) ? (p / TW) : 0;               // row inside the pass
                const float lx1 = (rx & 1) ? 0.75f : 0.25f, lx0w = 1.f - lx1;
                const float* rbase = raw + ((fl * LH) * LW + (rx >> 1)) * CC + c4 * 4;
                float* hbase = hi + ((fl * RH) * RW + rx) * CCP + c4 * 4;
                auto lerp4 = [&](const float* r, float wx1, float wx0, float wy1, float wy0) {
                    const float4 a00 = *reinterpret_cast<const float4*>(r);
                    const float4 a01 = *reinterpret_cast<const float4*>(r + CC);
                    const float4 a10 = *reinterpret_cast<const float4*>(r + LW * CC);
                    const float4 a11 = *reinterpret_cast<const float4*>(r + LW * CC + CC);
                    float4 v;
                    v.x = wy0 * (wx0 * a00.x + wx1 * a01.x) + wy1 * (wx0 * a10.x + wx1 * a11.x);
                    v.y = wy0 * (wx0 * a00.y + wx1 * a01.y) + wy1 * (wx0 * a10.y + wx1 * a11.y);
                    v.z = wy0 * (wx0 * a00.z + wx1 * a01.z) + wy1 * (wx0 * a10.z + wx1 * a11.z);
                    v.w = wy0 * (wx0 * a00.w + wx1 * a01.w) + wy1 * (wx0 * a10.w + wx1 * a11.w);
                    return v;
                };
                const bool top = (y0 == 0), bot = (y0 + TH == Hout), lft = (x0 == 0), rgt = (x0 + TW == Wout);
                // main part: columns 1..TW of the staged region (hi-res x0 .. x0+TW-1 -> region column rx+1)
#pragma unroll 2
                for (int k = 0; k < RH / RPP; ++k) {
                    const int ry = k * RPP + rsub;
                    // region column c = rx + 1 <-> hi-res X = x0 + rx: source columns ((rx+1)>>1, +1), weight by parity
                    const int cx = rx + 1;
                    const float wx1 = (cx & 1) ? 0.75f : 0.25f, wx0 = 1.f - wx1;
                    const float wy1 = (ry & 1) ? 0.75f : 0.25f, wy0 = 1.f - wy1;
                    const float* r = raw + ((fl * LH + (ry >> 1)) * LW + (cx >> 1)) * CC + c4 * 4;
                    float4 v = lerp4(r, wx1, wx0, wy1, wy0);
                    const bool zero = (top && ry == 0) || (bot && ry == RH - 1);
                    if (zero) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(hi + ((fl * RH + ry) * RW + cx) * CCP + c4 * 4) = v;
                }
                // the two halo columns (region columns 0 and RW-1)
                for (int e = p; e < TF * RH * 2; e += 32) {
                    const int side = e & 1;
                    const int ry = (e >> 1) % RH, f2 = (e >> 1) / RH;
                    const int cx = side ? RW - 1 : 0;
                    const float wx1 = (cx & 1) ? 0.75f : 0.25f, wx0 = 1.f - wx1;
                    const float wy1 = (ry & 1) ? 0.75f : 0.25f, wy0 = 1.f - wy1;
                    const float* r = raw + ((f2 * LH + (ry >> 1)) * LW + (cx >> 1)) * CC + c4 * 4;
                    float4 v = lerp4(r, wx1, wx0, wy1, wy0);
                    const bool zero = (top && ry == 0) || (bot && ry == RH - 1) || (lft && side == 0) || (rgt && side == 1);
                    if (zero) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(hi + ((f2 * RH + ry) * RW + cx) * CCP + c4 * 4) = v;
                }
                (void)lx1; (void)lx0w; (void)rbase; (void)hbase;
            }
            __syncthreads();
            // next stage's global loads go out now and land while the MFMAs below run
            if (chunk + 1 < nchunk) issue_loads(tile, chunk + 1);
            else if (tile + (int)gridDim.x < ntiles) issue_loads(tile + gridDim.x, 0);

            // ---- MFMA main loop over (tap, 16-channel group) ----
            // Weights are double-buffered in registers one step ahead; the stream of steps is contiguous across
            // chunks and the packed buffer carries one zero step of padding, so the prefetch never branches.
            const float4* wp = wbase + (size_t)chunk * NSTEP * CT * 64;
            __builtin_amdgcn_s_setprio(1);      // MFMA phase outranks the staging VALU of co-resident workgroups
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int tapoff = ((tap / 3) * RW + (tap % 3)) * CCP;
#pragma unroll
                for (int cgl = 0; cgl < CC / 16; ++cgl) {
                    const int st = tap * (CC / 16) + cgl;
                    float4 wcur[CT];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) wcur[ct] = wnext[ct];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) wnext[ct] = wp[((st + 1) * CT + ct) * 64];
                    float4 b[PR];
#pragma unroll
                    for (int pt = 0; pt < PR; ++pt)
                        b[pt] = *reinterpret_cast<const float4*>(hi + pixoff[pt] + tapoff + cgl * 16);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                        for (int pt = 0; pt < PR; ++pt) {
                            acc[ct][pt] = mfma16(wcur[ct].x, b[pt].x, acc[ct][pt]);
                            acc[ct][pt] = mfma16(wcur[ct].y, b[pt].y, acc[ct][pt]);
                            acc[ct][pt] = mfma16(wcur[ct].z, b[pt].z, acc[ct][pt]);
                            acc[ct][pt] = mfma16(wcur[ct].w, b[pt].w, acc[ct][pt]);
                        }
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }

        // ---- epilogue ----
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float4 bv = *reinterpret_cast<const float4*>(a.bias + ct * 16 + q * 4);
#pragma unroll
            for (int pt = 0; pt < PR; ++pt) {
                acc[ct][pt][0] += bv.x; acc[ct][pt][1] += bv.y; acc[ct][pt][2] += bv.z; acc[ct][pt][3] += bv.w;
            }
        }
        const int mode = a.head_mode;
        if (mode == GCPX_HEAD_RAW || mode == GCPX_HEAD_DLM_BOTH) {
#pragma unroll
            for (int pt = 0; pt < PR; ++pt) {
                const int f = f0 + pfl[pt];
                if (f >= F) continue;
                float* op = a.out + (((size_t)f * Hout + (y0 + py[pt])) * Wout + (x0 + px[pt])) * a.out_pitch;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int c = ct * 16 + q * 4;
                    if (c < a.out_pitch) {
                        f32x4 v = acc[ct][pt];
                        if (a.out_act == GCPX_ACT_LRELU) {
                            v[0] = lrelu(v[0], 0.2f); v[1] = lrelu(v[1], 0.2f); v[2] = lrelu(v[2], 0.2f); v[3] = lrelu(v[3], 0.2f);
                        }
                        *reinterpret_cast<float4*>(op + c) = make_float4(v[0], v[1], v[2], v[3]);
                        if constexpr (UP) {
                            if (a.stats_partial) {
                                st1[ct] += v;
                                st2[ct] += v * v;
                            }
                        }
                    }
                }
            }
        }
        if constexpr (CT >= 5) {
            if (mode == GCPX_HEAD_DLM_MEAN || mode == GCPX_HEAD_DLM_BOTH) {
                // kernel channel order: slot 8k..8k+7 = {logit_k, mu_r, mu_g, mu_b, c0, c1, c2, pad}, k = 0..9;
                // lanes with even q hold the first half of mixture 2*ct + q/2, lane+16 holds the second half.
#pragma unroll
                for (int pt = 0; pt < PR; ++pt) {
                    float lg[5], mr[5], mg[5], mb[5];
#pragma unroll
                    for (int ct = 0; ct < 5; ++ct) {
                        const f32x4 v = acc[ct][pt];
                        const float o0 = __shfl_xor(v[0], 16), o1 = __shfl_xor(v[1], 16), o2 = __shfl_xor(v[2], 16);
                        const float c0 = fast_tanh(o0), c1 = fast_tanh(o1), c2 = fast_tanh(o2);
                        lg[ct] = v[0];
                        mr[ct] = v[1];
                        mg[ct] = v[2] + c0 * mr[ct];
                        mb[ct] = v[3] + c1 * mr[ct] + c2 * mg[ct];
                    }
                    float m = lg[0];
#pragma unroll
                    for (int ct = 1; ct < 5; ++ct) m = fmaxf(m, lg[ct]);
                    m = fmaxf(m, __shfl_xor(m, 32));
                    float S = 0.f, Sr = 0.f, Sg = 0.f, Sb = 0.f;
#pragma unroll
                    for (int ct = 0; ct < 5; ++ct) {
                        const float w = __expf(lg[ct] - m);
                        S += w; Sr += w * mr[ct]; Sg += w * mg[ct]; Sb += w * mb[ct];
                    }
                    S += __shfl_xor(S, 32); Sr += __shfl_xor(Sr, 32); Sg += __shfl_xor(Sg, 32); Sb += __shfl_xor(Sb, 32);
                    const int f = f0 + pfl[pt];
                    if (q == 0 && f < F) {
                        const float inv = 1.f / S;
                        const size_t plane = (size_t)Hout * Wout;
                        float* ip = a.images + (size_t)f * 3 * plane + (size_t)(y0 + py[pt]) * Wout + (x0 + px[pt]);
                        ip[0] = fminf(fmaxf(Sr * inv, -1.f), 1.f);
                        ip[plane] = fminf(fmaxf(Sg * inv, -1.f), 1.f);
                        ip[2 * plane] = fminf(fmaxf(Sb * inv, -1.f), 1.f);
                    }
                }
            }
        }
        if (mode == GCPX_HEAD_TANH_NCHW) {
#pragma unroll
            for (int pt = 0; pt < PR; ++pt) {
                const int f = f0 + pfl[pt];
                if (q == 0 && f < F) {
                    const size_t plane = (size_t)Hout * Wout;
                    float* ip = a.images + (size_t)f * 3 * plane + (size_t)(y0 + py[pt]) * Wout + (x0 + px[pt]);
                    ip[0] = tanhf(acc[0][pt][0]);
                    ip[plane] = tanhf(acc[0][pt][1]);
                    ip[2 * plane] = tanhf(acc[0][pt][2]);
                }
            }
        }
    }

    // ---- per-workgroup BatchNorm partial sums (deterministic: no atomics) ----
    if (UP && a.stats_partial) {
        __syncthreads();
        float* red = hi;   // [4 waves][2][CT*16]
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
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * CT * 16 + c];
            a.stats_partial[((size_t)blockIdx.x * 2 + which) * CT * 16 + c] = s;
        }
    }
}

// -----------------------------------------------------------------------------------------------------------
// Output head (3x3 conv, 16 input channels, no upsample): wave-autonomous variant.
//   * the packed weights (9 steps x CT KiB) are loaded into LDS once per 512-thread workgroup (1 per CU);
//   * every wavefront then works alone on items of 4 rows x 16 pixels: it stages its own haloed 6 x 18 x 16ch
//     region in a wave-private LDS buffer (global loads prefetched in registers during the previous item's MFMAs),
//     runs 9 x 4 x CT x 4 MFMAs with weights and activations both read from LDS, and finishes the mixture mean in
//     registers.  No barrier in steady state: the 8 wavefronts of a CU drift apart, so one wave's staging / epilogue
//     VALU work overlaps the other waves' MFMAs on the same SIMD.
// -----------------------------------------------------------------------------------------------------------
//   * REM: the output channels are CT full 16-channel MFMA tiles plus a 4-channel remainder (the 100-channel mixture head =
//     6 x 16 + 4).  The remainder runs on v_mfma_f32_4x4x1_16b_f32 — 16 independent 4x4 outer products per instruction:
//     A = the 4 remainder channels (same in every block), B = one pixel per lane — i.e. 4 channels x 64 pixels x 1 k at the
//     same FLOP rate as the 16x16x4 tile, so no MFMA cycle is spent on padded channels (a 7th full tile would waste 12 / 16).
//     Its weights sit in the storage of weight tile CT: float4 index kg * 4 + c holds W[16 CT + c][k = 4 kg .. 4 kg + 3].
template <int CT, bool REM>
struct HeadCfg {
    static constexpr int RW = 18, RH = 6, CCP = 24;        // item = 4 rows x 16 pixels (+ halo): 1.69x staged / output pixel;
                                                             // pitch 24 floats: conflict-free ds_read_b128 of the B operand
    static constexpr int REGION_FLOATS = RH * RW * CCP;                 // 2592 floats per wave
    static constexpr int WT = CT + (REM ? 1 : 0);                       // weight tiles per tap in LDS
    static constexpr int W_FLOAT4 = 9 * WT * 64;
    static constexpr int LDS_BYTES = W_FLOAT4 * 16 + 8 * REGION_FLOATS * 4;
    static constexpr int NS = (RH * RW * 4 + 63) / 64;                   // float4 slots per lane
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float f4get(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

template <int CT, bool REM>
__global__ void __launch_bounds__(512, 2) conv3x3_head_kernel(const gcpx_conv_args a, const int items_per_wave,
                                                              const int nitems) {
    using Cfg = HeadCfg<CT, REM>;
    constexpr int RW = Cfg::RW, RH = Cfg::RH, CCP = Cfg::CCP, NS = Cfg::NS, WT = Cfg::WT;
    extern __shared__ float4 smem4[];
    float4* wl = smem4;                                                   // [9][CT][64] float4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-level indices in SGPRs (SALU, not VALU)
    float* reg = reinterpret_cast<float*>(smem4 + Cfg::W_FLOAT4) + wave * Cfg::REGION_FLOATS;
    const int j = lane & 15, q = lane >> 4;
    const int H = a.Hout, W = a.Wout, F = a.F;
    const int ncb = W / 16, nrp = H / 4;

    for (int i = tid; i < Cfg::W_FLOAT4; i += 512) wl[i] = reinterpret_cast<const float4*>(a.wpk)[i];
    __syncthreads();

    int pixoff[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) pixoff[pt] = (pt * RW + j) * CCP + q * 4;
    const int rem_pix = (q * RW + j) * CCP;               // remainder: lane = pixel (row q, column j) of the item
    const gcpx_conv_src sr = a.src[0];

    const int gw = blockIdx.x * 8 + wave;
    int item = gw * items_per_wave;
    const int item_end = min(item + items_per_wave, nitems);

    float4 pre[NS];
    unsigned pre_ok = 0;
    // items of a frame are walked column-block-major (top to bottom inside a 16-pixel column block): the two halo rows an item
    // shares with the one below are re-read by the very next item of the same wavefront (cache hit) — strip-major order put them
    // 4 items apart, beyond the L1 and, summed over a die, the L2
    auto origin = [&](int it, int& f, int& y0, int& x0) {
        const int strip = it % nrp;
        const int t = it / nrp;
        y0 = strip * 4; f = t / ncb; x0 = (t % ncb) * 16;
    };
    int s_ry[NS], s_rx[NS];                               // staging slot -> (row, col) of the region, tile independent
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int t = (lane + 64 * k) >> 2;
        s_rx[k] = t % RW;
        s_ry[k] = t / RW;
    }
    auto issue_loads = [&](int it) {
        int f, y0, x0;
        origin(it, f, y0, x0);
        pre_ok = 0;
        const float* base = sr.ptr + (size_t)f * H * W * 16;                  // wave-uniform (scalar)
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            pre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int idx = lane + 64 * k;
            const int sy = y0 - 1 + s_ry[k], sx = x0 - 1 + s_rx[k];
            if (idx < RH * RW * 4 && sy >= 0 && sy < H && sx >= 0 && sx < W) {
                pre[k] = *reinterpret_cast<const float4*>(base + (unsigned)((__umul24(sy, W) + sx) * 16 + (idx & 3) * 4));
                pre_ok |= 1u << k;
            }
        }
    };
    if (item < item_end) issue_loads(item);

    for (; item < item_end; ++item) {
        int f, y0, x0;
        origin(item, f, y0, x0);
        // registers -> wave-private LDS region (BatchNorm affine + LeakyReLU of the producer applied here; the
        // zero padding of the conv stays exactly zero)
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = lane + 64 * k;
            if (idx < RH * RW * 4) {
                float4 v = pre[k];
                if (pre_ok & (1u << k)) v = affine_act4(v, sr.scale, sr.shift, (idx & 3) * 4, sr.act);
                *reinterpret_cast<float4*>(reg + (idx >> 2) * CCP + (idx & 3) * 4) = v;
            }
        }
        if (item + 1 < item_end) issue_loads(item + 1);

        f32x4 acc[CT][4];
        f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
        // one (tap) step: 4 ds_read_b128 of activations + CT of weights, 4*4*CT MFMAs.  The first step starts the
        // accumulators from the inline constant 0 (no v_mov zero-fill: VALU issue slots are as scarce as MFMA slots,
        // f32 MFMA and VALU do not overlap on a SIMD).
        // activations for the NEXT tap are fetched while the current tap's MFMAs run (the LDS round trip at every step
        // boundary was an un-overlapped bubble whenever the partner wavefront was not in its MFMA phase)
        float4 bnext[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) bnext[pt] = *reinterpret_cast<const float4*>(reg + pixoff[pt]);
        auto step = [&](const int tap, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            float4 b[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) b[pt] = bnext[pt];
            const float4* wp = wl + tap * WT * 64 + lane;
            float4 w[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) w[ct] = wp[ct * 64];
            [[maybe_unused]] float4 rw[4], rb[4];
            if constexpr (REM) {
                const int tapoff0 = ((tap / 3) * RW + (tap % 3)) * CCP;
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    rw[kg] = wl[(tap * WT + CT) * 64 + kg * 4 + (lane & 3)];
                    rb[kg] = *reinterpret_cast<const float4*>(reg + rem_pix + tapoff0 + kg * 4);
                }
            }
            {   // tap + 1 (for tap == 8 this reads one row past the wave's region: in-range LDS, value unused)
                const int t1 = tap + 1;
                const int tapoff = ((t1 / 3) * RW + (t1 % 3)) * CCP;
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) bnext[pt] = *reinterpret_cast<const float4*>(reg + pixoff[pt] + tapoff);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    if constexpr (FIRST) acc[ct][pt] = mfma16(w[ct].x, b[pt].x, f32x4{0, 0, 0, 0});
                    else acc[ct][pt] = mfma16(w[ct].x, b[pt].x, acc[ct][pt]);
                    acc[ct][pt] = mfma16(w[ct].y, b[pt].y, acc[ct][pt]);
                    acc[ct][pt] = mfma16(w[ct].z, b[pt].z, acc[ct][pt]);
                    acc[ct][pt] = mfma16(w[ct].w, b[pt].w, acc[ct][pt]);
                }
            }
            if constexpr (REM) {
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    acc4 = mfma4(rw[kg].x, rb[kg].x, acc4);
                    acc4 = mfma4(rw[kg].y, rb[kg].y, acc4);
                    acc4 = mfma4(rw[kg].z, rb[kg].z, acc4);
                    acc4 = mfma4(rw[kg].w, rb[kg].w, acc4);
                }
            }
        };
        // the wavefront that is in its MFMA phase outranks its SIMD partner's staging / epilogue VALU work
        __builtin_amdgcn_s_setprio(1);
        step(0, std::true_type{});
        step(1, std::false_type{});
        step(2, std::false_type{});
#pragma unroll 1
        for (int tap = 3; tap < 9; tap += 3) {
            step(tap, std::false_type{});
            step(tap + 1, std::false_type{});
            step(tap + 2, std::false_type{});
        }
        __builtin_amdgcn_s_setprio(0);

        // ---- epilogue ----
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float4 bv = *reinterpret_cast<const float4*>(a.bias + ct * 16 + q * 4);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                acc[ct][pt][0] += bv.x; acc[ct][pt][1] += bv.y; acc[ct][pt][2] += bv.z; acc[ct][pt][3] += bv.w;
            }
        }
        const int mode = a.head_mode;
        const size_t plane = (size_t)H * W;
        const int orow = a.raw_row_map ? a.raw_row_map[f] : f;
        if ((mode == GCPX_HEAD_RAW || mode == GCPX_HEAD_DLM_BOTH) && orow >= 0) {
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                float* op = a.out + (((size_t)orow * H + (y0 + pt)) * W + (x0 + j)) * a.out_pitch;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int c = ct * 16 + q * 4;
                    if (c < a.out_pitch) {
                        const f32x4 v = acc[ct][pt];
                        *reinterpret_cast<float4*>(op + c) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
            if constexpr (REM) {       // lane = pixel (row q, column j): its 4 remainder channels
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + CT * 16);
                float* op = a.out + (((size_t)orow * H + (y0 + q)) * W + (x0 + j)) * a.out_pitch + CT * 16;
                *reinterpret_cast<float4*>(op) = make_float4(acc4[0] + bv.x, acc4[1] + bv.y, acc4[2] + bv.z, acc4[3] + bv.w);
            }
        }
        if constexpr (CT >= 5) {
            if (mode == GCPX_HEAD_DLM_MEAN || mode == GCPX_HEAD_DLM_BOTH) {
                // kernel channel order: slot 8k..8k+7 = {logit_k, mu_r, mu_g, mu_b, c0, c1, c2, pad}, k = 0..9: in tile ct the
                // even 16-lane rows (q = 0, 2) hold the first half of mixtures 2ct, 2ct+1 and the odd rows (q = 1, 3) the
                // coefficient half.  One v_permlane16_swap per register exchanges "my odd-row half of pixel group s" with
                // "the partner's even-row half of pixel group s+2": afterwards EVERY lane owns both halves of one pixel —
                // even rows pixel groups 0/1, odd rows pixel groups 2/3 — so the mixture math runs on all 64 lanes for two
                // pixel groups instead of on 32 lanes for four.
#pragma unroll
                for (int ct = 0; ct < 5; ++ct)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[ct][s2][r]),
                                                                             __float_as_uint(acc[ct][s2 + 2][r]), false, false);
                            acc[ct][s2][r] = __uint_as_float(sw[0]);
                            acc[ct][s2 + 2][r] = __uint_as_float(sw[1]);
                        }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float lg[5], mr[5], mg[5], mb[5];
#pragma unroll
                    for (int ct = 0; ct < 5; ++ct) {
                        const f32x4 e = acc[ct][s2], o = acc[ct][s2 + 2];
                        const float c0 = fast_tanh(o[0]), c1 = fast_tanh(o[1]), c2 = fast_tanh(o[2]);
                        lg[ct] = e[0];
                        mr[ct] = e[1];
                        mg[ct] = e[2] + c0 * mr[ct];
                        mb[ct] = e[3] + c1 * mr[ct] + c2 * mg[ct];
                    }
                    float m = lg[0];
#pragma unroll
                    for (int ct = 1; ct < 5; ++ct) m = fmaxf(m, lg[ct]);
                    m = fmaxf(m, __shfl_xor(m, 32));
                    float S = 0.f, Sr = 0.f, Sg = 0.f, Sb = 0.f;
#pragma unroll
                    for (int ct = 0; ct < 5; ++ct) {
                        const float w = __expf(lg[ct] - m);
                        S += w; Sr += w * mr[ct]; Sg += w * mg[ct]; Sb += w * mb[ct];
                    }
                    S += __shfl_xor(S, 32); Sr += __shfl_xor(Sr, 32); Sg += __shfl_xor(Sg, 32); Sb += __shfl_xor(Sb, 32);
                    if (q < 2) {
                        const int pt = s2 + 2 * q;        // q = 0: pixel groups 0, 1 ; q = 1: pixel groups 2, 3
                        const float inv = 1.f / S;
                        float* ip = a.images + (size_t)f * 3 * plane + (size_t)(y0 + pt) * W + (x0 + j);
                        ip[0] = fminf(fmaxf(Sr * inv, -1.f), 1.f);
                        ip[plane] = fminf(fmaxf(Sg * inv, -1.f), 1.f);
                        ip[2 * plane] = fminf(fmaxf(Sb * inv, -1.f), 1.f);
                    }
                }
            }
        }
        if (mode == GCPX_HEAD_TANH_NCHW) {
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                if (q == 0) {
                    float* ip = a.images + (size_t)f * 3 * plane + (size_t)(y0 + pt) * W + (x0 + j);
                    ip[0] = tanhf(acc[0][pt][0]);
                    ip[plane] = tanhf(acc[0][pt][1]);
                    ip[2 * plane] = tanhf(acc[0][pt][2]);
                }
            }
        }
    }
}

template <int CT, bool REM>
int launch_head(const gcpx_conv_args* a, hipStream_t stream) {
    using Cfg = HeadCfg<CT, REM>;
    auto kern = conv3x3_head_kernel<CT, REM>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) {
            gcpx_set_error("conv3x3 head: hipFuncSetAttribute(%d B LDS): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        attr_set = true;
    }
    const int nitems = a->F * (a->Hout / 4) * (a->Wout / 16);
    int grid = gcpx_conv_grid() / 2;                     // one 512-thread workgroup per CU
    if (grid * 8 > nitems) grid = (nitems + 7) / 8;
    const int ipw = (nitems + grid * 8 - 1) / (grid * 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), Cfg::LDS_BYTES, stream, *a, ipw, nitems);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

// -----------------------------------------------------------------------------------------------------------
// Plain 3x3 conv at full resolution with several 16-channel input chunks (the data gradients of the output head,
// 112 -> 16, and of the 16-channel decoder blocks, 16 -> 32): the wave-autonomous scheme of the head kernel with a
// chunk loop inside the item.  All chunks' weights stay in LDS; a wavefront stages one 6 x 18 x 16ch region per chunk
// (the next chunk's / next item's global loads are in flight during the current chunk's MFMAs) and keeps the CT x 4
// accumulator tiles in registers across chunks.  Frames whose src_row_map entry is negative are zero-filled
// without touching the source (nodes that are not matched to a ground-truth frame carry no loss gradient).
// -----------------------------------------------------------------------------------------------------------
template <int CT>
struct WaveCfg {
    static constexpr int RW = 18, RH = 6, CCP = 24;
    static constexpr int REGION_FLOATS = RH * RW * CCP;
    static constexpr int NS = (RH * RW * 4 + 63) / 64;
    // + one region row: the tap-ahead operand fetch of the last tap reads one row past the last wave's region
    static int lds_bytes(int nchunk) { return nchunk * 9 * CT * 64 * 16 + 8 * REGION_FLOATS * 4 + RW * CCP * 4; }
};

template <int CT, int DEPTH>
__global__ void __launch_bounds__(512, 2) conv3x3_wave_kernel(const gcpx_conv_args a, const int nitems) {
    using Cfg = WaveCfg<CT>;
    constexpr int RW = Cfg::RW, RH = Cfg::RH, CCP = Cfg::CCP, NS = Cfg::NS;
    extern __shared__ float4 smem4[];
    const int nchunk = a.Cin / 16;
    const int wfloat4 = nchunk * 9 * CT * 64;
    float4* wl = smem4;                                                   // [nchunk][9][CT][64] float4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* reg = reinterpret_cast<float*>(smem4 + wfloat4) + wave * Cfg::REGION_FLOATS;
    const int j = lane & 15, q = lane >> 4;
    const int H = a.Hout, W = a.Wout;
    const int ncb = W / 16, nrp = H / 4, ipf = ncb * nrp;
    const gcpx_conv_src sr = a.src[0];
    const int Cs = sr.C;                                                  // channel pitch of the source

    for (int i = tid; i < wfloat4; i += 512) wl[i] = reinterpret_cast<const float4*>(a.wpk)[i];
    // frames without a source row get zeros; the main loop below only walks the rows that exist
    if (a.src_row_frames) {
        const int f4_per_frame = H * W * a.out_pitch / 4;
        for (int f = blockIdx.x; f < a.F; f += gridDim.x) {
            if (a.src_row_map[f] >= 0) continue;
            float4* op = reinterpret_cast<float4*>(a.out + (size_t)f * H * W * a.out_pitch);
            for (int i = tid; i < f4_per_frame; i += 512) op[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();

    const float* bbase = reg + j * CCP + q * 4;           // B operand of (region row r, tap column dx): + (r * RW + dx) * CCP

    // items are dealt round-robin over all wavefronts of the grid (consecutive workgroup ids sit on different XCDs: renumbered so
    // that an XCD owns a contiguous run of items per round and the halo rows shared by vertically adjacent items are L2 hits)
    const int stride = gridDim.x * 8;
    const int lb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
    const int first = lb * 8 + wave;
    const int nmine = first < nitems ? (nitems - first + stride - 1) / stride : 0;
    const int nsteps = nmine * nchunk;

    int s_ry[NS], s_rx[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int t = (lane + 64 * k) >> 2;
        s_rx[k] = t % RW;
        s_ry[k] = t / RW;
    }
    // item k of this wavefront -> (source row = index of the active frame, tile origin)
    auto geom = [&](int k, int& srow, int& y0, int& x0) {
        const int it = first + k * stride;
        srow = it / ipf;
        const int rem = it - srow * ipf;
        y0 = (rem / ncb) * 4; x0 = (rem % ncb) * 16;
    };

    // ---- prefetch cursor: DEPTH (item, chunk) steps ahead of the compute cursor; per-item slot offsets computed once per item ----
    int pk = 0, pchunk = 0;                               // step the next issue() will load
    unsigned poff[NS];                                    // byte offset of the slot inside the source frame; outside the image: beyond the buffer
    unsigned pmask = 0;                                   // slots inside the image
    const float* pbase = sr.ptr;
    const unsigned frame_bytes = (unsigned)H * W * Cs * 4;
    // a source row that no frame reads (src_row_frames[row] < 0: padded time steps) is skipped: no loads, no MFMAs, no store.  The
    // entry of the NEXT item is requested while the current one is set up, so the lookup never sits in front of a prefetch.
    int pf_next_v = 0;
    auto frame_of_item = [&](int k) { return a.src_row_frames[(first + k * stride) / ipf]; };
    if (a.src_row_frames && nmine > 0) pf_next_v = frame_of_item(0);
    auto enter_item = [&]() {                             // (pk) -> poff, pbase
        int srow, y0, x0;
        geom(pk, srow, y0, x0);
        pbase = sr.ptr + (size_t)srow * H * W * Cs;
        bool live = true;
        if (a.src_row_frames) {
            live = __builtin_amdgcn_readfirstlane(pf_next_v) >= 0;
            if (pk + 1 < nmine) pf_next_v = frame_of_item(pk + 1);
        }
        pmask = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = lane + 64 * k;
            const int sy = y0 - 1 + s_ry[k], sx = x0 - 1 + s_rx[k];
            const bool ok = live && idx < RH * RW * 4 && sy >= 0 && sy < H && sx >= 0 && sx < W;
            poff[k] = ok ? (unsigned)((__umul24(sy, W) + sx) * Cs + (idx & 3) * 4) * 4u : 0x80000000u;
            pmask |= ok ? 1u << k : 0u;
        }
    };
    // the staging loads are buffer loads against a per-frame descriptor: a slot outside the image carries an offset beyond the
    // buffer and the hardware returns zeros — no per-slot branch, no zero-fill of the destination registers
    auto issue = [&](float4 (&pre)[NS], unsigned& ok) {
        if (pk >= nmine) return;
        if (pchunk == 0) enter_item();
        const unsigned long long pb = reinterpret_cast<unsigned long long>(pbase);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pb), hi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, frame_bytes, 0x00020000);
        const int soff = pchunk * 64;
        ok = pmask;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, poff[k], soff, 0);
            pre[k] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
        if (++pchunk == nchunk) { pchunk = 0; ++pk; }
    };

    f32x4 acc[CT][4];
    int ck = 0, cchunk = 0;                               // compute cursor
    int f_v = 0;                                          // output frame of the compute item: requested at its first chunk, used at its last
    auto step = [&](float4 (&pre)[NS], unsigned& ok) {
        if (__builtin_amdgcn_readfirstlane(ok) == 0) {    // skipped item (lane 0 of a live item always has slots inside the image)
            issue(pre, ok);
            if (++cchunk == nchunk) { cchunk = 0; ++ck; }
            return;
        }
        // registers -> wave-private LDS region (producer's affine + activation applied here; conv zero padding stays zero)
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = lane + 64 * k;
            if (idx < RH * RW * 4) {
                float4 v = pre[k];
                if (ok & (1u << k)) v = affine_act4(v, sr.scale, sr.shift, cchunk * 16 + (idx & 3) * 4, sr.act);
                *reinterpret_cast<float4*>(reg + (idx >> 2) * CCP + (idx & 3) * 4) = v;
            }
        }
        issue(pre, ok);                                   // this register set is free again: load the step DEPTH ahead
        if (cchunk == 0) {
            if (a.src_row_frames) f_v = a.src_row_frames[(first + ck * stride) / ipf];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // operands: the 9 x CT weight fragments of the chunk and the 6 x 3 distinct activation fragments (tap (dy, dx) of pixel row
        // pt reads region row pt + dy: 18 loads serve 36 operand uses), requested three region rows ahead of their MFMAs in three
        // rotating row buffers; the sched_barriers keep the compiler from sinking the loads next to their uses
        const float4* wc = wl + cchunk * 9 * CT * 64 + lane;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {                 // one 16-channel output tile at a time: 9 weight + 3 x 3 activation fragments live
            float4 w[9], b0[3], b1[3], b2[3];
            auto ldrow = [&](int r, float4 (&b)[3]) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) b[dx] = *reinterpret_cast<const float4*>(bbase + (r * RW + dx) * CCP);
            };
            auto phase = [&](const int r, const float4 (&b)[3]) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int pt = r - dy;
                            if (pt < 0 || pt > 3) continue;
                            acc[ct][pt] = mfma16(f4get(w[dy * 3 + dx], kk), f4get(b[dx], kk), acc[ct][pt]);
                        }
                    }
                }
            };
#pragma unroll
            for (int t = 0; t < 3; ++t) w[t] = wc[(t * CT + ct) * 64];
            ldrow(0, b0);
#pragma unroll
            for (int t = 3; t < 9; ++t) w[t] = wc[(t * CT + ct) * 64];
            ldrow(1, b1);
            ldrow(2, b2);
            __builtin_amdgcn_sched_barrier(0);
            phase(0, b0);
            __builtin_amdgcn_sched_barrier(0);
            ldrow(3, b0);
            __builtin_amdgcn_sched_barrier(0);
            phase(1, b1);
            __builtin_amdgcn_sched_barrier(0);
            ldrow(4, b1);
            __builtin_amdgcn_sched_barrier(0);
            phase(2, b2);
            __builtin_amdgcn_sched_barrier(0);
            ldrow(5, b2);
            __builtin_amdgcn_sched_barrier(0);
            phase(3, b0);
            phase(4, b1);
            phase(5, b2);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);

        if (++cchunk == nchunk) {
            int srow, y0, x0;
            geom(ck, srow, y0, x0);
            const int f = a.src_row_frames ? __builtin_amdgcn_readfirstlane(f_v) : srow;
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                float* op = a.out + (((size_t)f * H + (y0 + pt)) * W + (x0 + j)) * a.out_pitch;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float4 bs = *reinterpret_cast<const float4*>(a.bias + ct * 16 + q * 4);
                    const f32x4 v = acc[ct][pt];
                    *reinterpret_cast<float4*>(op + ct * 16 + q * 4) = make_float4(v[0] + bs.x, v[1] + bs.y, v[2] + bs.z, v[3] + bs.w);
                }
            }
            cchunk = 0; ++ck;
        }
    };

    float4 preA[NS];
    unsigned okA = 0;
    issue(preA, okA);
    if constexpr (DEPTH == 2) {
        float4 preB[NS];
        unsigned okB = 0;
        issue(preB, okB);
        for (int s = 0; s < nsteps; s += 2) {
            step(preA, okA);
            if (s + 1 < nsteps) step(preB, okB);
        }
    } else {
        for (int s = 0; s < nsteps; ++s) step(preA, okA);
    }
}

template <int CT, int DEPTH>
int launch_wave(const gcpx_conv_args* a, hipStream_t stream) {
    using Cfg = WaveCfg<CT>;
    auto kern = conv3x3_wave_kernel<CT, DEPTH>;
    const int lds = Cfg::lds_bytes(a->Cin / 16);
    static int attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            gcpx_set_error("conv3x3 wave: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        attr_lds = lds;
    }
    const int frames = a->src_row_frames ? a->n_src_rows : a->F;
    const int nitems = frames * (a->Hout / 4) * (a->Wout / 16);
    int grid = gcpx_conv_grid() / 2;                     // one 512-thread workgroup per CU
    if (nitems == 0) grid = a->src_row_frames ? grid : 0;
    else if (grid * 8 > nitems && !a->src_row_frames) grid = (nitems + 7) / 8;
    if (grid == 0) return GCPX_OK;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, *a, nitems);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

// -----------------------------------------------------------------------------------------------------------
// Decoder blocks with 16 output channels (additional_conv_layer, pyramid-0): wave-autonomous variant of the
// upsample + concat + BN/LReLU-on-load + 3x3 conv.  Same idea as the head kernel: one 512-thread workgroup per CU
// holds the packed weights in LDS; every wavefront owns items of 4 rows x 16 output pixels, stages the 4 x 10
// low-res patch (16 channels per chunk) and its bilinear 6 x 18 up-sampling in wave-private LDS, and runs
// 9 x 4 x 4 MFMAs per chunk.  No workgroup barrier in steady state.
// -----------------------------------------------------------------------------------------------------------
struct UpWaveCfg {
    static constexpr int RW = 18, RH = 6, LW = 10, LH = 4, CC = 16, CCP = 24;   // pitch 24: conflict-free B-operand reads
    static constexpr int RAW_FLOATS = LH * LW * CC;          // 640
    static constexpr int HI_FLOATS = RH * RW * CCP;          // 2592
    static constexpr int WAVE_FLOATS = RAW_FLOATS + HI_FLOATS;
    static constexpr int NS = (LH * LW * 4 + 63) / 64;       // raw float4 slots per lane (3)
    static int lds_bytes(int nchunk) { return nchunk * 9 * 64 * 16 + 8 * WAVE_FLOATS * 4 + 8 * 2 * 16 * 4; }
};

__global__ void __launch_bounds__(512, 2) conv3x3_up16_kernel(const gcpx_conv_args a, const int items_per_wave,
                                                              const int nitems) {
    using Cfg = UpWaveCfg;
    constexpr int RW = Cfg::RW, RH = Cfg::RH, LW = Cfg::LW, LH = Cfg::LH, CC = Cfg::CC, CCP = Cfg::CCP, NS = Cfg::NS;
    extern __shared__ float4 smem4[];
    const int nchunk = a.Cin / CC;
    float4* wl = smem4;                                                       // [nchunk][9][64] float4
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-level indices live in SGPRs: item decode / pointer bases then cost SALU, not VALU, issue slots
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* wbuf = reinterpret_cast<float*>(smem4 + nchunk * 9 * 64) + wave * Cfg::WAVE_FLOATS;
    float* raw = wbuf;
    float* hi = wbuf + Cfg::RAW_FLOATS;
    float* red = reinterpret_cast<float*>(smem4 + nchunk * 9 * 64) + 8 * Cfg::WAVE_FLOATS;   // [8 waves][2][16]
    const int j = lane & 15, q = lane >> 4;
    const int H = a.Hout, W = a.Wout, F = a.F, Hin = a.Hin, Win = a.Win;
    const int ncb = W / 16, nrq = H / 4;

    for (int i = tid; i < nchunk * 9 * 64; i += 512) wl[i] = reinterpret_cast<const float4*>(a.wpk)[i];
    __syncthreads();

    int pixoff[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) pixoff[pt] = (pt * RW + j) * CCP + q * 4;
    const int c0 = a.src[0].C;

    const int gw = blockIdx.x * 8 + wave;
    int item = gw * items_per_wave;
    const int item_end = min(item + items_per_wave, nitems);

    auto origin = [&](int it, int& f, int& y0, int& x0) {          // column-block-major inside a frame (see the head kernel)
        const int strip = it % nrq;
        const int t = it / nrq;
        y0 = strip * 4; f = t / ncb; x0 = (t % ncb) * 16;
    };
    // raw-patch slot k of this lane: float4 index lane + 64k -> (row, col, 4-channel group); tile independent
    int s_ry[NS], s_rx[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int t = (lane + 64 * k) >> 2;
        s_rx[k] = t % LW;
        s_ry[k] = t / LW;
    }
    float4 pre[NS];
    auto issue_loads = [&](int it, int chunk) {
        int f, y0, x0;
        origin(it, f, y0, x0);
        const int cg = chunk * CC;
        const bool first = cg < c0;
        const gcpx_conv_src& sr = first ? a.src[0] : a.src[1];
        const int cl = first ? cg : cg - c0;
        const int srcC = sr.C;
        const float* base = sr.ptr + (size_t)(f / sr.frame_div) * Hin * Win * srcC + cl;     // wave-uniform (scalar)
        const int ly0 = y0 / 2 - 1, lx0 = x0 / 2 - 1;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = lane + 64 * k;
            pre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < LH * LW * 4) {
                const int sy = min(max(ly0 + s_ry[k], 0), Hin - 1);          // replicate clamp (bilinear border rule)
                const int sx = min(max(lx0 + s_rx[k], 0), Win - 1);
                const unsigned off = __umul24(__umul24(sy, Win) + sx, srcC) + (idx & 3) * 4;
                pre[k] = *reinterpret_cast<const float4*>(base + off);
            }
        }
    };

    f32x4 st1 = f32x4{0, 0, 0, 0}, st2 = f32x4{0, 0, 0, 0};
    if (item < item_end) issue_loads(item, 0);

    for (; item < item_end; ++item) {
        int f, y0, x0;
        origin(item, f, y0, x0);
        f32x4 acc[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) acc[pt] = f32x4{0, 0, 0, 0};
        const bool top = (y0 == 0), bot = (y0 + 4 == H), lft = (x0 == 0), rgt = (x0 + 16 == W);

        for (int chunk = 0; chunk < nchunk; ++chunk) {
            // ---- registers -> raw patch (BatchNorm affine + LeakyReLU of the producer applied here) ----
            {
                const int cg = chunk * CC;
                const bool first = cg < c0;
                const gcpx_conv_src& sr = first ? a.src[0] : a.src[1];
                const int cl = first ? cg : cg - c0;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int idx = lane + 64 * k;
                    if (idx < LH * LW * 4)
                        *reinterpret_cast<float4*>(raw + idx * 4) = affine_act4(pre[k], sr.scale, sr.shift, cl + (idx & 3) * 4, sr.act);
                }
            }
            // next stage's loads go out now
            if (chunk + 1 < nchunk) issue_loads(item, chunk + 1);
            else if (item + 1 < item_end) issue_loads(item + 1, 0);
            // ---- bilinear x2 into the haloed 6 x 18 region (see conv3x3_kernel for the index algebra) ----
            {
                const int c4 = lane & 3, p = lane >> 2;                      // 16 columns per pass
                auto lerp_store = [&](int ry, int cx, bool zero) {
                    const float wx1 = (cx & 1) ? 0.75f : 0.25f, wx0 = 1.f - wx1;
                    const float wy1 = (ry & 1) ? 0.75f : 0.25f, wy0 = 1.f - wy1;
                    const float* r = raw + (((ry >> 1) * LW + (cx >> 1)) * CC + c4 * 4);
                    const float4 a00 = *reinterpret_cast<const float4*>(r);
                    const float4 a01 = *reinterpret_cast<const float4*>(r + CC);
                    const float4 a10 = *reinterpret_cast<const float4*>(r + LW * CC);
                    const float4 a11 = *reinterpret_cast<const float4*>(r + LW * CC + CC);
                    float4 v;
                    v.x = wy0 * (wx0 * a00.x + wx1 * a01.x) + wy1 * (wx0 * a10.x + wx1 * a11.x);
                    v.y = wy0 * (wx0 * a00.y + wx1 * a01.y) + wy1 * (wx0 * a10.y + wx1 * a11.y);
                    v.z = wy0 * (wx0 * a00.z + wx1 * a01.z) + wy1 * (wx0 * a10.z + wx1 * a11.z);
                    v.w = wy0 * (wx0 * a00.w + wx1 * a01.w) + wy1 * (wx0 * a10.w + wx1 * a11.w);
                    if (zero) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(hi + (ry * RW + cx) * CCP + c4 * 4) = v;
                };
#pragma unroll
                for (int ry = 0; ry < RH; ++ry)
                    lerp_store(ry, p + 1, (top && ry == 0) || (bot && ry == RH - 1));
                if (p < 12) {                                               // the two halo columns: 6 rows x 2 sides
                    const int ry = p >> 1, side = p & 1;
                    lerp_store(ry, side ? RW - 1 : 0,
                               (top && ry == 0) || (bot && ry == RH - 1) || (lft && side == 0) || (rgt && side == 1));
                }
            }
            // ---- MFMAs: 9 taps x 4 pixel groups x 4 k-steps ----
            const float4* wp = wl + chunk * 9 * 64 + lane;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int tapoff = ((tap / 3) * RW + (tap % 3)) * CCP;
                const float4 w = wp[tap * 64];
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    const float4 b = *reinterpret_cast<const float4*>(hi + pixoff[pt] + tapoff);
                    acc[pt] = mfma16(w.x, b.x, acc[pt]);
                    acc[pt] = mfma16(w.y, b.y, acc[pt]);
                    acc[pt] = mfma16(w.z, b.z, acc[pt]);
                    acc[pt] = mfma16(w.w, b.w, acc[pt]);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        // ---- epilogue: bias, raw NHWC store, BatchNorm partial sums ----
        const float4 bv = *reinterpret_cast<const float4*>(a.bias + q * 4);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            f32x4 v = acc[pt];
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            float* obase = a.out + ((size_t)f * H + y0) * W * 16;                     // wave-uniform
            *reinterpret_cast<float4*>(obase + (unsigned)((pt * W + x0 + j) * 16 + q * 4)) = make_float4(v[0], v[1], v[2], v[3]);
            st1 += v;
            st2 += v * v;
        }
    }
    if (a.stats_partial) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s1 = row16_sum(st1[r]);
            const float s2 = row16_sum(st2[r]);
            if (j == 0) {
                red[(wave * 2 + 0) * 16 + q * 4 + r] = s1;
                red[(wave * 2 + 1) * 16 + q * 4 + r] = s2;
            }
        }
        __syncthreads();
        if (tid < 32) {
            const int which = tid >> 4, c = tid & 15;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += red[(w * 2 + which) * 16 + c];
            a.stats_partial[((size_t)blockIdx.x * 2 + which) * 16 + c] = sum;
        }
    }
}

int launch_up16(const gcpx_conv_args* a, hipStream_t stream, bool query_only) {
    const int nchunk = a->Cin / 16;
    const int lds = UpWaveCfg::lds_bytes(nchunk);
    int grid = gcpx_conv_grid() / 2;                      // one 512-thread workgroup per CU
    const int nitems = a->F * (a->Hout / 4) * (a->Wout / 16);
    if (!a->stats_partial && grid * 8 > nitems) grid = (nitems + 7) / 8;
    if (query_only) return grid;
    if (a->wpk_split && a->split_layout == GCPX_SPLIT_ROWFOLD) return gcpx_launch_up16_fold(a, stream, grid);
    if (a->wpk_split && a->split_layout == GCPX_SPLIT_ROWFOLD16) return gcpx_launch_up16_fold16(a, stream, grid);
    if (a->wpk_split) return gcpx_launch_up16_split(a, stream, grid);
    static int lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_up16_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            gcpx_set_error("conv3x3 up16: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        lds_set = lds;
    }
    const int ipw = (nitems + grid * 8 - 1) / (grid * 8);
    hipLaunchKernelGGL(conv3x3_up16_kernel, dim3(grid), dim3(512), lds, stream, *a, ipw, nitems);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

int g_conv_grid = 0;

template <bool UP, int CC, int CT, int TILE>
int launch(const gcpx_conv_args* a, hipStream_t stream, bool query_only = false) {
    using Cfg = ConvCfg<UP, CC, CT, TILE>;
    if (a->Hout % Cfg::TH || a->Wout % Cfg::TW) {
        gcpx_set_error("conv3x3: output %dx%d not divisible by tile %dx%d", a->Hout, a->Wout, Cfg::TH, Cfg::TW);
        return GCPX_ERR_UNSUPPORTED;
    }
    const int ntx = a->Wout / Cfg::TW, nty = a->Hout / Cfg::TH;
    const int nfg = (a->F + Cfg::TF - 1) / Cfg::TF;
    const int ntiles = ntx * nty * nfg;
    auto kern = conv3x3_kernel<UP, CC, CT, TILE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) {
            gcpx_set_error("conv3x3: hipFuncSetAttribute(%d B LDS): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        attr_set = true;
    }
    // persistent grid: CUs x resident workgroups per CU for this variant; with stats_partial every one of the
    // gcpx_conv3x3_grid(a) rows must be written, so the grid is not clamped to the tile count then
    int grid = (gcpx_conv_grid() / 2) * ((TILE == 3) ? 4 : 2);
    if (!a->stats_partial && grid > ntiles) grid = ntiles;
    if (query_only) return grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES, stream, *a, ntx, nty, ntiles);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

}  // namespace

extern "C" int gcpx_conv_grid(void) {
    if (g_conv_grid == 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                cus = prop.multiProcessorCount;
        }
        g_conv_grid = cus * 2;
    }
    return g_conv_grid;
}

int gcpx_launch_head_split(const gcpx_conv_args* a, hipStream_t stream);      // conv3x3_split.hip

static int conv3x3_dispatch(const gcpx_conv_args* a, hipStream_t stream, bool query_only) {
    GCPX_CHECK_ARG(a != nullptr, "null args");
    GCPX_CHECK_ARG(a->nsrc == 1 || a->nsrc == 2, "nsrc must be 1 or 2");
    GCPX_CHECK_ARG(a->src[0].C % 16 == 0 && (a->nsrc == 1 || a->src[1].C % 16 == 0), "source channels must be multiples of 16");
    // (a 16-output-channel data gradient on the split-f16 wave kernel may walk only the LEADING Cin channels of a wider source —
    //  src[0].C stays the pixel pitch: the adaptive model's head gradient, whose slots behind 8 x mixtures are zero)
    const bool leading = !a->upsample && a->nsrc == 1 && a->Cin < a->src[0].C && a->Cin >= 32 && a->Cin % 16 == 0 && a->Cout == 16 && a->wpk_split &&
                         a->split_layout == GCPX_SPLIT_PLAIN && (!a->src_row_map || a->src_row_frames) && !getenv("GCPX_DGRAD_TILED");
    GCPX_CHECK_ARG(leading || a->Cin == a->src[0].C + (a->nsrc == 2 ? a->src[1].C : 0), "Cin != sum of sources");
    GCPX_CHECK_ARG(a->wpk && a->bias, "weights/bias missing");
    GCPX_CHECK_ARG(a->F > 0, "F <= 0");
    if (a->upsample) GCPX_CHECK_ARG(a->Hout == 2 * a->Hin && a->Wout == 2 * a->Win, "upsample: Hout != 2*Hin");
    else GCPX_CHECK_ARG(a->Hout == a->Hin && a->Wout == a->Win, "no upsample: Hout != Hin");
    const int CT = (a->Cout + 15) / 16;
    const bool need_out = a->head_mode == GCPX_HEAD_RAW || a->head_mode == GCPX_HEAD_DLM_BOTH;
    GCPX_CHECK_ARG(!need_out || a->out, "out is NULL");
    GCPX_CHECK_ARG(need_out || a->images, "images is NULL");
    GCPX_CHECK_ARG(!need_out || a->out_pitch % 4 == 0, "out_pitch % 4");
    GCPX_CHECK_ARG(!a->images_rows || (a->wpk_split && a->raw_row_map && a->images && a->Cout == 100 && !a->upsample &&
                                       (a->head_mode == GCPX_HEAD_DLM_MEAN || a->head_mode == GCPX_HEAD_DLM_BOTH ||
                                        a->head_mode == GCPX_HEAD_DLM_NLL || a->head_mode == GCPX_HEAD_DLM_NLL_GRAD)),
                   "images_rows: split-f16 mixture head with raw_row_map and images only");
    const int W = a->Wout;
    GCPX_CHECK_ARG(!a->addend || (a->upsample && a->addend_frame_div > 0 && a->wpk_split && (a->Cout != 16 || a->split_layout == GCPX_SPLIT_ROWFOLD16)),
                   "addend: split-f16 upsampling blocks (32 / 64 output channels, or the 16-channel row-folded block), addend_frame_div > 0");
    if (!a->upsample) {
        GCPX_CHECK_ARG(a->nsrc == 1 && a->src[0].frame_div == 1, "non-upsampling 3x3 conv takes one per-frame source");
        if (a->Cin == 16 && !a->src_row_map && W % 16 == 0 && a->Hout % 4 == 0) {
            // 100-channel mixture head: 6 full tiles + the 4-channel remainder (weights packed by packing.pack_dlm_head)
            if (a->Cout == 100) {
                GCPX_CHECK_ARG(!need_out || a->out_pitch >= 100, "mixture head: out_pitch < 100");
                GCPX_CHECK_ARG(a->head_mode != GCPX_HEAD_DLM_NLL || a->wpk_split, "GCPX_HEAD_DLM_NLL is built for the split-f16 head (wpk_split)");
                if (a->wpk_split) return query_only ? gcpx_conv_grid() / 2 : gcpx_launch_head_split(a, stream);
                return query_only ? gcpx_conv_grid() / 2 : launch_head<6, true>(a, stream);
            }
            if (CT == 1) return query_only ? gcpx_conv_grid() / 2 : launch_head<1, false>(a, stream);
        }
        // data gradients of the decoder blocks (3x3 conv with the transposed, flipped weights): workgroup-tiled kernel
        GCPX_CHECK_ARG(a->head_mode == GCPX_HEAD_RAW && (!a->stats_partial || a->bwd_r), "plain 3x3 conv stores raw output, no statistics");
        if (a->bwd_r) {
            // activation backward in the epilogue: only the split-f16 wave-autonomous kernel with 16 output channels has it
            const bool fits = a->wpk_split && a->split_layout == GCPX_SPLIT_PLAIN && W % 16 == 0 && a->Hout % 4 == 0 && a->Cout == 16 &&
                              a->out_pitch == 16 && a->Cin >= 32 && a->Cin % 16 == 0 && (!a->src_row_map || a->src_row_frames) &&
                              a->stats_partial && a->bwd_scale && a->bwd_shift && a->bwd_mean && a->bwd_rstd && !getenv("GCPX_DGRAD_TILED");
            GCPX_CHECK_ARG(fits, "bwd_r: split-f16 plain 3x3 conv, 16 output channels at pitch 16, Cin >= 32, stats_partial and the four BatchNorm vectors");
        }
        static const bool tiled_only = getenv("GCPX_DGRAD_TILED") != nullptr;
        static const int depth = getenv("GCPX_DGRAD_DEPTH") ? atoi(getenv("GCPX_DGRAD_DEPTH")) : 2;
        GCPX_CHECK_ARG(!a->src_row_frames || (a->src_row_map && a->n_src_rows >= 0), "src_row_frames needs src_row_map and n_src_rows");
        // wave-autonomous kernel: 16 output channels, several 16-channel chunks, every frame has a source row or the caller gave
        // the inverse map
        if (!tiled_only && W % 16 == 0 && a->Hout % 4 == 0 && a->Cout == 16 && a->out_pitch % 4 == 0 && CT == 1 && a->Cin >= 32 &&
            (!a->src_row_map || a->src_row_frames) && WaveCfg<1>::lds_bytes(a->Cin / 16) <= 152 * 1024) {
            if (query_only) return gcpx_conv_grid() / 2;
            if (a->wpk_split && a->split_layout == GCPX_SPLIT_PLAIN) {
                const int st = gcpx_launch_wave_split(a, stream, 1, depth);
                if (st != -1) return st;
            }
            GCPX_CHECK_ARG(!leading, "Cin < src[0].C: the split-f16 wave kernel did not take the problem");
            return depth == 2 ? launch_wave<1, 2>(a, stream) : launch_wave<1, 1>(a, stream);
        }
        GCPX_CHECK_ARG(!leading, "Cin < src[0].C is built for the 16-output-channel wave kernel only (W % 16 == 0, H % 4 == 0)");
        static const bool wave32 = getenv("GCPX_DGRAD_NOWAVE32") == nullptr;     // 32 output channels (16-channel decoder blocks): 874 vs 912 us tiled
        if (wave32 && !tiled_only && W % 16 == 0 && a->Hout % 4 == 0 && a->Cout == 32 && a->out_pitch % 4 == 0 && a->Cin % 16 == 0 &&
            (!a->src_row_map || a->src_row_frames) && WaveCfg<2>::lds_bytes(a->Cin / 16) <= 152 * 1024) {
            if (query_only) return gcpx_conv_grid() / 2;
            if (a->wpk_split && a->split_layout == GCPX_SPLIT_PLAIN) {
                const int st = gcpx_launch_wave_split(a, stream, 2, 1);
                if (st != -1) return st;
            }
            return launch_wave<2, 1>(a, stream);
        }
        const int tile = W >= 32 ? 0 : (W == 16 ? 1 : (W == 8 ? 2 : -1));
#define GCPX_PLAIN(CC_, CT_)                                                            \
        do {                                                                            \
            if (tile == 0) return launch<false, CC_, CT_, 0>(a, stream, query_only);    \
            if (tile == 1) return launch<false, CC_, CT_, 1>(a, stream, query_only);    \
            if (tile == 2) return launch<false, CC_, CT_, 2>(a, stream, query_only);    \
        } while (0)
        if (a->Cin % 16 == 0 && CT == 4) GCPX_PLAIN(16, 4);
        if (a->Cin % 16 == 0 && CT == 2) GCPX_PLAIN(16, 2);
        if (a->Cin % 16 == 0 && CT == 1) GCPX_PLAIN(16, 1);
#undef GCPX_PLAIN
    } else {
        GCPX_CHECK_ARG(a->Cin % 32 == 0 || (a->Cin == 16 && a->wpk_split && a->split_layout == GCPX_SPLIT_ROWFOLD16),
                       "upsampling 3x3 conv expects Cin % 32 == 0 (16 channels: the row-folded split-f16 block, GCPX_SPLIT_ROWFOLD16)");
        GCPX_CHECK_ARG(a->head_mode == GCPX_HEAD_RAW, "decoder blocks store raw output");
        // 16-output-channel blocks: wave-autonomous kernel; its weights are packed in 16-channel chunks
        // (packing.pack_conv3x3(w, 16)), every other block in 32-channel chunks
        if (a->Cout == 16) {
            GCPX_CHECK_ARG(W % 16 == 0 && a->Hout % 4 == 0 && a->out_pitch == 16 && a->Cin <= 64, "16-channel block shape");
            return launch_up16(a, stream, query_only);
        }
        if (a->wpk_split && !query_only && a->Cout == CT * 16 && a->out_pitch == a->Cout && a->out_act == GCPX_ACT_NONE) {
            // same grid as the exact kernel (launch<> in query mode): the statistics rows are the same
            if (W == 16 && CT == 2) return gcpx_launch_up32_split(a, stream, 0, launch<true, 32, 2, 1>(a, stream, true));
            if (W == 8 && CT == 2) return gcpx_launch_up32_split(a, stream, 1, launch<true, 32, 2, 2>(a, stream, true));
            if (W == 8 && CT == 4) return gcpx_launch_up32_split(a, stream, 2, launch<true, 32, 4, 2>(a, stream, true));
        }
        // (only the split-f16 kernels above add gcpx_conv_args.addend: anything that falls through must not drop it silently)
        GCPX_CHECK_ARG(query_only || !a->addend, "addend: split-f16 upsampling blocks with 32 / 64 output channels only");
        if (W == 16 && CT == 2) return launch<true, 32, 2, 1>(a, stream, query_only);
        if (W == 8 && CT == 2) return launch<true, 32, 2, 2>(a, stream, query_only);
        if (W == 8 && CT == 4) return launch<true, 32, 4, 2>(a, stream, query_only);
    }
    gcpx_set_error("conv3x3: unsupported shape (up=%d Cin=%d Cout=%d W=%d)", a->upsample, a->Cin, a->Cout, W);
    return GCPX_ERR_UNSUPPORTED;
}

extern "C" int gcpx_conv3x3(const gcpx_conv_args* a, void* stream_) {
    return conv3x3_dispatch(a, reinterpret_cast<hipStream_t>(stream_), false);
}

extern "C" int gcpx_conv3x3_grid(const gcpx_conv_args* a) { return conv3x3_dispatch(a, nullptr, true); }
