// Decoder convolutions on the f16 matrix pipes of gfx950 with f32-equivalent arithmetic ("split-f16", split_mfma.h): the 16-channel
// upsampling blocks, the 32 / 64-channel upsampling blocks, the row-folded 32 -> 16 block, plain 3x3 convs of the backward pass and the
// device-side weight split.  The output head lives in conv3x3_head_split.hip.
#include "common.h"
#include "split_mfma.h"

#include <cstdlib>
#include <type_traits>

namespace {

// -----------------------------------------------------------------------------------------------------------
// Decoder blocks with 16 output channels (pyramid-0, additional_conv_layer) in split-f16: the wave-autonomous scheme of
// conv3x3_up16_kernel (conv3x3.hip) — raw 4 x 10 low-res patch per 16-channel chunk -> BatchNorm affine + LeakyReLU -> bilinear x2
// into the haloed 6 x 18 region — with the region written as two f16 planes and 5 k-steps x 4 pixel groups x 3 MFMAs per chunk.
// The power-of-two scale follows the chunks: a chunk with larger values than any before lowers it and the accumulators are
// rescaled (exact); a chunk with smaller values keeps it (its pieces are then small against the item's largest value, which is all
// a per-item scale promises).
// -----------------------------------------------------------------------------------------------------------
struct SplitUpCfg {
    static constexpr int RW = 18, RH = 6, LW = 10, LH = 4, CC = 16, KS = 5;
    static constexpr int RAW_BYTES = LH * LW * CC * 4;                     // 2560: f32 low-res patch
    static constexpr int PLANE_BYTES = RH * RW * 32;                       // 3456
    static constexpr int WAVE_BYTES = RAW_BYTES + 2 * PLANE_BYTES;         // 9472
    static constexpr int NS = (LH * LW * 4 + 63) / 64;                     // raw float4 slots per lane (3)
    static constexpr int W_CHUNK_BYTES = KS * 2 * 1024;                    // 10240 per 16-channel chunk
    static int lds_bytes(int nchunk) { return nchunk * W_CHUNK_BYTES + 8 * WAVE_BYTES + 8 * 2 * 16 * 4; }
};

__global__ void __launch_bounds__(512, 2) conv3x3_up16_split_kernel(const gcpx_conv_args a, const int items_per_wave,
                                                                    const int nitems) {
    using Cfg = SplitUpCfg;
    constexpr int RW = Cfg::RW, RH = Cfg::RH, LW = Cfg::LW, LH = Cfg::LH, CC = Cfg::CC, NS = Cfg::NS, KS = Cfg::KS;
    extern __shared__ float4 smem4[];
    const int nchunk = a.Cin / CC;
    const char* wl = reinterpret_cast<const char*>(smem4);                           // [nchunk][KS][1][2][64] x 16 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* wbuf = reinterpret_cast<char*>(smem4) + nchunk * Cfg::W_CHUNK_BYTES + wave * Cfg::WAVE_BYTES;
    float* raw = reinterpret_cast<float*>(wbuf);
    char* hi = wbuf + Cfg::RAW_BYTES;
    float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(smem4) + nchunk * Cfg::W_CHUNK_BYTES + 8 * Cfg::WAVE_BYTES);
    const int j = lane & 15, q = lane >> 4;
    const int H = a.Hout, W = a.Wout, Hin = a.Hin, Win = a.Win;
    const int ncb = W / 16, nrq = H / 4;

    for (int i = tid; i < nchunk * Cfg::W_CHUNK_BYTES / 16; i += 512) smem4[i] = reinterpret_cast<const float4*>(a.wpk_split)[i];
    __syncthreads();

    int tapoff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int tap = min(2 * s + (q >> 1), 8);
        tapoff[s] = ((tap / 3) * RW + (tap % 3) + j) * 32 + (q & 1) * 16;
    }
    const int c0 = a.src[0].C;
    const int ew = a.w_split_log2_dev ? __builtin_amdgcn_readfirstlane(*a.w_split_log2_dev) : a.w_split_log2;
    const float4 bv = *reinterpret_cast<const float4*>(a.bias + q * 4);

    const int gw = blockIdx.x * 8 + wave;
    int item = gw * items_per_wave;
    const int item_end = min(item + items_per_wave, nitems);

    auto origin = [&](int it, int& f, int& y0, int& x0) {
        const int strip = it % nrq;
        const int t = it / nrq;
        y0 = strip * 4; f = t / ncb; x0 = (t % ncb) * 16;
    };
    int s_rc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int t = (lane + 64 * k) >> 2;
        s_rc[k] = ((t / LW) << 8) | (t % LW);
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
        const float* base = sr.ptr + (size_t)(f / sr.frame_div) * Hin * Win * srcC + cl;
        const int ly0 = y0 / 2 - 1, lx0 = x0 / 2 - 1;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int idx = lane + 64 * k;
            pre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < LH * LW * 4) {
                const int sy = min(max(ly0 + (s_rc[k] >> 8), 0), Hin - 1);          // replicate clamp (bilinear border rule)
                const int sx = min(max(lx0 + (s_rc[k] & 255), 0), Win - 1);
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
        f32x4 acc[1][4];
        const bool top = (y0 == 0), bot = (y0 + 4 == H), lft = (x0 == 0), rgt = (x0 + 16 == W);
        int ex = 0;                                          // the accumulators hold (sum) 2^(ex + ew)

        for (int chunk = 0; chunk < nchunk; ++chunk) {
            // ---- registers -> raw patch (BatchNorm affine + LeakyReLU of the producer), largest magnitude of the chunk ----
            float amax = 0.f;
            {
                const int cg = chunk * CC;
                const bool first = cg < c0;
                const gcpx_conv_src& sr = first ? a.src[0] : a.src[1];
                const int cl = first ? cg : cg - c0;
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int idx = lane + 64 * k;
                    if (idx < LH * LW * 4) {
                        const float4 v = affine_act4(pre[k], sr.scale, sr.shift, cl + (idx & 3) * 4, sr.act);
                        *reinterpret_cast<float4*>(raw + idx * 4) = v;
                        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                    }
                }
            }
            if (chunk + 1 < nchunk) issue_loads(item, chunk + 1);
            else if (item + 1 < item_end) issue_loads(item + 1, 0);
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
            // bilinear values are convex combinations of the patch: |v| <= amax.  amax 2^ec in [2^14, 2^15)
            int ec = 14 + 127 - (int)((__float_as_uint(amax) >> 23) & 0xff);
            ec = __builtin_amdgcn_readfirstlane(amax > 0.f ? max(-100, min(min(100, 126 - ew), ec)) : min(100, 126 - ew));
            if (chunk == 0) ex = ec;
            else if (ec < ex) {                               // larger values than before: lower the scale, rescale the sums (exact)
                const float r = __uint_as_float((unsigned)(127 + ec - ex) << 23);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[0][pt] *= r;
                ex = ec;
            }
            const float sx2 = __uint_as_float((unsigned)(127 + ex) << 23);
            // ---- bilinear x2 into the haloed 6 x 18 region, as two f16 planes ----
            {
                const int c4 = lane & 3, p = lane >> 2;
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
                    v.x *= sx2; v.y *= sx2; v.z *= sx2; v.w *= sx2;
                    h4 p1, p2;
                    p1[0] = (_Float16)v.x; p1[1] = (_Float16)v.y; p1[2] = (_Float16)v.z; p1[3] = (_Float16)v.w;
                    p2[0] = (_Float16)(v.x - (float)p1[0]); p2[1] = (_Float16)(v.y - (float)p1[1]);
                    p2[2] = (_Float16)(v.z - (float)p1[2]); p2[3] = (_Float16)(v.w - (float)p1[3]);
                    char* dst = hi + (ry * RW + cx) * 32 + c4 * 8;
                    *reinterpret_cast<h4*>(dst) = p1;
                    *reinterpret_cast<h4*>(dst + Cfg::PLANE_BYTES) = p2;
                };
#pragma unroll
                for (int ry = 0; ry < RH; ++ry)
                    lerp_store(ry, p + 1, (top && ry == 0) || (bot && ry == RH - 1));
                if (p < 12) {
                    const int ry = p >> 1, side = p & 1;
                    lerp_store(ry, side ? RW - 1 : 0,
                               (top && ry == 0) || (bot && ry == RH - 1) || (lft && side == 0) || (rgt && side == 1));
                }
            }
            // ---- MFMAs: 5 k-steps x 4 pixel groups x 3 ----
            if (chunk == 0) mfma_tiles<0, 1, 1, true>(wl, hi, tapoff, lane, acc);
            else mfma_tiles<0, 1, 1, false>(wl + chunk * Cfg::W_CHUNK_BYTES, hi, tapoff, lane, acc);
        }
        // ---- epilogue: scale back (exact), bias, raw NHWC store, BatchNorm partial sums ----
        const float inv = __uint_as_float((unsigned)(127 - ex - ew) << 23);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            f32x4 v = acc[0][pt];
            v[0] = fmaf(v[0], inv, bv.x); v[1] = fmaf(v[1], inv, bv.y); v[2] = fmaf(v[2], inv, bv.z); v[3] = fmaf(v[3], inv, bv.w);
            float* obase = a.out + ((size_t)f * H + y0) * W * 16;
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

// -----------------------------------------------------------------------------------------------------------
// Decoder blocks with 32 / 64 output channels at low resolution (pyramid-1: 128 -> 32 @16x16, pyramid-2: 128 -> 64 @8x8) in split-f16:
// the workgroup-tiled scheme of conv3x3_kernel<true, 32, CT, TILE> (conv3x3.hip) — 256 output pixels per 256-thread workgroup, the
// low-res patch of a 32-channel chunk staged f32 (BatchNorm affine + LeakyReLU applied), bilinear x2 LDS -> LDS into the haloed
// region — with the region written as two f16 planes (64 B per pixel and plane) and one k-step = (tap, 32 channels): 9 k-steps x
// CT x 4 pixel groups x 3 MFMAs per chunk, weights streamed from L2 one k-step ahead.  The power-of-two scale is per (tile, chunk)
// (largest staged magnitude over the workgroup) and follows the chunks as in conv3x3_up16_split_kernel.
// -----------------------------------------------------------------------------------------------------------
template <int TILE> struct SplitTileShape;
template <> struct SplitTileShape<1> { static constexpr int TH = 16, TW = 16, TF = 1; };
template <> struct SplitTileShape<2> { static constexpr int TH = 8, TW = 8, TF = 4; };

template <int CT, int TILE>
struct SplitTileCfg {
    using TS = SplitTileShape<TILE>;
    static constexpr int TH = TS::TH, TW = TS::TW, TF = TS::TF, CC = 32, C4 = 8;
    static constexpr int RH = TH + 2, RW = TW + 2, LH = TH / 2 + 2, LW = TW / 2 + 2;
    static constexpr int PLANE_BYTES = TF * RH * RW * 64;
    static constexpr int RAW_BYTES = TF * LH * LW * CC * 4;
    static constexpr int LDS_BYTES = 2 * PLANE_BYTES + RAW_BYTES + 64;          // + the 4 per-wavefront maxima
    static constexpr int NSLOT = TF * LH * LW * C4;
    static constexpr int NS = (NSLOT + 255) / 256;
    static constexpr int PR = TH * TW * TF / 64;
};

template <int CT, int TILE>
__global__ void __launch_bounds__(256, 2) conv3x3_up32_split_kernel(const gcpx_conv_args a, const int ntx, const int nty, const int ntiles) {
    using Cfg = SplitTileCfg<CT, TILE>;
    constexpr int PR = Cfg::PR, TH = Cfg::TH, TW = Cfg::TW, TF = Cfg::TF, RH = Cfg::RH, RW = Cfg::RW;
    constexpr int LH = Cfg::LH, LW = Cfg::LW, CC = Cfg::CC, C4 = Cfg::C4, NS = Cfg::NS, NSLOT = Cfg::NSLOT;
    static_assert(PR == 4, "256 output pixels per workgroup");
    extern __shared__ float4 smem4[];
    char* hi = reinterpret_cast<char*>(smem4);
    float* raw = reinterpret_cast<float*>(hi + 2 * Cfg::PLANE_BYTES);
    float* wmax = reinterpret_cast<float*>(hi + 2 * Cfg::PLANE_BYTES + Cfg::RAW_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int nchunk = a.Cin / CC;
    const int Hout = a.Hout, Wout = a.Wout, F = a.F;
    const int ew = a.w_split_log2_dev ? __builtin_amdgcn_readfirstlane(*a.w_split_log2_dev) : a.w_split_log2;

    int pixoff[PR], pfl[PR], py[PR], px[PR];
#pragma unroll
    for (int pt = 0; pt < PR; ++pt) {
        const int p = (wave * PR + pt) * 16 + j;
        pfl[pt] = p / (TH * TW);
        const int rem = p % (TH * TW);
        py[pt] = rem / TW;
        px[pt] = rem % TW;
        pixoff[pt] = ((pfl[pt] * RH + py[pt]) * RW + px[pt]) * 64 + q * 16;          // bytes: channels 8 q .. 8 q + 7
    }
    auto slot = [&](int k, int& c4, int& rx, int& ry, int& fl, int& lds) {
        const int idx = tid + 256 * k;
        c4 = idx % C4;
        int t = idx / C4;
        rx = t % LW; t /= LW;
        ry = t % LH;
        fl = (idx < NSLOT) ? t / LH : -1;
        lds = ((fl * LH + ry) * LW + rx) * CC + c4 * 4;
    };
    // packed pieces: [chunk][tap][CT][2][64] x 16 B
    const char* wbase = reinterpret_cast<const char*>(a.wpk_split) + lane * 16;
    const int nstep = nchunk * 9;

    f32x4 st1[CT], st2[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { st1[ct] = f32x4{0, 0, 0, 0}; st2[ct] = f32x4{0, 0, 0, 0}; }

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
            const int f = f0 + s_fl;
            if (f < F) {
                const int sy = min(max(y0 / 2 - 1 + s_ry, 0), a.Hin - 1);
                const int sx = min(max(x0 / 2 - 1 + s_rx, 0), a.Win - 1);
                const int cg = chunk * CC + s_c4 * 4;
                const bool first = cg < c0;
                const gcpx_conv_src& sr = first ? a.src[0] : a.src[1];
                const int cl = first ? cg : cg - c0;
                pre[k] = *reinterpret_cast<const float4*>(sr.ptr + (((size_t)(f / sr.frame_div) * a.Hin + sy) * a.Win + sx) * sr.C + cl);
                pre_ok |= 1u << k;
            }
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
        int ex = 0;                                                   // the accumulators hold (sum) 2^(ex + ew)
        const bool top = (y0 == 0), bot = (y0 + TH == Hout), lft = (x0 == 0), rgt = (x0 + TW == Wout);

        h8 wr[3][CT][2];
        auto wload = [&](const int stn_, h8 (&w)[CT][2]) __attribute__((always_inline)) {
            const char* wp = wbase + (size_t)min(stn_, nstep - 1) * CT * 2048;           // (past the end: the last step again, never used)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                w[ct][0] = *reinterpret_cast<const h8*>(wp + ct * 2048);
                w[ct][1] = *reinterpret_cast<const h8*>(wp + ct * 2048 + 1024);
            }
        };
        if constexpr (CT <= 2) {
            wload(0, wr[0]);
            wload(1, wr[1]);
        }

        for (int chunk = 0; chunk < nchunk; ++chunk) {
            // ---- registers -> low-res patch (affine + LeakyReLU), largest magnitude of the chunk over the workgroup ----
            float amax = 0.f;
            {
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
                    *reinterpret_cast<float4*>(raw + s_lds) = v;
                    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                }
            }
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
            if (lane == 0) wmax[wave] = amax;
            __syncthreads();                       // patch complete, previous chunk's reads of the region done, maxima visible
            amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            int ec = 14 + 127 - (int)((__float_as_uint(amax) >> 23) & 0xff);
            ec = __builtin_amdgcn_readfirstlane(amax > 0.f ? max(-100, min(min(100, 126 - ew), ec)) : min(100, 126 - ew));
            if (chunk == 0) ex = ec;
            else if (ec < ex) {
                const float r = __uint_as_float((unsigned)(127 + ec - ex) << 23);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PR; ++pt) acc[ct][pt] *= r;
                ex = ec;
            }
            const float sx2 = __uint_as_float((unsigned)(127 + ex) << 23);
            // ---- bilinear x2 (align_corners=False) LDS -> LDS, written as two f16 planes (index algebra: conv3x3_kernel) ----
            {
                const int c4 = tid & (C4 - 1);
                const int p = tid >> 3;                                   // 0..31
                constexpr int RPP = 32 / (TW * TF);                       // region rows per pass (2 for 16x16, 1 for 8x8x4)
                const int fl = (TF > 1) ? p / TW : 0;
                const int rx = p % TW;
                const int rsub = (RPP == 2) ? (p / TW) : 0;
                auto lerp_store = [&](int f2, int ry, int cx, bool zero) {
                    const float wx1 = (cx & 1) ? 0.75f : 0.25f, wx0 = 1.f - wx1;
                    const float wy1 = ((ry & 1) ? 0.75f : 0.25f), wy0 = 1.f - wy1;
                    const float* r = raw + ((f2 * LH + (ry >> 1)) * LW + (cx >> 1)) * CC + c4 * 4;
                    const float4 a00 = *reinterpret_cast<const float4*>(r);
                    const float4 a01 = *reinterpret_cast<const float4*>(r + CC);
                    const float4 a10 = *reinterpret_cast<const float4*>(r + LW * CC);
                    const float4 a11 = *reinterpret_cast<const float4*>(r + LW * CC + CC);
                    float4 v;
                    v.x = wy0 * (wx0 * a00.x + wx1 * a01.x) + wy1 * (wx0 * a10.x + wx1 * a11.x);
                    v.y = wy0 * (wx0 * a00.y + wx1 * a01.y) + wy1 * (wx0 * a10.y + wx1 * a11.y);
                    v.z = wy0 * (wx0 * a00.z + wx1 * a01.z) + wy1 * (wx0 * a10.z + wx1 * a11.z);
                    v.w = wy0 * (wx0 * a00.w + wx1 * a01.w) + wy1 * (wx0 * a10.w + wx1 * a11.w);
                    const float sc = zero ? 0.f : sx2;
                    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                    h4 p1, p2;
                    p1[0] = (_Float16)v.x; p1[1] = (_Float16)v.y; p1[2] = (_Float16)v.z; p1[3] = (_Float16)v.w;
                    p2[0] = (_Float16)(v.x - (float)p1[0]); p2[1] = (_Float16)(v.y - (float)p1[1]);
                    p2[2] = (_Float16)(v.z - (float)p1[2]); p2[3] = (_Float16)(v.w - (float)p1[3]);
                    char* dst = hi + ((f2 * RH + ry) * RW + cx) * 64 + c4 * 8;
                    *reinterpret_cast<h4*>(dst) = p1;
                    *reinterpret_cast<h4*>(dst + Cfg::PLANE_BYTES) = p2;
                };
#pragma unroll 2
                for (int k = 0; k < RH / RPP; ++k) {
                    const int ry = k * RPP + rsub;
                    lerp_store(fl, ry, rx + 1, (top && ry == 0) || (bot && ry == RH - 1));
                }
                for (int e = p; e < TF * RH * 2; e += 32) {
                    const int side = e & 1;
                    const int ry = (e >> 1) % RH, f2 = (e >> 1) / RH;
                    lerp_store(f2, ry, side ? RW - 1 : 0,
                               (top && ry == 0) || (bot && ry == RH - 1) || (lft && side == 0) || (rgt && side == 1));
                }
            }
            __syncthreads();
            if (chunk + 1 < nchunk) issue_loads(tile, chunk + 1);
            else if (tile + (int)gridDim.x < ntiles) issue_loads(tile + gridDim.x, 0);

            if constexpr (CT <= 2) {
                // ---- MFMAs: 9 k-steps (tap, 32 channels) x CT x 4 pixel groups x 3 ----
                // weight fragments TWO k-steps ahead (three register sets in rotation, the tap loop unrolled by three so that the rotation
                // is a renaming): one step of MFMAs (12 CT x 16 cycles, shared with the partner wavefront) did not cover an L2 round trip
                // under load — the kernel spent 40-47 % of its wave cycles parked at these waits (PMC)
                auto tap_mfmas = [&](const int tap, const h8 (&wc)[CT][2]) __attribute__((always_inline)) {
                    const int tapoff = ((tap / 3) * RW + (tap % 3)) * 64;
                    h8 b1[PR], b2[PR];
    #pragma unroll
                    for (int pt = 0; pt < PR; ++pt) {
                        b1[pt] = *reinterpret_cast<const h8*>(hi + pixoff[pt] + tapoff);
                        b2[pt] = *reinterpret_cast<const h8*>(hi + pixoff[pt] + tapoff + Cfg::PLANE_BYTES);
                    }
    #pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
    #pragma unroll
                        for (int pt = 0; pt < PR; ++pt) acc[ct][pt] = mfma32h(wc[ct][1], b1[pt], acc[ct][pt]);
    #pragma unroll
                        for (int pt = 0; pt < PR; ++pt) acc[ct][pt] = mfma32h(wc[ct][0], b2[pt], acc[ct][pt]);
    #pragma unroll
                        for (int pt = 0; pt < PR; ++pt) acc[ct][pt] = mfma32h(wc[ct][0], b1[pt], acc[ct][pt]);
                    }
                };
                __builtin_amdgcn_s_setprio(1);
    #pragma unroll 1
                for (int t3 = 0; t3 < 9; t3 += 3) {      // (the last trip of a chunk requests the first two steps of the next one)
                    wload(chunk * 9 + t3 + 2, wr[2]);
                    tap_mfmas(t3, wr[0]);
                    wload(chunk * 9 + t3 + 3, wr[0]);
                    tap_mfmas(t3 + 1, wr[1]);
                    wload(chunk * 9 + t3 + 4, wr[1]);
                    tap_mfmas(t3 + 2, wr[2]);
                }
            } else {          // (64 output channels: the third register set would spill; one step ahead as before)
                // ---- MFMAs: 9 k-steps (tap, 32 channels) x CT x 4 pixel groups x 3 ----
                h8 wn[CT][2];
                {
                    const char* wp = wbase + (size_t)(chunk * 9) * CT * 2048;
    #pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        wn[ct][0] = *reinterpret_cast<const h8*>(wp + ct * 2048);
                        wn[ct][1] = *reinterpret_cast<const h8*>(wp + ct * 2048 + 1024);
                    }
                }
                __builtin_amdgcn_s_setprio(1);
    #pragma unroll 1
                for (int tap = 0; tap < 9; ++tap) {
                    const int tapoff = ((tap / 3) * RW + (tap % 3)) * 64;
                    h8 wc[CT][2];
    #pragma unroll
                    for (int ct = 0; ct < CT; ++ct) { wc[ct][0] = wn[ct][0]; wc[ct][1] = wn[ct][1]; }
                    {
                        const int stn = min(chunk * 9 + tap + 1, nstep - 1);        // one k-step ahead (the last one re-reads itself)
                        const char* wp = wbase + (size_t)stn * CT * 2048;
    #pragma unroll
                        for (int ct = 0; ct < CT; ++ct) {
                            wn[ct][0] = *reinterpret_cast<const h8*>(wp + ct * 2048);
                            wn[ct][1] = *reinterpret_cast<const h8*>(wp + ct * 2048 + 1024);
                        }
                    }
                    h8 b1[PR], b2[PR];
    #pragma unroll
                    for (int pt = 0; pt < PR; ++pt) {
                        b1[pt] = *reinterpret_cast<const h8*>(hi + pixoff[pt] + tapoff);
                        b2[pt] = *reinterpret_cast<const h8*>(hi + pixoff[pt] + tapoff + Cfg::PLANE_BYTES);
                    }
    #pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
    #pragma unroll
                        for (int pt = 0; pt < PR; ++pt) acc[ct][pt] = mfma32h(wc[ct][1], b1[pt], acc[ct][pt]);
    #pragma unroll
                        for (int pt = 0; pt < PR; ++pt) acc[ct][pt] = mfma32h(wc[ct][0], b2[pt], acc[ct][pt]);
    #pragma unroll
                        for (int pt = 0; pt < PR; ++pt) acc[ct][pt] = mfma32h(wc[ct][0], b1[pt], acc[ct][pt]);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }

        // ---- epilogue: scale back (exact), bias, raw NHWC store, BatchNorm partial sums ----
        const float inv = __uint_as_float((unsigned)(127 - ex - ew) << 23);
#pragma unroll
        for (int pt = 0; pt < PR; ++pt) {
            const int f = f0 + pfl[pt];
            if (f >= F) continue;
            float* op = a.out + (((size_t)f * Hout + (y0 + py[pt])) * Wout + (x0 + px[pt])) * a.out_pitch;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + ct * 16 + q * 4);
                f32x4 v = acc[ct][pt];
                v[0] = fmaf(v[0], inv, bv.x); v[1] = fmaf(v[1], inv, bv.y); v[2] = fmaf(v[2], inv, bv.z); v[3] = fmaf(v[3], inv, bv.w);
                if (a.addend) {
                    // the sequence's share of the conv (its skip channels, convolved once per sequence): gcpx_conv_args.addend
                    const float4 ad = *reinterpret_cast<const float4*>(a.addend + (((size_t)(f / a.addend_frame_div) * Hout + (y0 + py[pt])) * Wout +
                                                                                   (x0 + px[pt])) * a.out_pitch + ct * 16 + q * 4);
                    v[0] += ad.x; v[1] += ad.y; v[2] += ad.z; v[3] += ad.w;
                }
                *reinterpret_cast<float4*>(op + ct * 16 + q * 4) = make_float4(v[0], v[1], v[2], v[3]);
                if (a.stats_partial) {
                    st1[ct] += v;
                    st2[ct] += v * v;
                }
            }
        }
    }

    if (a.stats_partial) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(hi);   // [4 waves][2][CT*16]
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

template <int CT, int TILE>
int launch_up32_split(const gcpx_conv_args* a, hipStream_t stream, int grid) {
    using Cfg = SplitTileCfg<CT, TILE>;
    const int ntx = a->Wout / Cfg::TW, nty = a->Hout / Cfg::TH;
    const int nfg = (a->F + Cfg::TF - 1) / Cfg::TF;
    auto kern = conv3x3_up32_split_kernel<CT, TILE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) {
            gcpx_set_error("conv3x3 split up32: hipFuncSetAttribute(%d B LDS): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES, stream, *a, ntx, nty, ntx * nty * nfg);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}


// -----------------------------------------------------------------------------------------------------------
// Decoder blocks with 32 input and 16 output channels (pyramid-0, additional_conv_layer) with the VERTICAL half of the bilinear x2
// folded into the weights (gcpx_fold_upsample_weights).  conv3x3_up16_split_kernel is bound by its LDS traffic and staging VALU: every
// 16-pixel B fragment serves only three MFMAs, and the region is interpolated from four corners per element at full resolution.
// Folding the rows turns output rows 2 y and 2 y + 1 into two 3x3 convs over the SAME three horizontally-interpolated low-resolution
// rows y - 1 .. y + 1:
//   * the staged region has R + 2 low-resolution rows for 2 R output rows (two corners per element, half as many rows);
//   * a B fragment (row, tx) is read once per item and serves up to 3 rows x 2 parities x 3 MFMAs;
//   * one k-step = one (row tap, tx) x 32 channels: 9 k-steps per output group, no padded tenth tap.
// The conv's zero padding in x is exact (the out-of-image region columns are zero); in y the fold implies replicate padding, and the
// conv's zero row above row 0 / below row H - 1 is restored by three extra k-steps with -W0 / -W2 on the clamped border row (its
// horizontally interpolated values are exactly what the folded weights multiplied W0 / W2 with).
// One power-of-two scale per item (largest staged magnitude of the raw patch: the interpolated values are convex combinations).
// -----------------------------------------------------------------------------------------------------------
struct FoldCfg {
    static constexpr int R = 4;                                    // low-resolution rows per item: 8 output rows x 16 columns
    static constexpr int PR = R + 2, PW = 10, RW = 18;             // raw patch rows / columns (low-res), region columns (hi-res)
    static constexpr int RAW_BYTES = PR * PW * 32 * 4;             // 7680: f32 patch, aliases the start of the planes
    static constexpr int PLANE_BYTES = PR * RW * 64;               // 6912: one f16 piece, 32 channels x 2 B per pixel
    static constexpr int WAVE_BYTES = 2 * PLANE_BYTES;             // 13824
    static constexpr int NT = 24;                                  // fragment sets: 2 parities x 9 taps + 2 x 3 corrections
    static constexpr int W_BYTES = NT * 2048;                      // 49152
    static constexpr int LDS_BYTES = W_BYTES + 8 * WAVE_BYTES + 8 * 2 * 16 * 4;      // 160768 of 163840
    static constexpr int NS = (PR * PW * 8 + 63) / 64;             // raw float4 slots per lane (8)
    static constexpr int NO = (PR * RW * 8 + 63) / 64;             // region float4 outputs per lane (14)
};

template <bool PRIO>
__global__ void __launch_bounds__(512, 2) conv3x3_up16_fold_kernel(const gcpx_conv_args a, const int items_per_wave, const int nitems) {
    using Cfg = FoldCfg;
    constexpr int R = Cfg::R, PR = Cfg::PR, PW = Cfg::PW, RW = Cfg::RW, NS = Cfg::NS, NO = Cfg::NO;
    extern __shared__ float4 smem4[];
    const char* wl = reinterpret_cast<const char*>(smem4);                           // [24][2][64] x 16 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* wbuf = reinterpret_cast<char*>(smem4) + Cfg::W_BYTES + wave * Cfg::WAVE_BYTES;
    float* raw = reinterpret_cast<float*>(wbuf);
    float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(smem4) + Cfg::W_BYTES + 8 * Cfg::WAVE_BYTES);
    const int j = lane & 15, q = lane >> 4;
    const int H = a.Hout, W = a.Wout, Hin = a.Hin, Win = a.Win;
    const int ncb = W / 16, nrb = Hin / R;

    for (int i = tid; i < Cfg::W_BYTES / 16; i += 512) smem4[i] = reinterpret_cast<const float4*>(a.wpk_split)[i];
    __syncthreads();

    // a lane stages the same channel quad in every slot: its source, BatchNorm affine and activation are fixed
    const int c4 = lane & 7;
    const int c0 = a.src[0].C;
    const bool first = c4 * 4 < c0;
    const int cl = first ? c4 * 4 : c4 * 4 - c0;
    const float* sptr = first ? a.src[0].ptr : a.src[1].ptr;
    const int srcC = first ? a.src[0].C : a.src[1].C;
    const float* scp = first ? a.src[0].scale : a.src[1].scale;
    const float* shp = first ? a.src[0].shift : a.src[1].shift;
    const int sact = first ? a.src[0].act : a.src[1].act;
    float4 bn_s = make_float4(1.f, 1.f, 1.f, 1.f), bn_t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scp) {
        bn_s = *reinterpret_cast<const float4*>(scp + cl);
        bn_t = *reinterpret_cast<const float4*>(shp + cl);
    }
    const float slope = sact == GCPX_ACT_LRELU ? 0.2f : 1.f;
    const int fdiv0 = a.src[0].frame_div, fdiv1 = a.nsrc > 1 ? a.src[1].frame_div : 1;
    const int ew = a.w_split_log2_dev ? __builtin_amdgcn_readfirstlane(*a.w_split_log2_dev) : a.w_split_log2;
    const float4 bv = *reinterpret_cast<const float4*>(a.bias + q * 4);

    const int gw = blockIdx.x * 8 + wave;
    int item = gw * items_per_wave;
    const int item_end = min(item + items_per_wave, nitems);

    auto origin = [&](int it, int& f, int& y0, int& x0) {           // y0: first low-resolution row, x0: first output column
        const int rb = it % nrb;
        const int t = it / nrb;
        y0 = rb * R; f = t / ncb; x0 = (t % ncb) * 16;
    };
    // Slot maps with compile-time rows: a patch row is 10 pixels x 8 channel quads = 64 + 16 slots, a region row 18 x 8 = 2 x 64 + 16.
    // Steps 0 .. 5 (patch) / 0 .. 11 (region) take the first 8 (16) pixels of one row — the lane-dependent part of every address is
    // the same in all of them — and the leftover columns of all rows share the last two steps.
    const int ps = lane >> 3;                                        // pixel inside a step
    const int WinC = Win * srcC;
    const int rowA = lane >> 4, colA = 8 + (ps & 1);                 // leftover patch slots: rows 0..3 (step 6), 4..5 (step 7; lanes < 32)
    float4 pre[NS];
    auto issue_loads = [&](int it) {
        int f, y0, x0;
        origin(it, f, y0, x0);
        const int fs = first ? f / fdiv0 : f / fdiv1;
        const float* base = sptr + (size_t)fs * Hin * WinC + cl;
        const int ly0 = y0 - 1, lx0 = x0 / 2 - 1;
        const int sxm = min(max(lx0 + ps, 0), Win - 1) * srcC;                      // replicate clamp (bilinear border rule)
        const int sxl = min(max(lx0 + colA, 0), Win - 1) * srcC;
#pragma unroll
        for (int k = 0; k < PR; ++k) {
            const int sy = min(max(ly0 + k, 0), Hin - 1);
            pre[k] = *reinterpret_cast<const float4*>(base + (unsigned)(sy * WinC + sxm));
        }
        pre[PR] = *reinterpret_cast<const float4*>(base + (unsigned)(min(max(ly0 + rowA, 0), Hin - 1) * WinC + sxl));
        pre[PR + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < 32) pre[PR + 1] = *reinterpret_cast<const float4*>(base + (unsigned)(min(max(ly0 + 4 + rowA, 0), Hin - 1) * WinC + sxl));
    };
    // LDS offsets (bytes): patch [row][10][32] f32, region planes [row][18][32] f16
    const int raw_wr_m = ps * 128 + c4 * 16;                          // + row * 1280
    const int raw_wr_l = (rowA * 10 + colA) * 128 + c4 * 16;          // step 6; step 7: + 4 * 1280
    const int raw_rd_m = (ps >> 1) * 128 + c4 * 16;                   // + (row * 10 + 4 h) * 128: patch columns col / 2, col / 2 + 1
    const int raw_rd_l = ((ps >> 1) * 10 + 8) * 128 + c4 * 16;        // leftover region columns 16, 17 of row ps / 2 (+ 4 rows in the last step)
    const int reg_wr_m = ps * 64 + c4 * 8;                            // + (row * 18 + 8 h) * 64
    const int reg_wr_l = ((ps >> 1) * 18 + 16 + (ps & 1)) * 64 + c4 * 8;
    const float wa_m = (ps & 1) ? 0.25f : 0.75f;                      // region column c: 0.75 / 0.25 (c even) or 0.25 / 0.75 (c odd); 16 + (ps & 1) likewise

    f32x4 st1 = f32x4{0, 0, 0, 0}, st2 = f32x4{0, 0, 0, 0};
    // An item's output is stored one iteration late, behind the NEXT item's staging.  The memory counter is in order: stores issued
    // right before the loop's back-edge would have to complete before the staging may touch the prefetched patch (the store round
    // trip, every item); issued behind the staging they are older than the next prefetch and long complete when it is waited for.
    f32x4 outv[R][2];
    float* optr = nullptr;
    auto store_out = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int yi = 0; yi < R; ++yi)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const f32x4 v = outv[yi][py];
                *reinterpret_cast<float4*>(optr + (unsigned)((2 * yi + py) * W * 16)) = make_float4(v[0], v[1], v[2], v[3]);
            }
    };
    if (item < item_end) issue_loads(item);

    for (; item < item_end; ++item) {
        int f, y0, x0;
        origin(item, f, y0, x0);
        const bool top = (y0 == 0), bot = (y0 + R == Hin), lft = (x0 == 0), rgt = (x0 + 16 == W);

        // ---- registers -> raw patch (BatchNorm affine + LeakyReLU of the producer), the item's largest magnitude ----
        float amax = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            float4 v = pre[k];
            v.x = fmaf(v.x, bn_s.x, bn_t.x); v.y = fmaf(v.y, bn_s.y, bn_t.y); v.z = fmaf(v.z, bn_s.z, bn_t.z); v.w = fmaf(v.w, bn_s.w, bn_t.w);
            v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope); v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
            char* dst = wbuf + (k < PR ? raw_wr_m + k * 1280 : raw_wr_l + (k - PR) * 4 * 1280);
            if (k < NS - 1 || lane < 32) *reinterpret_cast<float4*>(dst) = v;
            else v = make_float4(0.f, 0.f, 0.f, 0.f);
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        if (optr) store_out();
        amax = wave_max_nonneg(amax);
        int ex = 14 + 127 - (int)((__float_as_uint(amax) >> 23) & 0xff);            // amax 2^ex in [2^14, 2^15)
        ex = __builtin_amdgcn_readfirstlane(amax > 0.f ? max(-100, min(min(100, 126 - ew), ex)) : min(100, 126 - ew));
        const float sx2 = __uint_as_float((unsigned)(127 + ex) << 23);

        // ---- horizontal half of the bilinear x2: patch (PR x 10) -> region (PR x 18) as two f16 planes.  Region column c is output
        //      column x0 - 1 + c = 0.75 / 0.25 (c even) or 0.25 / 0.75 (c odd) of patch columns c / 2 and c / 2 + 1.  The planes
        //      overwrite the patch: every value is formed first (the wavefront's LDS operations execute in order) ----
        h4 o1[NO], o2[NO];
        const float sc_first = (lft && ps == 0) ? 0.f : sx2;                        // region column 0 / 17: the conv's zero padding in x
        const float sc_last = (rgt && (ps & 1)) ? 0.f : sx2;
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            const bool main_step = k < 2 * PR;
            const int r = k >> 1, h = k & 1;
            const char* rp = wbuf + (main_step ? raw_rd_m + (r * 10 + 4 * h) * 128 : raw_rd_l + (k - 2 * PR) * 4 * 1280);
            const float4 A = *reinterpret_cast<const float4*>(rp);
            const float4 B = *reinterpret_cast<const float4*>(rp + 128);
            const float sc = main_step ? (h == 0 ? sc_first : sx2) : sc_last;
            const float wa = wa_m * sc, wb = (1.f - wa_m) * sc;
            float4 v;
            v.x = fmaf(wa, A.x, wb * B.x); v.y = fmaf(wa, A.y, wb * B.y); v.z = fmaf(wa, A.z, wb * B.z); v.w = fmaf(wa, A.w, wb * B.w);
            h4 p1, p2;
            p1[0] = (_Float16)v.x; p1[1] = (_Float16)v.y; p1[2] = (_Float16)v.z; p1[3] = (_Float16)v.w;
            p2[0] = (_Float16)fmaf((float)p1[0], -1.f, v.x); p2[1] = (_Float16)fmaf((float)p1[1], -1.f, v.y);
            p2[2] = (_Float16)fmaf((float)p1[2], -1.f, v.z); p2[3] = (_Float16)fmaf((float)p1[3], -1.f, v.w);
            o1[k] = p1;
            o2[k] = p2;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            const bool main_step = k < 2 * PR;
            const int r = k >> 1, h = k & 1;
            char* dst = wbuf + (main_step ? reg_wr_m + (r * RW + 8 * h) * 64 : reg_wr_l + (k - 2 * PR) * 4 * RW * 64);
            if (k < NO - 1 || lane < 32) {
                *reinterpret_cast<h4*>(dst) = o1[k];
                *reinterpret_cast<h4*>(dst + Cfg::PLANE_BYTES) = o2[k];
            }
        }

        // ---- MFMAs: 18 blocks (tx, row tap, parity) of R x 3.  Every block's two weight pieces are read while the block before it
        //      computes (12 x 16 cycles of the matrix pipe cover an LDS round trip).  Row taps ascend inside a tx, so rows 0, 1 of the
        //      row fragments die early: rows 0..3 of tx + 1 are read during blocks 2..5 of tx, rows 4, 5 during its own first two
        //      blocks — at most 8 row fragments are live.  The border corrections ride at the end of their tx ----
        f32x4 acc[R][2];
#pragma unroll
        for (int yi = 0; yi < R; ++yi) { acc[yi][0] = f32x4{0, 0, 0, 0}; acc[yi][1] = f32x4{0, 0, 0, 0}; }
        auto load_w = [&](const int t, h8 (&w)[2]) __attribute__((always_inline)) {
            const char* wp = wl + t * 2048 + lane * 16;
            w[0] = *reinterpret_cast<const h8*>(wp);
            w[1] = *reinterpret_cast<const h8*>(wp + 1024);
        };
        auto load_b = [&](const int tx, const int r, h8 (&b)[2]) __attribute__((always_inline)) {
            const char* bp = wbuf + ((r * RW + j + tx) * 64 + q * 16);
            b[0] = *reinterpret_cast<const h8*>(bp);
            b[1] = *reinterpret_cast<const h8*>(bp + Cfg::PLANE_BYTES);
        };
        if (item + 1 < item_end) issue_loads(item + 1);              // in flight during this item's MFMAs (issued here, not before the
                                                                     // interpolation: there its 32 registers meant spills, and a
                                                                     // scratch reload waits for every load in flight)
        h8 wq[2][2], bq[2][PR][2], wt[2], wb[2];
        load_w(0, wq[0]);
#pragma unroll
        for (int r = 0; r < 4; ++r) load_b(0, r, bq[0][r]);
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
        static_for<0, 18>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value, tx = t / 6, blk = t % 6, dyl = blk / 2, py = blk % 2;
            constexpr bool more = t + 1 < 18;
            constexpr bool pre_next = blk >= 2 && tx + 1 < 3;                    // rows 0..3 of tx + 1
            constexpr bool pre_own = blk < 2;                                    // rows 4, 5 of this tx
            constexpr bool pre_c = blk == 4;                                     // correction weights of this tx
            if constexpr (more) {
                constexpr int t1 = t + 1, tx1 = t1 / 6, blk1 = t1 % 6;
                load_w((blk1 % 2) * 9 + (blk1 / 2) * 3 + tx1, wq[t1 & 1]);
            }
            if constexpr (pre_next) load_b(tx + 1, blk - 2, bq[(tx + 1) & 1][blk - 2]);
            if constexpr (pre_own) load_b(tx, 4 + blk, bq[tx & 1][4 + blk]);
            if constexpr (pre_c) { load_w(18 + tx, wt); load_w(21 + tx, wb); }
            const h8 w1 = wq[t & 1][0], w2 = wq[t & 1][1];
            // small terms first: they are added to the accumulator while it is still small
#pragma unroll
            for (int yi = 0; yi < R; ++yi) acc[yi][py] = mfma32h(w2, bq[tx & 1][yi + dyl][0], acc[yi][py]);
#pragma unroll
            for (int yi = 0; yi < R; ++yi) acc[yi][py] = mfma32h(w1, bq[tx & 1][yi + dyl][1], acc[yi][py]);
#pragma unroll
            for (int yi = 0; yi < R; ++yi) acc[yi][py] = mfma32h(w1, bq[tx & 1][yi + dyl][0], acc[yi][py]);
            __builtin_amdgcn_sched_group_barrier(0x100, (more ? 2 : 0) + ((pre_next || pre_own) ? 2 : 0) + (pre_c ? 4 : 0), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            if constexpr (blk == 5) {
                if (top) {                                           // output row 0: the tap row above the image is zero, not row 0 again
                    acc[0][0] = mfma32h(wt[1], bq[tx & 1][0][0], acc[0][0]);
                    acc[0][0] = mfma32h(wt[0], bq[tx & 1][0][1], acc[0][0]);
                    acc[0][0] = mfma32h(wt[0], bq[tx & 1][0][0], acc[0][0]);
                }
                if (bot) {                                           // output row H - 1 likewise
                    acc[R - 1][1] = mfma32h(wb[1], bq[tx & 1][PR - 1][0], acc[R - 1][1]);
                    acc[R - 1][1] = mfma32h(wb[0], bq[tx & 1][PR - 1][1], acc[R - 1][1]);
                    acc[R - 1][1] = mfma32h(wb[0], bq[tx & 1][PR - 1][0], acc[R - 1][1]);
                }
            }
        });
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);

        // ---- epilogue: scale back (exact), bias, BatchNorm partial sums; the raw NHWC store follows the next item's staging ----
        const float inv = __uint_as_float((unsigned)(127 - ex - ew) << 23);
        optr = a.out + (((size_t)f * H + 2 * y0) * W + x0 + j) * 16 + q * 4;
#pragma unroll
        for (int yi = 0; yi < R; ++yi)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                f32x4 v = acc[yi][py];
                v[0] = fmaf(v[0], inv, bv.x); v[1] = fmaf(v[1], inv, bv.y); v[2] = fmaf(v[2], inv, bv.z); v[3] = fmaf(v[3], inv, bv.w);
                outv[yi][py] = v;
                st1 += v;
                st2 += v * v;
            }
    }
    if (optr) store_out();
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

// -----------------------------------------------------------------------------------------------------------
// The row-folded block over SIXTEEN input channels (GCPX_SPLIT_ROWFOLD16): additional_conv_layer (16 + 16 -> 16 channels @64x64) after its
// skip half has been hoisted out (gcpx_conv_args.addend: the skip activations of I_0 are the same for the N nodes of a sequence, so they
// are convolved once per sequence — by this same kernel over the sequence's frames — and arrive as an addend).  Same scheme as
// conv3x3_up16_fold_kernel with half the channels per pixel: a 16x16x32 MFMA k-step now holds TWO horizontal taps x 16 channels
// (lane groups q >> 1 = 0 / 1 read pixel column tx / tx + 1), so the three taps of a row are a pair (tx 0, 1) and a single (tx 2, the
// other half of its k-step meets zero weights): 12 blocks of R x 3 MFMAs per item instead of 18, half the staging arithmetic.
// Fragment sets (packing.conv3x3_fold16_gather): [py][dyl][kind] (12), then the border corrections [top / bottom][kind] (4).
// -----------------------------------------------------------------------------------------------------------
struct Fold16Cfg {
    static constexpr int R = 4;
    static constexpr int PR = R + 2, PW = 10, RW = 18, CC = 16;
    static constexpr int RAW_BYTES = PR * PW * CC * 4;             // 3840: f32 patch, aliases the start of the planes
    static constexpr int PLANE_BYTES = PR * RW * 32;               // 3456: one f16 piece, 16 channels x 2 B per pixel
    static constexpr int WAVE_BYTES = 2 * PLANE_BYTES;             // 6912
    static constexpr int NT = 16;
    static constexpr int W_BYTES = NT * 2048;                      // 32768
    static constexpr int LDS_BYTES = W_BYTES + 8 * WAVE_BYTES + 8 * 2 * 16 * 4;
    static constexpr int NS = (PR * PW * 4 + 63) / 64;             // raw float4 slots per lane (4)
    static constexpr int NO = (PR * RW * 4 + 63) / 64;             // region float4 outputs per lane (7)
};

__global__ void __launch_bounds__(512, 2) conv3x3_up16_fold16_kernel(const gcpx_conv_args a, const int items_per_wave, const int nitems) {
    using Cfg = Fold16Cfg;
    constexpr int R = Cfg::R, PR = Cfg::PR, PW = Cfg::PW, RW = Cfg::RW, NS = Cfg::NS, NO = Cfg::NO;
    extern __shared__ float4 smem4[];
    const char* wl = reinterpret_cast<const char*>(smem4);                           // [16][2][64] x 16 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* wbuf = reinterpret_cast<char*>(smem4) + Cfg::W_BYTES + wave * Cfg::WAVE_BYTES;
    float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(smem4) + Cfg::W_BYTES + 8 * Cfg::WAVE_BYTES);
    const int j = lane & 15, q = lane >> 4;
    const int H = a.Hout, W = a.Wout, Hin = a.Hin, Win = a.Win;
    const int ncb = W / 16, nrb = Hin / R;

    for (int i = tid; i < Cfg::W_BYTES / 16; i += 512) smem4[i] = reinterpret_cast<const float4*>(a.wpk_split)[i];
    __syncthreads();

    // a lane stages the same channel quad in every slot
    const int c4 = lane & 3, ps = lane >> 2;                          // channel quad, pixel inside a step of 16
    const float* sptr = a.src[0].ptr;
    const int srcC = a.src[0].C;
    float4 bn_s = make_float4(1.f, 1.f, 1.f, 1.f), bn_t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.src[0].scale) {
        bn_s = *reinterpret_cast<const float4*>(a.src[0].scale + c4 * 4);
        bn_t = *reinterpret_cast<const float4*>(a.src[0].shift + c4 * 4);
    }
    const float slope = a.src[0].act == GCPX_ACT_LRELU ? 0.2f : 1.f;
    const int fdiv0 = a.src[0].frame_div;
    const int ew = a.w_split_log2_dev ? __builtin_amdgcn_readfirstlane(*a.w_split_log2_dev) : a.w_split_log2;
    const float4 bv = *reinterpret_cast<const float4*>(a.bias + q * 4);

    const int gw = blockIdx.x * 8 + wave;
    int item = gw * items_per_wave;
    const int item_end = min(item + items_per_wave, nitems);
    auto origin = [&](int it, int& f, int& y0, int& x0) {           // y0: first low-resolution row, x0: first output column
        const int rb = it % nrb;
        const int t = it / nrb;
        y0 = rb * R; f = t / ncb; x0 = (t % ncb) * 16;
    };
    // slot maps (lane constants): patch slot k = pixel ps + 16 k of the 6 x 10 patch, region slot k = pixel ps + 16 k of the 6 x 18 region
    int p_row[NS], p_col[NS], p_lds[NS];
    bool p_ok[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int pix = ps + 16 * k;
        p_ok[k] = pix < PR * PW;
        p_row[k] = min(pix, PR * PW - 1) / PW;
        p_col[k] = min(pix, PR * PW - 1) % PW;
        p_lds[k] = (p_row[k] * PW + p_col[k]) * 64 + c4 * 16;
    }
    int r_rd[NO], r_wr[NO];
    float r_wa[NO];
    bool r_ok[NO], r_c0[NO], r_c17[NO];
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        const int pix = ps + 16 * k;
        r_ok[k] = pix < PR * RW;
        const int r = min(pix, PR * RW - 1) / RW, c = min(pix, PR * RW - 1) % RW;
        r_rd[k] = (r * PW + (c >> 1)) * 64 + c4 * 16;                 // patch columns c / 2 and c / 2 + 1
        r_wr[k] = (r * RW + c) * 32 + c4 * 8;
        r_wa[k] = (c & 1) ? 0.25f : 0.75f;
        r_c0[k] = c == 0; r_c17[k] = c == RW - 1;
    }
    float4 pre[NS];
    auto issue_loads = [&](int it) {
        int f, y0, x0;
        origin(it, f, y0, x0);
        const float* base = sptr + (size_t)(f / fdiv0) * Hin * Win * srcC + c4 * 4;
        const int ly0 = y0 - 1, lx0 = x0 / 2 - 1;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int sy = min(max(ly0 + p_row[k], 0), Hin - 1), sx = min(max(lx0 + p_col[k], 0), Win - 1);      // replicate clamp (bilinear border rule)
            pre[k] = *reinterpret_cast<const float4*>(base + (unsigned)((sy * Win + sx) * srcC));
        }
    };

    f32x4 st1 = f32x4{0, 0, 0, 0}, st2 = f32x4{0, 0, 0, 0};
    f32x4 outv[R][2];
    float* optr = nullptr;
    auto store_out = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int yi = 0; yi < R; ++yi)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const f32x4 v = outv[yi][py];
                *reinterpret_cast<float4*>(optr + (unsigned)((2 * yi + py) * W * 16)) = make_float4(v[0], v[1], v[2], v[3]);
            }
    };
    if (item < item_end) issue_loads(item);

    // B fragment of (kind, region row): kind 0 = taps tx 0 | 1 (lane groups q >> 1), kind 1 = tap tx 2 (the other half re-reads it: finite, zero weights)
    const int boff0 = (j + (q >> 1)) * 32 + (q & 1) * 16, boff1 = (j + 2) * 32 + (q & 1) * 16;

    for (; item < item_end; ++item) {
        int f, y0, x0;
        origin(item, f, y0, x0);
        const bool top = (y0 == 0), bot = (y0 + R == Hin), lft = (x0 == 0), rgt = (x0 + 16 == W);

        // ---- registers -> raw patch (BatchNorm affine + LeakyReLU of the producer), the item's largest magnitude ----
        float amax = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            float4 v = pre[k];
            v.x = fmaf(v.x, bn_s.x, bn_t.x); v.y = fmaf(v.y, bn_s.y, bn_t.y); v.z = fmaf(v.z, bn_s.z, bn_t.z); v.w = fmaf(v.w, bn_s.w, bn_t.w);
            v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope); v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
            if (p_ok[k]) *reinterpret_cast<float4*>(wbuf + p_lds[k]) = v;
            else v = make_float4(0.f, 0.f, 0.f, 0.f);
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        // the next item's patch: requested as soon as this one's registers are free (16 registers: no spill), AHEAD of the previous
        // item's stores — with 144 MFMAs an item is too short to cover an HBM round trip from behind the interpolation (PMC: 56 % of the
        // wave cycles parked in s_waitcnt), and the in-order memory counter then never waits for a store on the way to these loads
        if (item + 1 < item_end) issue_loads(item + 1);
        if (optr) store_out();
        amax = wave_max_nonneg(amax);
        int ex = 14 + 127 - (int)((__float_as_uint(amax) >> 23) & 0xff);            // amax 2^ex in [2^14, 2^15)
        ex = __builtin_amdgcn_readfirstlane(amax > 0.f ? max(-100, min(min(100, 126 - ew), ex)) : min(100, 126 - ew));
        const float sx2 = __uint_as_float((unsigned)(127 + ex) << 23);

        // ---- horizontal half of the bilinear x2: patch (6 x 10) -> region (6 x 18) as two f16 planes (they overwrite the patch: every
        //      value is formed first, the wavefront's LDS operations execute in order) ----
        h4 o1[NO], o2[NO];
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            const float4 A = *reinterpret_cast<const float4*>(wbuf + r_rd[k]);
            const float4 B = *reinterpret_cast<const float4*>(wbuf + r_rd[k] + 64);
            const float sc = ((lft && r_c0[k]) || (rgt && r_c17[k])) ? 0.f : sx2;    // region column 0 / 17: the conv's zero padding in x
            const float wa = r_wa[k] * sc, wb = (1.f - r_wa[k]) * sc;
            float4 v;
            v.x = fmaf(wa, A.x, wb * B.x); v.y = fmaf(wa, A.y, wb * B.y); v.z = fmaf(wa, A.z, wb * B.z); v.w = fmaf(wa, A.w, wb * B.w);
            h4 p1, p2;
            p1[0] = (_Float16)v.x; p1[1] = (_Float16)v.y; p1[2] = (_Float16)v.z; p1[3] = (_Float16)v.w;
            p2[0] = (_Float16)fmaf((float)p1[0], -1.f, v.x); p2[1] = (_Float16)fmaf((float)p1[1], -1.f, v.y);
            p2[2] = (_Float16)fmaf((float)p1[2], -1.f, v.z); p2[3] = (_Float16)fmaf((float)p1[3], -1.f, v.w);
            o1[k] = p1;
            o2[k] = p2;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            if (r_ok[k]) {
                *reinterpret_cast<h4*>(wbuf + r_wr[k]) = o1[k];
                *reinterpret_cast<h4*>(wbuf + r_wr[k] + Cfg::PLANE_BYTES) = o2[k];
            }
        }

        // ---- MFMAs: 12 blocks (kind, row tap, parity) of R x 3; the weight pieces of the next block and the row fragments of the next
        //      kind are read while a block computes ----
        f32x4 acc[R][2];
#pragma unroll
        for (int yi = 0; yi < R; ++yi) { acc[yi][0] = f32x4{0, 0, 0, 0}; acc[yi][1] = f32x4{0, 0, 0, 0}; }
        auto load_w = [&](const int t, h8 (&w)[2]) __attribute__((always_inline)) {
            const char* wp = wl + t * 2048 + lane * 16;
            w[0] = *reinterpret_cast<const h8*>(wp);
            w[1] = *reinterpret_cast<const h8*>(wp + 1024);
        };
        auto load_b = [&](const int kind, const int r, h8 (&b)[2]) __attribute__((always_inline)) {
            const char* bp = wbuf + r * RW * 32 + (kind ? boff1 : boff0);
            b[0] = *reinterpret_cast<const h8*>(bp);
            b[1] = *reinterpret_cast<const h8*>(bp + Cfg::PLANE_BYTES);
        };
        h8 wq[2][2], bq[2][PR][2], wt[2], wb[2];
        load_w(0, wq[0]);
#pragma unroll
        for (int r = 0; r < 4; ++r) load_b(0, r, bq[0][r]);
        __builtin_amdgcn_s_setprio(1);
        static_for<0, 12>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value, kind = t / 6, blk = t % 6, dyl = blk / 2, py = blk % 2;
            constexpr bool more = t + 1 < 12;
            constexpr bool pre_next = blk >= 2 && kind + 1 < 2;                  // rows 0..3 of the next kind
            constexpr bool pre_own = blk < 2;                                    // rows 4, 5 of this kind
            constexpr bool pre_c = blk == 4;                                     // correction weights of this kind
            if constexpr (more) {
                constexpr int t1 = t + 1, kind1 = t1 / 6, blk1 = t1 % 6;
                load_w((blk1 % 2) * 6 + (blk1 / 2) * 2 + kind1, wq[t1 & 1]);
            }
            if constexpr (pre_next) load_b(kind + 1, blk - 2, bq[(kind + 1) & 1][blk - 2]);
            if constexpr (pre_own) load_b(kind, 4 + blk, bq[kind & 1][4 + blk]);
            if constexpr (pre_c) { load_w(12 + kind, wt); load_w(14 + kind, wb); }
            const h8 w1 = wq[t & 1][0], w2 = wq[t & 1][1];
            // small terms first: they are added to the accumulator while it is still small
#pragma unroll
            for (int yi = 0; yi < R; ++yi) acc[yi][py] = mfma32h(w2, bq[kind & 1][yi + dyl][0], acc[yi][py]);
#pragma unroll
            for (int yi = 0; yi < R; ++yi) acc[yi][py] = mfma32h(w1, bq[kind & 1][yi + dyl][1], acc[yi][py]);
#pragma unroll
            for (int yi = 0; yi < R; ++yi) acc[yi][py] = mfma32h(w1, bq[kind & 1][yi + dyl][0], acc[yi][py]);
            __builtin_amdgcn_sched_group_barrier(0x100, (more ? 2 : 0) + ((pre_next || pre_own) ? 2 : 0) + (pre_c ? 4 : 0), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            if constexpr (blk == 5) {
                if (top) {                                           // output row 0: the tap row above the image is zero, not row 0 again
                    acc[0][0] = mfma32h(wt[1], bq[kind & 1][0][0], acc[0][0]);
                    acc[0][0] = mfma32h(wt[0], bq[kind & 1][0][1], acc[0][0]);
                    acc[0][0] = mfma32h(wt[0], bq[kind & 1][0][0], acc[0][0]);
                }
                if (bot) {                                           // output row H - 1 likewise
                    acc[R - 1][1] = mfma32h(wb[1], bq[kind & 1][PR - 1][0], acc[R - 1][1]);
                    acc[R - 1][1] = mfma32h(wb[0], bq[kind & 1][PR - 1][1], acc[R - 1][1]);
                    acc[R - 1][1] = mfma32h(wb[0], bq[kind & 1][PR - 1][0], acc[R - 1][1]);
                }
            }
        });
        __builtin_amdgcn_s_setprio(0);

        // ---- epilogue: scale back (exact), bias, the sequence's addend, BatchNorm partial sums; the raw NHWC store follows the next
        //      item's staging ----
        const float inv = __uint_as_float((unsigned)(127 - ex - ew) << 23);
        optr = a.out + (((size_t)f * H + 2 * y0) * W + x0 + j) * 16 + q * 4;
        const float* adp = a.addend ? a.addend + (((size_t)(f / a.addend_frame_div) * H + 2 * y0) * W + x0 + j) * 16 + q * 4 : nullptr;
#pragma unroll
        for (int yi = 0; yi < R; ++yi)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                f32x4 v = acc[yi][py];
                v[0] = fmaf(v[0], inv, bv.x); v[1] = fmaf(v[1], inv, bv.y); v[2] = fmaf(v[2], inv, bv.z); v[3] = fmaf(v[3], inv, bv.w);
                if (adp) {
                    const float4 ad = *reinterpret_cast<const float4*>(adp + (unsigned)((2 * yi + py) * W * 16));
                    v[0] += ad.x; v[1] += ad.y; v[2] += ad.z; v[3] += ad.w;
                }
                outv[yi][py] = v;
                st1 += v;
                st2 += v * v;
            }
    }
    if (optr) store_out();
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

// -----------------------------------------------------------------------------------------------------------
// Plain 3x3 conv at full resolution with several 16-channel input chunks in split-f16: the data gradients of the output head
// (112 -> 16) and of the 16-channel decoder blocks (16 -> 32).  Control flow of conv3x3_wave_kernel (conv3x3.hip): items dealt
// round-robin over the grid's wavefronts, buffer loads with hardware bounds-check zeros, a prefetch cursor DEPTH (item, chunk) steps
// ahead, rows that no frame reads skipped, frames without a source row zero-filled up front.  Arithmetic of
// conv3x3_up16_split_kernel: a chunk's 6 x 18 x 16ch region as two f16 planes, 5 k-steps x CT x 4 tiles x 3 MFMAs, the
// power-of-two scale following the chunks (loss gradients of 1e-6 need it: unscaled they would sit in the f16 subnormals).
// -----------------------------------------------------------------------------------------------------------
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));

template <int CT>
struct WaveSplitCfg {
    static constexpr int RW = 18, RH = 6, KS = 5;
    static constexpr int PLANE_BYTES = RH * RW * 32;
    static constexpr int REGION_BYTES = 2 * PLANE_BYTES;
    static constexpr int W_CHUNK_BYTES = KS * CT * 2 * 1024;
    static constexpr int NS = (RH * RW * 4 + 63) / 64;
    static int lds_bytes(int nchunk) { return nchunk * W_CHUNK_BYTES + 8 * REGION_BYTES + 256 + 1024; }      // + ACTB: four 16-channel vectors, [8 wavefronts][2][16] sums
};

// ACTB (CT = 1): the activation + BatchNorm-statistics step of the backward pass in the epilogue (gcpx_conv_args.bwd_r)
template <int CT, int DEPTH, bool ACTB = false>
__global__ void __launch_bounds__(512, 2) conv3x3_wave_split_kernel(const gcpx_conv_args a, const int nitems) {
    using Cfg = WaveSplitCfg<CT>;
    constexpr int RW = Cfg::RW, RH = Cfg::RH, NS = Cfg::NS, KS = Cfg::KS;
    extern __shared__ float4 smem4[];
    const int nchunk = a.Cin / 16;
    const char* wl = reinterpret_cast<const char*>(smem4);                  // [nchunk][KS][CT][2][64] x 16 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* reg = reinterpret_cast<char*>(smem4) + nchunk * Cfg::W_CHUNK_BYTES + wave * Cfg::REGION_BYTES;
    const int j = lane & 15, q = lane >> 4;
    const int H = a.Hout, W = a.Wout;
    const int ncb = W / 16, nrp = H / 4, ipf = ncb * nrp;
    const gcpx_conv_src sr = a.src[0];
    const int Cs = sr.C;

    for (int i = tid; i < nchunk * Cfg::W_CHUNK_BYTES / 16; i += 512) smem4[i] = reinterpret_cast<const float4*>(a.wpk_split)[i];
    if (a.src_row_frames) {                                 // frames without a source row get zeros
        const int f4_per_frame = H * W * a.out_pitch / 4;
        for (int f = blockIdx.x; f < a.F; f += gridDim.x) {
            if (a.src_row_map[f] >= 0) continue;
            float4* op = reinterpret_cast<float4*>(a.out + (size_t)f * H * W * a.out_pitch);
            for (int i = tid; i < f4_per_frame; i += 512) op[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if constexpr (ACTB) {
        float* bx0 = reinterpret_cast<float*>(reinterpret_cast<char*>(smem4) + nchunk * Cfg::W_CHUNK_BYTES + 8 * Cfg::REGION_BYTES);
        if (tid < 64) {
            const float* src = tid < 16 ? a.bwd_scale : tid < 32 ? a.bwd_shift : tid < 48 ? a.bwd_mean : a.bwd_rstd;
            bx0[tid] = src[tid & 15];
        }
        if (tid < 256) bx0[64 + tid] = 0.f;
    }
    __syncthreads();

    int tapoff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int tap = min(2 * s + (q >> 1), 8);
        tapoff[s] = ((tap / 3) * RW + (tap % 3) + j) * 32 + (q & 1) * 16;
    }
    const int ew = a.w_split_log2_dev ? __builtin_amdgcn_readfirstlane(*a.w_split_log2_dev) : a.w_split_log2;
    const int ex_cap = min(100, 126 - ew);

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
    auto geom = [&](int k, int& srow, int& y0, int& x0) {
        const int it = first + k * stride;
        srow = it / ipf;
        const int rem = it - srow * ipf;
        y0 = (rem / ncb) * 4; x0 = (rem % ncb) * 16;
    };

    // ---- prefetch cursor (conv3x3_wave_kernel) ----
    int pk = 0, pchunk = 0;
    unsigned poff[NS];
    unsigned pmask = 0;
    const float* pbase = sr.ptr;
    const unsigned frame_bytes = (unsigned)H * W * Cs * 4;
    int pf_next_v = 0;
    auto frame_of_item = [&](int k) { return a.src_row_frames[(first + k * stride) / ipf]; };
    if (a.src_row_frames && nmine > 0) pf_next_v = frame_of_item(0);
    auto enter_item = [&]() {
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
            const u32x4s v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, poff[k], soff, 0);
            pre[k] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
        if (++pchunk == nchunk) { pchunk = 0; ++pk; }
    };

    f32x4 acc[CT][4];
    int ck = 0, cchunk = 0;
    int f_v = 0;
    int ex = 0;                                             // the accumulators hold (sum) 2^(ex + ew)
    // ACTB: this lane's four channels of the layer behind the gradient (BatchNorm affine, batch statistics) and its running sums
    // (vectors and running sums live in LDS: held in registers across the prefetch pipeline they spill, and a scratch reload waits
    // for every load in flight)
    const int bx_off = nchunk * Cfg::W_CHUNK_BYTES + 8 * Cfg::REGION_BYTES;   // (uniform; the lane addresses are rebuilt in every epilogue)
    auto step = [&](float4 (&pre)[NS], unsigned& ok) {
        if (__builtin_amdgcn_readfirstlane(ok) == 0) {      // skipped item
            issue(pre, ok);
            if (++cchunk == nchunk) { cchunk = 0; ++ck; }
            return;
        }
        // registers -> (producer's affine + activation) -> the chunk's largest magnitude -> two f16 planes (instruction-count notes:
        // conv3x3_head_split.hip; slot k of a lane is float4 number lane + 64 k of the region: LDS byte 8 (lane + 64 k))
        float amax = 0.f;
        const bool plain = sr.scale == nullptr && sr.act == GCPX_ACT_NONE;       // (a gradient tensor: nothing to apply)
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (!plain && (ok & (1u << k))) pre[k] = affine_act4(pre[k], sr.scale, sr.shift, cchunk * 16 + (lane & 3) * 4, sr.act);
            amax = vmax3abs(amax, pre[k].x, pre[k].y);
            amax = vmax3abs(amax, pre[k].z, pre[k].w);
        }
        amax = wave_max_nonneg(amax);
        int ec = 14 + 127 - (int)((__float_as_uint(amax) >> 23) & 0xff);
        ec = amax > 0.f ? max(-100, min(ex_cap, ec)) : ex_cap;
        if (cchunk == 0) ex = ec;
        else if (ec < ex) {                                 // larger values than before: lower the scale, rescale the sums (exact)
            const float r = __uint_as_float((unsigned)(127 + ec - ex) << 23);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[ct][pt] *= r;
            ex = ec;
        }
        const float sx2 = __uint_as_float((unsigned)(127 + ex) << 23);
        char* const dst0 = reg + lane * 8;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (lane + 64 * k < RH * RW * 4) {
                h4 p1, p2;
                split4(pre[k], sx2, p1, p2);
                *reinterpret_cast<h4*>(dst0 + k * 512) = p1;
                *reinterpret_cast<h4*>(dst0 + k * 512 + Cfg::PLANE_BYTES) = p2;
            }
        }
        issue(pre, ok);                                     // this register set is free again: load the step DEPTH ahead
        if (cchunk == 0 && a.src_row_frames) f_v = a.src_row_frames[(first + ck * stride) / ipf];
        if (cchunk == 0) mfma_tiles<0, CT, CT, true>(wl, reg, tapoff, lane, acc);
        else mfma_tiles<0, CT, CT, false>(wl + cchunk * Cfg::W_CHUNK_BYTES, reg, tapoff, lane, acc);

        if (++cchunk == nchunk) {
            int srow, y0, x0;
            geom(ck, srow, y0, x0);
            const int f = a.src_row_frames ? __builtin_amdgcn_readfirstlane(f_v) : srow;
            const float inv = __uint_as_float((unsigned)(127 - ex - ew) << 23);
            float bs1[4] = {0.f, 0.f, 0.f, 0.f}, bs2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                float* op = a.out + (((size_t)f * H + (y0 + pt)) * W + (x0 + j)) * a.out_pitch;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const float4 bs = *reinterpret_cast<const float4*>(a.bias + ct * 16 + q * 4);
                    const f32x4 v = acc[ct][pt];
                    float4 o = make_float4(fmaf(v[0], inv, bs.x), fmaf(v[1], inv, bs.y), fmaf(v[2], inv, bs.z), fmaf(v[3], inv, bs.w));
                    if constexpr (ACTB) {
                        int q2 = q;
                        asm volatile("" : "+v"(q2));       // (opaque: otherwise the address is hoisted out of the item loop and spilled)
                        const float4* bvec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(smem4) + bx_off) + q2;
                        const float4 b_sc = bvec[0], b_sh = bvec[4], b_mu = bvec[8], b_rs = bvec[12];
                        const float4 rv = *reinterpret_cast<const float4*>(a.bwd_r + (op - a.out) + q2 * 4);      // (out_pitch == 16: the same offsets)
                        o.x *= fmaf(rv.x, b_sc.x, b_sh.x) > 0.f ? 1.f : 0.2f; o.y *= fmaf(rv.y, b_sc.y, b_sh.y) > 0.f ? 1.f : 0.2f;
                        o.z *= fmaf(rv.z, b_sc.z, b_sh.z) > 0.f ? 1.f : 0.2f; o.w *= fmaf(rv.w, b_sc.w, b_sh.w) > 0.f ? 1.f : 0.2f;
                        bs1[0] += o.x; bs1[1] += o.y; bs1[2] += o.z; bs1[3] += o.w;
                        bs2[0] += o.x * (rv.x - b_mu.x) * b_rs.x; bs2[1] += o.y * (rv.y - b_mu.y) * b_rs.y;
                        bs2[2] += o.z * (rv.z - b_mu.z) * b_rs.z; bs2[3] += o.w * (rv.w - b_mu.w) * b_rs.w;
                    }
                    *reinterpret_cast<float4*>(op + ct * 16 + q * 4) = o;
                }
            }
            if constexpr (ACTB) {
                // the item's sums over its 16 pixel lanes, added to the wavefront's LDS sums by one lane per channel group (same
                // order in every run: deterministic)
#pragma unroll
                for (int k = 0; k < 4; ++k) { bs1[k] = row16_sum_dpp(bs1[k]); bs2[k] = row16_sum_dpp(bs2[k]); }
                if (j == 0) {
                    int q2 = q;
                    asm volatile("" : "+v"(q2));
                    float* bsum = reinterpret_cast<float*>(reinterpret_cast<char*>(smem4) + bx_off + 256) + wave * 32 + q2 * 4;   // this wavefront's [2][16] sums
                    float4 t1 = *reinterpret_cast<float4*>(bsum), t2 = *reinterpret_cast<float4*>(bsum + 16);
                    t1.x += bs1[0]; t1.y += bs1[1]; t1.z += bs1[2]; t1.w += bs1[3];
                    t2.x += bs2[0]; t2.y += bs2[1]; t2.z += bs2[2]; t2.w += bs2[3];
                    *reinterpret_cast<float4*>(bsum) = t1;
                    *reinterpret_cast<float4*>(bsum + 16) = t2;
                }
            }
            cchunk = 0; ++ck;
        }
    };

    float4 preA[NS];
    unsigned okA = 0;
    issue(preA, okA);
    if constexpr (DEPTH == 3) {
        float4 preB[NS], preC[NS];
        unsigned okB = 0, okC = 0;
        issue(preB, okB);
        issue(preC, okC);
        for (int s = 0; s < nsteps; s += 3) {
            step(preA, okA);
            if (s + 1 < nsteps) step(preB, okB);
            if (s + 2 < nsteps) step(preC, okC);
        }
    } else if constexpr (DEPTH == 2) {
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
    if constexpr (ACTB) {
        // per-workgroup sums: the 8 wavefronts' LDS sums in a fixed order
        __syncthreads();
        const float* red = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem4) + nchunk * Cfg::W_CHUNK_BYTES + 8 * Cfg::REGION_BYTES + 256);
        if (tid < 32) {
            const int which = tid >> 4, c = tid & 15;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += red[(w * 2 + which) * 16 + c];
            a.stats_partial[((size_t)blockIdx.x * 2 + which) * 16 + c] = sum;
        }
    }
}

template <int CT, int DEPTH, bool ACTB = false>
int launch_wave_split_t(const gcpx_conv_args* a, hipStream_t stream) {
    using Cfg = WaveSplitCfg<CT>;
    auto kern = conv3x3_wave_split_kernel<CT, DEPTH, ACTB>;
    const int lds = Cfg::lds_bytes(a->Cin / 16);
    static int attr_lds = 0;
    if (lds > attr_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            gcpx_set_error("conv3x3 wave split: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        attr_lds = lds;
    }
    const int frames = a->src_row_frames ? a->n_src_rows : a->F;
    const int nitems = frames * (a->Hout / 4) * (a->Wout / 16);
    int grid = gcpx_conv_grid() / 2;
    if (nitems == 0) grid = (a->src_row_frames || ACTB) ? grid : 0;
    else if (grid * 8 > nitems && !a->src_row_frames && !ACTB) grid = (nitems + 7) / 8;      // (ACTB: every row of stats_partial is written)
    if (grid == 0) return GCPX_OK;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, *a, nitems);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

// out[t][co][ci] of gcpx_fold_upsample_weights (float64 sums in tap-row order; every product is exact)
__global__ void __launch_bounds__(256) fold_up_weights_kernel(const float* __restrict__ w, const int Cout, const int Cin, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = Cout * Cin;
    if (i >= 24 * n) return;
    const int t = i / n, e = i - t * n;
    const float* wp = w + (size_t)e * 9;
    double s;
    if (t < 18) {
        const int py = t / 9, dyl = (t % 9) / 3, tx = t % 3;
        const double cf[2][3][3] = {{{0.75, 0.25, 0.0}, {0.25, 0.75, 0.75}, {0.0, 0.0, 0.25}},
                                    {{0.25, 0.0, 0.0}, {0.75, 0.75, 0.25}, {0.0, 0.25, 0.75}}};
        s = cf[py][dyl][0] * (double)wp[tx];
        s = s + cf[py][dyl][1] * (double)wp[3 + tx];
        s = s + cf[py][dyl][2] * (double)wp[6 + tx];
    } else if (t < 21) {
        s = -(double)wp[t - 18];
    } else {
        s = -(double)wp[6 + t - 21];
    }
    out[i] = (float)s;
}

}  // namespace

// 16-output-channel upsampling decoder blocks with split-f16 weights (packing.pack_conv3x3_split); grid as launch_up16
int gcpx_launch_up16_split(const gcpx_conv_args* a, hipStream_t stream, int grid) {
    const int nchunk = a->Cin / 16;
    const int lds = SplitUpCfg::lds_bytes(nchunk);
    const int nitems = a->F * (a->Hout / 4) * (a->Wout / 16);
    static int lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_up16_split_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            gcpx_set_error("conv3x3 split up16: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        lds_set = lds;
    }
    const int ipw = (nitems + grid * 8 - 1) / (grid * 8);
    hipLaunchKernelGGL(conv3x3_up16_split_kernel, dim3(grid), dim3(512), lds, stream, *a, ipw, nitems);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

namespace {
// One 1024-thread workgroup per tensor: largest magnitude of the gathered weights -> exponent -> the two f16 pieces.
__global__ void __launch_bounds__(1024) split_pack_kernel(const float* __restrict__ theta, const int* __restrict__ idx, const int n,
                                                           _Float16* __restrict__ out, int* __restrict__ log2_out) {
    __shared__ float red[16];
    const int tid = threadIdx.x;
    float m = 0.f;
    for (int i = tid; i < n; i += 1024) {
        const int k = idx[i];
        m = fmaxf(m, k >= 0 ? fabsf(theta[k]) : 0.f);
    }
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    // largest magnitude times 2^e in [2^14, 2^15); the range the kernels' exponent bookkeeping covers (packing.split_f16 asserts it)
    int e = m > 0.f ? 14 + 127 - (int)((__float_as_uint(m) >> 23) & 0xff) : 0;
    e = max(-20, min(100, e));
    if (tid == 0) *log2_out = e;
    const float sc = __uint_as_float((unsigned)(127 + e) << 23);
    for (int i = tid; i < n; i += 1024) {
        const int k = idx[i];
        const float v = (k >= 0 ? theta[k] : 0.f) * sc;
        const _Float16 h1 = (_Float16)v;
        const _Float16 h2 = (_Float16)(v - (float)h1);
        const int o = (i >> 9) * 1024 + (i & 511);
        out[o] = h1;
        out[o + 512] = h2;
    }
}
}  // namespace

extern "C" int gcpx_split_pack(const float* theta, const int32_t* idx, int32_t n, void* out, int32_t* log2_out, void* stream_) {
    GCPX_CHECK_ARG(theta && idx && out && log2_out, "null pointer");
    GCPX_CHECK_ARG(n > 0 && n % 512 == 0, "n must be a positive multiple of 512");
    hipLaunchKernelGGL(split_pack_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream_), theta, idx, n,
                       reinterpret_cast<_Float16*>(out), log2_out);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

// 32 / 64-output-channel upsampling blocks (Cin % 32 == 0) with split-f16 weights (packing.pack_conv3x3_split32): which = 0 for the
// 16x16 tile with 2 channel tiles, 1 for 8x8x4 with 2, 2 for 8x8x4 with 4; grid as conv3x3.hip's launch<> computes it
int gcpx_launch_up32_split(const gcpx_conv_args* a, hipStream_t stream, int which, int grid) {
    if (which == 0) return launch_up32_split<2, 1>(a, stream, grid);
    if (which == 1) return launch_up32_split<2, 2>(a, stream, grid);
    return launch_up32_split<4, 2>(a, stream, grid);
}

// 32 -> 16 channel upsampling blocks with row-folded split-f16 weights (GCPX_SPLIT_ROWFOLD; packing.pack_conv3x3_fold); grid as launch_up16
int gcpx_launch_up16_fold(const gcpx_conv_args* a, hipStream_t stream, int grid) {
    using Cfg = FoldCfg;
    GCPX_CHECK_ARG(a->Cin == 32 && a->Cout == 16 && a->Hin % Cfg::R == 0 && a->Hout == 2 * a->Hin && a->Wout == 2 * a->Win && a->Wout % 16 == 0,
                   "row-folded block: 32 -> 16 channels, Hin a multiple of 4, output width a multiple of 16");
    GCPX_CHECK_ARG(a->src[0].C % 4 == 0 && (a->nsrc == 1 ? a->src[0].C == 32 : a->src[0].C + a->src[1].C == 32), "source channels");
    static const bool prio = getenv("GCPX_FOLD_NOPRIO") == nullptr;
    auto kern = prio ? conv3x3_up16_fold_kernel<true> : conv3x3_up16_fold_kernel<false>;
    static bool attr_set = false;
    if (!attr_set) {
        for (auto k : {conv3x3_up16_fold_kernel<true>, conv3x3_up16_fold_kernel<false>}) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
            if (e != hipSuccess) {
                gcpx_set_error("conv3x3 row-folded up16: hipFuncSetAttribute(%d B LDS): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
                return GCPX_ERR_HIP;
            }
        }
        attr_set = true;
    }
    const int nitems = a->F * (a->Hin / Cfg::R) * (a->Wout / 16);
    const int ipw = (nitems + grid * 8 - 1) / (grid * 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), Cfg::LDS_BYTES, stream, *a, ipw, nitems);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

// 16 -> 16 channel row-folded block (GCPX_SPLIT_ROWFOLD16: packing.pack_conv3x3_fold16); grid as launch_up16
int gcpx_launch_up16_fold16(const gcpx_conv_args* a, hipStream_t stream, int grid) {
    using Cfg = Fold16Cfg;
    GCPX_CHECK_ARG(a->Cin == 16 && a->Cout == 16 && a->nsrc == 1 && a->src[0].C == 16 && a->Hin % Cfg::R == 0 && a->Hout == 2 * a->Hin &&
                   a->Wout == 2 * a->Win && a->Wout % 16 == 0,
                   "row-folded 16-channel block: one 16-channel source, 16 output channels, Hin a multiple of 4, output width a multiple of 16");
    auto kern = conv3x3_up16_fold16_kernel;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) {
            gcpx_set_error("conv3x3 row-folded up16 (16 channels): hipFuncSetAttribute(%d B LDS): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        attr_set = true;
    }
    const int nitems = a->F * (a->Hin / Cfg::R) * (a->Wout / 16);
    const int ipw = (nitems + grid * 8 - 1) / (grid * 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), Cfg::LDS_BYTES, stream, *a, ipw, nitems);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_fold_upsample_weights(const float* w, int32_t Cout, int32_t Cin, float* out, void* stream_) {
    GCPX_CHECK_ARG(w && out && Cout > 0 && Cin > 0, "bad arguments");
    const int n = 24 * Cout * Cin;
    hipLaunchKernelGGL(fold_up_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), w, Cout, Cin, out);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

// plain 3x3 conv, 16 (ct = 1) or 32 (ct = 2) output channels, 16-channel input chunks, split-f16 weights (packing.pack_conv3x3_split):
// -1 when the shape has no split form (too many chunks for the LDS)
int gcpx_launch_wave_split(const gcpx_conv_args* a, hipStream_t stream, int ct, int depth) {
    const int nchunk = a->Cin / 16;
    if (ct == 1) {
        if (WaveSplitCfg<1>::lds_bytes(nchunk) > 160 * 1024) return -1;
        // (one register set in the prefetch pipeline: with two, the epilogue's extra values spill — 1107 against 1083 us for the head's
        // data gradient at c2)
        if (a->bwd_r) {
            static const int d = getenv("GCPX_WAVE_DEPTH") ? atoi(getenv("GCPX_WAVE_DEPTH")) : 1;
            return d == 3 ? launch_wave_split_t<1, 3, true>(a, stream) : d == 2 ? launch_wave_split_t<1, 2, true>(a, stream) : launch_wave_split_t<1, 1, true>(a, stream);
        }
        return depth == 2 ? launch_wave_split_t<1, 2>(a, stream) : launch_wave_split_t<1, 1>(a, stream);
    }
    if (WaveSplitCfg<2>::lds_bytes(nchunk) > 160 * 1024) return -1;
    return launch_wave_split_t<2, 1>(a, stream);
}
