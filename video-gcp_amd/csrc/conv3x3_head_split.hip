// Output head of the decoder (gen_head 16 -> 100 channels @ full resolution, DecoderModule.decode_seq,
// /root/reference/gcp/prediction/models/tree/tree_dense_rec.py:42) in split-f16 arithmetic (split_mfma.h) with the mixture mean and
// the discretised-logistic-mixture likelihood of the matched frames (decoder.nll, frame_binding.py:88-99) in its epilogue.
//
// Work decomposition = the wave-autonomous scheme of conv3x3_head_kernel (conv3x3.hip): one 512-thread workgroup per CU keeps all
// packed weights (5 k-steps x 7 channel tiles x 2 pieces x 1 KiB) in LDS; every wavefront owns items of 4 rows x 16 pixels, stages its
// haloed 6 x 18 x 16ch region as two f16 planes (32 B per pixel and plane: the ds_read_b128 operand reads are conflict-free without
// padding) and runs 5 k-steps (two taps x 16 channels = K 32 each; the 10th tap has zero weights) x 7 x 4 tiles x 3 MFMAs.
//
// On a gfx950 SIMD the matrix pipe's time and the VALU's time ADD (tools/mfma_valu_overlap.hip), and a transcendental costs about five
// plain VALU issues (2 against 10.7 cycles per wave instruction with both wavefronts of a SIMD busy): the kernel's time is its MFMA count
// plus its VALU budget, so everything outside the MFMA pass is written for instruction count (profiles/r05_head_isa_budget.txt):
//   * staging: slots 0..5 of a lane are ONE source row each (a 1 KiB coalesced read at lane * 16 B from a scalar row pointer), slot 6
//     the two halo columns; no per-slot address arithmetic; BatchNorm affine + LeakyReLU as fma / mul / max, the two f16 pieces as
//     v_fma_mixlo/hi_f16 pairs; the item's largest magnitude by DPP row operations and two permlane swaps;
//   * items by incremental (frame, column block, strip) bookkeeping in scalar registers (no integer division per item);
//   * mixture mean and likelihood in ONE loop over a lane's five mixtures (the tanh of the colour coefficients evaluated once), all
//     cross-lane sums over the lane pair of a pixel by v_permlane32_swap;
//   * likelihood: exp2-domain arguments, cdf_plus - cdf_min = (e_m - e_p) / ((1 + e_p)(1 + e_m)) with ONE reciprocal and ONE logarithm
//     per mixture (the three colour channels' numerators and denominators multiplied under a 2^-10 scale that keeps every product a
//     normal f32), the saturated-pixel cases (x = -1 / +1) as selects on the same terms, and a rarely taken exact path
//     (the formulas of dlm_nll_kernel, csrc/loss.hip) for the lanes whose bin probability falls below 1e-5 or whose terms leave the range.
#include "common.h"
#include "split_mfma.h"

#include <cstdlib>
#include <type_traits>

// (ablation switch of the gradient rows' scalar-register statement: profiles/r05_head_store_hazard.txt)
#ifndef GCPX_STORE_WAIT
#define GCPX_STORE_WAIT 1
#endif

namespace {

constexpr float L2E = 1.44269504088896341f, LN2 = 0.69314718055994531f;

// Two values that go through the same f32 arithmetic travel as a pair — of SCALAR instructions.  This file is built with
// -fno-slp-vectorize (csrc/build.sh) and holds no packed-f32 VALU instruction: v_pk_*_f32 measured no faster here (a packed operation
// costs its two scalar issues), and every wrong value of the round-4 corruption of the training variant's gradient rows came out of a
// packed multiply whose low lane read the high half of a register pair (profiles/r05_head_store_hazard.txt; tools/isa_hazard_scan.py
// keeps this file at zero such instructions).
struct f32x2 {
    float x, y;
};
__device__ __forceinline__ f32x2 operator+(const f32x2 a, const f32x2 b) { return f32x2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f32x2 operator-(const f32x2 a, const f32x2 b) { return f32x2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f32x2 operator*(const f32x2 a, const f32x2 b) { return f32x2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f32x2 operator-(const f32x2 a) { return f32x2{-a.x, -a.y}; }
__device__ __forceinline__ f32x2 pk_fma(const f32x2 a, const f32x2 b, const f32x2 c) { return f32x2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ f32x2 pk2(const float a, const float b) { return f32x2{a, b}; }
__device__ __forceinline__ f32x2 pk1(const float a) { return f32x2{a, a}; }

// scale back + bias, and the raw NHWC store of channel tiles C0 .. C0 + NC - 1.  FOLD (the kernels that never store raw parameters):
// the lanes that hold colour coefficients (odd lane groups, registers 0..2 of tiles 0..4) scale by `inv3` = inv x 2 log2(e) — the
// argument of the tanh's v_exp_f32 comes straight out of this fma (their bias slots are scaled alike when the bias is loaded)
template <int C0, int NC, bool FOLD = false>
__device__ __forceinline__ void finish_tiles(const gcpx_conv_args& a, const float* bias_l, f32x4 (&acc)[NC][4], const float inv, const float inv3,
                                             const bool store_raw, const int orow, const int y0, const int x0, const int j, const int q) {
    const f32x2 inv2 = pk1(inv), inv01 = pk1(FOLD ? inv3 : inv), inv23 = pk2(FOLD ? inv3 : inv, inv);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float4 bv = *reinterpret_cast<const float4*>(bias_l + (C0 + c) * 16 + q * 4);
        const f32x2 b01 = pk2(bv.x, bv.y), b23 = pk2(bv.z, bv.w);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const f32x2 lo = pk_fma(pk2(acc[c][pt][0], acc[c][pt][1]), inv01, b01), hi = pk_fma(pk2(acc[c][pt][2], acc[c][pt][3]), inv23, b23);
            acc[c][pt][0] = lo.x; acc[c][pt][1] = lo.y; acc[c][pt][2] = hi.x; acc[c][pt][3] = hi.y;
        }
    }
    (void)inv2;
    if (store_raw) {
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            float* op = a.out + (((size_t)orow * a.Hout + (y0 + pt)) * a.Wout + (x0 + j)) * a.out_pitch;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int ch = (C0 + c) * 16 + q * 4;
                if (ch < a.out_pitch) {
                    const f32x4 v = acc[c][pt];
                    *reinterpret_cast<float4*>(op + ch) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

__device__ __forceinline__ float exp2_hw(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float log2_hw(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float rcp_hw(float x) { return __builtin_amdgcn_rcpf(x); }
// hardware exp / log / reciprocal forms of the exact path (same helpers as csrc/loss.hip)
__device__ __forceinline__ float sigmoid_fast_s(float x) { return rcp_hw(1.f + exp_hw(-x)); }
__device__ __forceinline__ float softplus_s(float x) { return x > 20.f ? x : log_hw(1.f + exp_hw(x)); }

// NLL = 1: head mode GCPX_HEAD_DLM_NLL — frames matched to a ground-truth frame (raw_row_map entry >= 0) additionally evaluate the
// discretised-logistic-mixture likelihood of that frame in the epilogue and write one partial sum per item
// (nll_partial[item of the frame][row]); their raw parameters are never stored (frame_binding.py:88-99 -> decoder.nll).
// NLL = 2: GCPX_HEAD_DLM_NLL_GRAD (training forward) — the same, and the gradient of (nll_scale x row weight x) that likelihood
// w.r.t. the 100 parameters of every pixel goes to row raw_row_map[f] of `out` (112-slot layout): what gcpx_dlm_nll_bwd computes from
// the stored parameters, without storing them.
template <int NLL>
__global__ void __launch_bounds__(512, 2) conv3x3_head_split_kernel(const gcpx_conv_args a, const int items_per_wave,
                                                                    const int nitems) {
    using Cfg = SplitHeadCfg;
    constexpr int RW = Cfg::RW, RH = Cfg::RH, CT = Cfg::CT, KS = Cfg::KS;
    constexpr int NSL = 7;                                                          // staging slots per lane: six region rows + the halo columns
    // @phase prologue wave
    extern __shared__ float4 smem4[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* wl = reinterpret_cast<const char*>(smem4);                          // [KS][CT][2][64] x 16 B
    char* reg = reinterpret_cast<char*>(smem4) + Cfg::W_BYTES + wave * Cfg::REGION_BYTES;
    const int j = lane & 15, q = lane >> 4;
    const int H = a.Hout, W = a.Wout;
    const int ncb = W / 16, nrp = H / 4;

    for (int i = tid; i < Cfg::W_BYTES / 16; i += 512) smem4[i] = reinterpret_cast<const float4*>(a.wpk_split)[i];
    float* bias_l = reinterpret_cast<float*>(reinterpret_cast<char*>(smem4) + Cfg::W_BYTES + 8 * Cfg::REGION_BYTES);
    // (FOLD: slots 8k+4 .. 8k+6, the colour coefficients of mixture k, carry 2 log2(e) x bias — finish_tiles)
    constexpr bool FOLD = NLL != 0;
    if (tid < CT * 16) bias_l[tid] = tid < a.out_pitch ? a.bias[tid] * (FOLD && tid < 80 && (tid & 7) >= 4 && (tid & 7) < 7 ? 2.f * L2E : 1.f) : 0.f;
    __syncthreads();

    // operand address of k-step s: tap 2 s + (q >> 1), channels 8 (q & 1) .. + 7 of pixel (row + ty, j + tx).  The 10th tap (s = 4,
    // q >= 2) has zero weights; it re-reads tap 8 so that the multiplicand is a staged (finite) value.
    int tapoff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int tap = min(2 * s + (q >> 1), 8);
        tapoff[s] = ((tap / 3) * RW + (tap % 3) + j) * 32 + (q & 1) * 16;
    }
    const gcpx_conv_src sr = a.src[0];
    const int ew = a.w_split_log2_dev ? __builtin_amdgcn_readfirstlane(*a.w_split_log2_dev) : a.w_split_log2;
    // every staging slot of a lane carries the same 4 channels: their BatchNorm affine is loaded once
    float4 bn_s = make_float4(1.f, 1.f, 1.f, 1.f), bn_t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sr.scale) {
        bn_s = *reinterpret_cast<const float4*>(sr.scale + (lane & 3) * 4);
        bn_t = *reinterpret_cast<const float4*>(sr.shift + (lane & 3) * 4);
    }
    const float slope = sr.act == GCPX_ACT_LRELU ? 0.2f : 1.f;                        // LeakyReLU as max(v, slope v) (slope <= 1)

    // ---- staging map: slot k < 6 = region row k, region column 1 + (lane >> 2), channels 4 (lane & 3) .. + 3 — a source row is one
    // contiguous 1 KiB read at lane * 16 B; slot 6 (lanes 0..47) = region row lane >> 3, column 0 / 17 ----
    const int hrow = lane >> 3, hside = (lane >> 2) & 1;
    const int hoff = (hrow * W + (hside ? 16 : -1)) * 16 + (lane & 3) * 4;           // floats from the item's row pointer
    char* const lds_i = reg + 32 + lane * 8;
    char* const lds_h = reg + (hrow * RW + hside * 17) * 32 + (lane & 3) * 8;

    const int gw = blockIdx.x * 8 + wave;
    // NLL: matched frames cost ~1.6x an unmatched one (two more channel tiles + the likelihood), and a contiguous range of items is
    // about ONE frame — so the items are dealt round-robin over the wavefronts (every wavefront sees the same mix; the eight
    // wavefronts of a workgroup still work on eight neighbouring items at a time)
    const int istep = NLL ? (int)gridDim.x * 8 : 1;
    const int item0 = NLL ? gw : gw * items_per_wave;
    const int item_end = NLL ? nitems : min(item0 + items_per_wave, nitems);
    const int n_iter = NLL ? (nitems + istep - 1) / istep : items_per_wave;

    // item = (frame * ncb + column block) * nrp + strip, advanced by `istep` in scalar registers
    struct Pos { int f, cb, strip; };
    const int st_strip = istep % nrp, st_cb = (istep / nrp) % ncb, st_f = istep / nrp / ncb;
    // @phase bookkeeping item
    auto advance = [&](Pos& p) __attribute__((always_inline)) {
        p.strip += st_strip;
        int c = p.strip >= nrp;
        p.strip -= c ? nrp : 0;
        p.cb += st_cb + c;
        c = p.cb >= ncb;
        p.cb -= c ? ncb : 0;
        p.f += st_f + c;
    };
    // @phase prologue wave
    Pos nxt;                                               // the item whose activations are in flight
    nxt.strip = item0 % nrp; nxt.cb = (item0 / nrp) % ncb; nxt.f = item0 / nrp / ncb;

    float4 pre[NSL];
    int pre_orow = 0;                                      // raw_row_map entry of the prefetched item's frame
    // @phase loads item
    auto halo_ok = [&](const int y0, const int x0) __attribute__((always_inline)) {
        return lane < 48 && (unsigned)(y0 - 1 + hrow) < (unsigned)H && (unsigned)(x0 + (hside ? 16 : -1)) < (unsigned)W;
    };
    // @phase loads item
    auto issue_loads = [&](const Pos& p) __attribute__((always_inline)) {
        const int y0 = p.strip * 4, x0 = p.cb * 16;
        // (the row above the frame for y0 = 0: formed, never dereferenced)
        const float* rowp = sr.ptr + (((long long)p.f * H + (y0 - 1)) * W + x0) * 16;
        pre_orow = a.raw_row_map ? a.raw_row_map[p.f] : p.f;
#pragma unroll
        for (int k = 0; k < RH; ++k) {
            const bool rv = !(k == 0 && y0 == 0) && !(k == RH - 1 && y0 + 4 == H);
            pre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rv) pre[k] = *reinterpret_cast<const float4*>(rowp + (long long)k * W * 16 + lane * 4);
        }
        pre[RH] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (halo_ok(y0, x0)) pre[RH] = *reinterpret_cast<const float4*>(rowp + hoff);
    };
    if (item0 < item_end) issue_loads(nxt);

    // The image pixels of an item are stored one iteration late, behind the NEXT item's staging: the memory counter is in order, so
    // stores issued right before the loop's back-edge would have to complete before the staging may touch the prefetched region
    // (the store round trip, every item); behind the staging they are older than the next prefetch and long complete by its wait.
    float pend[2][3];
    float* pend_ip = nullptr;
    // images_rows: the frames with a raw_row_map entry are stored a second (and third) time, at that row of a [rows][3][H][W] array —
    // the gathers of the matched / kept frames behind the head (tree_dense_rec.py:56-60, tree.py:62-65) without their 2 x 63 MB round
    // trip.  The second address differs from the first by a wave-uniform offset: it lives in scalar registers.
    long long pend_d2 = 0;
    bool pend_has2 = false;
    const size_t plane = (size_t)H * W;
    // @phase image_stores item
    auto flush_images = [&]() __attribute__((always_inline)) {
        if (pend_ip) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float* ip = pend_ip + s2 * W;
                ip[0] = pend[s2][0];
                ip[plane] = pend[s2][1];
                ip[2 * plane] = pend[s2][2];
            }
            if (pend_has2) {
#pragma unroll
                for (int cpy = 0; cpy < 2; ++cpy) {
                    if (cpy == 1 && a.images_rows_dup == 0) break;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        float* ip = pend_ip + pend_d2 + cpy * a.images_rows_dup + s2 * W;
                        ip[0] = pend[s2][0];
                        ip[plane] = pend[s2][1];
                        ip[2 * plane] = pend[s2][2];
                    }
                }
            }
        }
    };

    // ---- the item loop: per item a VALU half (the deferred epilogue of the PREVIOUS item — scale-back, mixture mean, likelihood — then
    // the staging of this one) and an MFMA half.  The two wavefronts of a SIMD run free: lock step through workgroup barriers (one
    // wavefront in its MFMA half while its partner is in its VALU half) measured 10-17 % slower in every mode (round 3).
    f32x4 acc[5][4];
    float4* stash = reinterpret_cast<float4*>(reinterpret_cast<char*>(smem4) + Cfg::LDS_BYTES + wave * Cfg::STASH_BYTES) + lane;
    float ls4[2][2];
    float tx[2][3];                                             // fused likelihood: this lane's target pixels of the item in flight
    const int mode = a.head_mode;
    // the item whose accumulators wait for their epilogue (wave-uniform)
    int p_valid = 0, p_f = 0, p_y0 = 0, p_x0 = 0, p_orow = -1;
    float p_inv = 1.f;

    // ======== epilogue of an item: mixture mean (all frames) + likelihood (matched frames: WN) ========
    auto mixtures = [&](auto wn_tag, const int f, const int y0, const int x0, const int orow) __attribute__((always_inline)) {
        constexpr bool WN = decltype(wn_tag)::value;
        float nll_item = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            // ---- softmax weights of this lane's five mixtures (the other five: lane ^ 32) ----
            // @phase softmax item
            float m = vmax(acc[0][s2][0], acc[1][s2][0]);
            m = vmax(m, acc[2][s2][0]); m = vmax(m, acc[3][s2][0]); m = vmax(m, acc[4][s2][0]);
            m = pair_max(m);
            const float mL = m * L2E;
            float w[5], S = 0.f;
#pragma unroll
            for (int ct = 0; ct < 5; ++ct) {
                w[ct] = exp2_hw(fmaf(acc[ct][s2][0], L2E, -mL));
                S += w[ct];
            }
            S = pair_sum(S);
            // likelihood: targets, their bin edges, the saturated-pixel cases (wave-level masks), the stashed green / blue log-scales
            // @phase nll_setup item
            float x[3] = {0.f, 0.f, 0.f}, xp[3], lsg[5], lsb[5], lp2[5], lse2 = 0.f;
            bool lo[3], hi[3];
            f32x2 xpgb = pk1(0.f);
            constexpr int NG = NLL == 2 ? 5 : 1;           // gradient bookkeeping only in the training variant
            float gm[NG][3], gs[NG][3], cf[NG][3];
            if constexpr (WN) {
                lse2 = mL + log2_hw(S);                     // log2 sum exp of all ten logits
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    x[c] = tx[s2][c];                       // (requested one half-phase earlier, beside the staging)
                    xp[c] = x[c] + 1.f / 255.f;
                    lo[c] = x[c] < -0.999f; hi[c] = x[c] > 0.999f;
                }
                xpgb = pk2(xp[1], xp[2]);
                const float4 st0 = stash[(2 * s2) * 64], st1 = stash[(2 * s2 + 1) * 64];
                lsg[0] = st0.x; lsg[1] = st0.z; lsg[2] = st1.x; lsg[3] = st1.z; lsg[4] = ls4[s2][0];
                lsb[0] = st0.y; lsb[1] = st0.w; lsb[2] = st1.y; lsb[3] = st1.w; lsb[4] = ls4[s2][1];
            }
            // @phase mean item
            float Sr = 0.f, Sg = 0.f, Sb = 0.f;
#pragma unroll
            for (int ct = 0; ct < 5; ++ct) {
                const f32x4 e = acc[ct][s2], o = acc[ct][s2 + 2];          // {logit, mean r, g, b}, {coefficients 0..2, log-scale r}
                // tanh of the colour coefficients, 1 - 2 / (exp(2 x) + 1): (c0, c1) as a pair, c2 alone
                const f32x2 a01 = FOLD ? pk2(o[0], o[1]) : pk2(o[0], o[1]) * pk1(2.f * L2E);      // (FOLD: scaled by finish_tiles)
                const f32x2 q01 = pk2(exp2_hw(a01.x), exp2_hw(a01.y)) + pk1(1.f);
                const float q2 = exp2_hw(FOLD ? o[2] : o[2] * (2.f * L2E)) + 1.f;
                const f32x2 c01 = pk_fma(pk2(rcp_hw(q01.x), rcp_hw(q01.y)), pk1(-2.f), pk1(1.f));
                const float c2 = fmaf(rcp_hw(q2), -2.f, 1.f);
                const float mr = e[1];
                const f32x2 gb = pk_fma(c01, pk1(mr), pk2(e[2], e[3]));         // (mean g, mean b without its green term)
                const float mg = gb.x, mb = fmaf(c2, mg, gb.y);
                Sr = fmaf(w[ct], mr, Sr); Sg = fmaf(w[ct], mg, Sg); Sb = fmaf(w[ct], mb, Sb);
                if constexpr (WN) {
                    // @phase likelihood matched
                    // ---- likelihood of mixture 2 ct + (q >> 1) at this lane's pixel.  With e_p = exp(-plus_in), e_m = exp(-min_in):
                    //   cdf_plus - cdf_min = (e_m - e_p) / ((1 + e_p)(1 + e_m));   x < -0.999: cdf_plus = 1 / (1 + e_p);
                    //   x > 0.999: 1 - cdf_min = e_m / (1 + e_m), i.e. e_p := 0.
                    // e_p, e_m come out of v_exp_f32 already scaled by 2^-SC (SC = 10 in the forward variant: the product of three
                    // numerators / denominators stays a normal f32 and the three channels share one v_rcp_f32 and one v_log_f32; the
                    // training variant needs the per-channel quotients anyway and runs unscaled).  Red alone, (green, blue) as a pair.
                    constexpr int SC = NLL == 1 ? 10 : 0;
                    constexpr float S1 = NLL == 1 ? 9.765625e-4f : 1.f, S2 = S1 * S1;       // 2^-SC, 2^-2SC
                    constexpr float LIM = 40.f - SC;                            // (e_m <= 2^40: only x > 0.999 lanes get there unharmed)
                    const f32x2 mgb = pk_fma(c01, pk1(x[0]), pk2(e[2], e[3]));
                    const float mean[3] = {e[1], mgb.x, fmaf(c2, x[1], mgb.y)};
                    const float lsr[3] = {o[3], lsg[ct], lsb[ct]};
                    float tl[3], num[3], den[3], lsc[3], cdq[3], ap[3], am[3], ep[3], em[3], P[3], M[3];
                    bool bad[3];
                    if constexpr (NLL == 2) { cf[ct][0] = c01.x; cf[ct][1] = c01.y; cf[ct][2] = c2; }
#pragma unroll
                    for (int c = 0; c < 3; ++c) lsc[c] = vmax(lsr[c], -7.f);
                    // log2(e) / scale straight out of v_exp_f32 (log2 of log2(e) folded into its argument): the factor that turns
                    // (x - mean) into an exp2 argument
                    const f32x2 lgb = pk_fma(pk2(lsc[1], lsc[2]), pk1(-L2E), pk1(0.52876637294f));
                    tl[0] = exp2_hw(fmaf(lsc[0], -L2E, 0.52876637294f)); tl[1] = exp2_hw(lgb.x); tl[2] = exp2_hw(lgb.y);
                    const f32x2 tgb = pk2(tl[1], tl[2]);
                    const f32x2 apgb = pk_fma(-tgb, xpgb - pk2(mean[1], mean[2]), pk1(-(float)SC));
                    const f32x2 amgb = pk_fma(tgb, pk1(2.f / 255.f), apgb);
                    ap[0] = fmaf(-tl[0], xp[0] - mean[0], -(float)SC); ap[1] = apgb.x; ap[2] = apgb.y;
                    am[0] = fminf(fmaf(tl[0], 2.f / 255.f, ap[0]), LIM); am[1] = fminf(amgb.x, LIM); am[2] = fminf(amgb.y, LIM);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        ep[c] = exp2_hw(ap[c]);
                        em[c] = exp2_hw(am[c]);
                        ep[c] = hi[c] ? 0.f : ep[c];
                    }
                    const f32x2 epgb = pk2(ep[1], ep[2]), emgb = pk2(em[1], em[2]);
                    const f32x2 Pgb = epgb + pk1(S1), Mgb = emgb + pk1(S1);     // 2^-SC (1 + e_p), 2^-SC (1 + e_m)
                    const f32x2 dgb = Pgb * Mgb, ngb = emgb - epgb, thgb = dgb * pk1(1e-5f / S1);
                    P[0] = ep[0] + S1; M[0] = em[0] + S1; P[1] = Pgb.x; P[2] = Pgb.y; M[1] = Mgb.x; M[2] = Mgb.y;
                    den[0] = P[0] * M[0]; den[1] = dgb.x; den[2] = dgb.y;
                    const float dif[3] = {em[0] - ep[0], ngb.x, ngb.y}, thr[3] = {den[0] * (1e-5f / S1), thgb.x, thgb.y};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        num[c] = lo[c] ? M[c] : dif[c];
                        // bin probability > 1e-5 (as the reference's branch, on num / den instead of the difference of two sigmoids);
                        // false for every non-finite or out-of-range term
                        bad[c] = !(num[c] > thr[c]);
                        if constexpr (NLL == 2) {
                            // d log(bin) / d mean, d log(bin) / d log_scale (dlm_nll_bwd_kernel, csrc/backward.hip) from cdf_plus = M / den,
                            // cdf_min = P / den; x > 0.999 follows from e_p = 0, x < -0.999 needs cdf_min := 0
                            const float rd = rcp_hw(den[c]);
                            const float sp = M[c] * rd, sm = lo[c] ? 0.f : P[c] * rd;
                            // s (1 - s) as e s^2 (1 - cdf_plus = e_p cdf_plus, likewise cdf_min): no cancellation where a sigmoid
                            // saturates — s - s^2 in f32 loses 2e-4 of the value at |mid| = 8 (what the round-4 kernel and an f32
                            // evaluation of the reference's expression carry)
                            const float pp_ = ep[c] * sp * sp, pm_ = em[c] * sm * sm;
                            cdq[c] = num[c] * rd;
                            const float rcd = rcp_hw(cdq[c]);
                            const float plus_in = ap[c] * -LN2, min_in = am[c] * -LN2;
                            gm[ct][c] = -(tl[c] * LN2) * (pp_ - pm_) * rcd;
                            gs[ct][c] = -(plus_in * pp_ - min_in * pm_) * rcd;
                        }
                    }
                    // @phase likelihood_offpath rare
                    float extra2 = 0.f;                                       // log2 terms of the lanes off the main path
                    if (__any(bad[0] || bad[1] || bad[2])) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            if (__any(bad[c])) {
                                // The reference's branch for vanishing bins: log(pdf at the bin centre) - log(127.5) = log(2/255 t s'(mid)),
                                // s'(mid) = g / (1 + g)^2 with g = exp(-|mid|) — as numerator 2/255 t, denominator (1 + g)^2 and -|mid| as a
                                // log2 term (nothing underflows whatever |mid|): one v_exp_f32 for the lanes that need it
                                const float xc = x[c] - mean[c], t = tl[c] * LN2;          // t = 1 / scale
                                const float u = fabsf(tl[c] * xc);
                                const float g = exp2_hw(-u);
                                const float A = fmaf(g, S1, S1);
                                float vn = t * (2.f / 255.f * S1), vd = A * A, ve = -u, dm = 0.f, ds = 0.f;
                                if constexpr (NLL == 2) {
                                    const float h = (1.f - g) * rcp_hw(1.f + g);       // |1 - 2 sigmoid(mid)|
                                    dm = xc < 0.f ? -t * h : t * h;
                                    ds = fmaf(u * LN2, h, -1.f);
                                }
                                // beyond that form's reach — saturated pixels whose terms left the range, scales beyond 1e30: the exact
                                // formulas of dlm_nll_kernel (csrc/loss.hip)
                                const bool odd = bad[c] && (lo[c] || hi[c] || !(t > 1e-30f));
                                if (__any(odd)) {
                                    const float is = t;
                                    const float plus_in = is * (xc + 1.f / 255.f), min_in = is * (xc - 1.f / 255.f), mid_in = is * xc;
                                    float v, dmx = 0.f, dsx = 0.f;
                                    if (lo[c]) {
                                        v = plus_in - softplus_s(plus_in);
                                        if constexpr (NLL == 2) { const float sp = sigmoid_fast_s(plus_in); dmx = -is * (1.f - sp); dsx = -plus_in * (1.f - sp); }
                                    } else if (hi[c]) {
                                        v = -softplus_s(min_in);
                                        if constexpr (NLL == 2) { const float sm = sigmoid_fast_s(min_in); dmx = is * sm; dsx = min_in * sm; }
                                    } else {
                                        v = mid_in - lsc[c] - 2.f * softplus_s(mid_in) - 4.8481163864f;   // log(127.5)
                                        if constexpr (NLL == 2) {
                                            const float smid = sigmoid_fast_s(mid_in);
                                            dmx = -is * (1.f - 2.f * smid);
                                            dsx = -mid_in * (1.f - 2.f * smid) - 1.f;
                                        }
                                    }
                                    vn = odd ? S1 : vn;                       // (their quotient is the 2^SC of a channel that takes no part)
                                    vd = odd ? S2 : vd;
                                    ve = odd ? v * L2E : ve;
                                    if constexpr (NLL == 2) { dm = odd ? dmx : dm; ds = odd ? dsx : ds; }
                                }
                                extra2 += bad[c] ? ve : 0.f;
                                num[c] = bad[c] ? vn : num[c];
                                den[c] = bad[c] ? vd : den[c];
                                if constexpr (NLL == 2) {
                                    cdq[c] = bad[c] ? vn * rcp_hw(vd) : cdq[c];
                                    gm[ct][c] = bad[c] ? dm : gm[ct][c];
                                    gs[ct][c] = bad[c] ? ds : gs[ct][c];
                                }
                            }
                        }
                    }
                    // @phase likelihood matched
                    if constexpr (NLL == 2) {
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            if (lsr[c] < -7.f) gs[ct][c] = 0.f;               // clamp(min=-7) blocks the gradient
                    }
                    float prod;
                    if constexpr (NLL == 2) prod = cdq[0] * cdq[1] * cdq[2];
                    else prod = (num[0] * num[1] * num[2]) * rcp_hw(den[0] * den[1] * den[2]);
                    // log2 of (mixture weight x the three bin probabilities)
                    lp2[ct] = fmaf(e[0], L2E, log2_hw(prod)) + (extra2 - lse2 - 3.f * SC);
                }
            }
            // @phase pixels item
            Sr = pair_sum(Sr); Sg = pair_sum(Sg); Sb = pair_sum(Sb);
            const float invS = rcp_hw(S);
            pend[s2][0] = fminf(fmaxf(Sr * invS, -1.f), 1.f);
            pend[s2][1] = fminf(fmaxf(Sg * invS, -1.f), 1.f);
            pend[s2][2] = fminf(fmaxf(Sb * invS, -1.f), 1.f);
            if (s2 == 0) {
                if (q < 2) pend_ip = a.images + (size_t)f * 3 * plane + (size_t)(y0 + 2 * q) * W + (x0 + j);   // row pt = s2 + 2 q
                pend_has2 = a.images_rows != nullptr && orow >= 0;
                const long long d2 = pend_has2 ? (a.images_rows - a.images) + ((long long)orow - f) * 3 * (long long)plane : 0;
                pend_d2 = ((long long)__builtin_amdgcn_readfirstlane((int)(d2 >> 32)) << 32) |
                          (unsigned)__builtin_amdgcn_readfirstlane((int)(d2 & 0xffffffffll));
            }
            if constexpr (WN) {
                // @phase logsumexp matched
                float mx = fmaxf(fmaxf(fmaxf(lp2[0], lp2[1]), fmaxf(lp2[2], lp2[3])), lp2[4]);
                mx = pair_max(mx);
                float r[5], se = 0.f;
#pragma unroll
                for (int ct = 0; ct < 5; ++ct) { r[ct] = exp2_hw(lp2[ct] - mx); se += r[ct]; }
                se = pair_sum(se);
                nll_item -= (mx + log2_hw(se)) * LN2;          // (both lanes of a pixel hold it; only q < 2 is summed below)
                if constexpr (NLL == 2) {
                    // @phase gradient_rows matched
                    // ---- gradient rows: slots 8k .. 8k+7 of this lane's mixtures k = 2 ct + (q >> 1), then its g / b log-scales ----
                    const float coef = a.nll_scale * (a.nll_row_weight ? a.nll_row_weight[orow] : 1.f);
                    const float inv_se = rcp_hw(se);
                    float* drow = a.out + ((size_t)orow * plane + (size_t)(y0 + s2 + 2 * (q & 1)) * W + (x0 + j)) * a.out_pitch;
                    const int h = q >> 1;
                    const float xr = x[0], xg = x[1];
                    float glg[5], glb[5];
#pragma unroll
                    for (int ct = 0; ct < 5; ++ct) {
                        const float wk = r[ct] * inv_se;                              // responsibility of the mixture
                        const float pik = exp2_hw(fmaf(acc[ct][s2][0], L2E, -lse2));
                        const float gw_ = -coef * wk;                                 // d (-logsumexp) / d s_k
                        const float g1 = gw_ * gm[ct][1], g2 = gw_ * gm[ct][2];
                        float* dk = drow + 8 * (2 * ct + h);
                        float4 va = make_float4(coef * (pik - wk), gw_ * gm[ct][0], g1, g2);
                        float4 vb = make_float4(g1 * xr * (1.f - cf[ct][0] * cf[ct][0]), g2 * xr * (1.f - cf[ct][1] * cf[ct][1]),
                                                g2 * xg * (1.f - cf[ct][2] * cf[ct][2]), gw_ * gs[ct][0]);
                        // The row's eight products pass this statement as single registers (round 4 found ~1e-7 of the stored values
                        // wrong, +-0 in lanes 32..63, run to run; round 5: every one of them the LOW result of a v_pk_mul_f32 that read
                        // the HIGH half of a register pair; wait states in front of the stores change nothing, scalar multiplies do —
                        // profiles/r05_head_store_hazard.txt.  The file is built without packed f32 at all; the statement keeps the
                        // rows scalar under any flags: GCPX_STORE_WAIT=0 + SLP vectorisation is the failing build)
#if GCPX_STORE_WAIT == 1
                        asm volatile("" : "+v"(va.x), "+v"(va.y), "+v"(va.z), "+v"(va.w), "+v"(vb.x), "+v"(vb.y), "+v"(vb.z), "+v"(vb.w));
#elif GCPX_STORE_WAIT == 2
                        asm volatile("s_nop 7" : "+v"(va.x), "+v"(va.y), "+v"(va.z), "+v"(va.w), "+v"(vb.x), "+v"(vb.y), "+v"(vb.z), "+v"(vb.w));
#endif
                        *reinterpret_cast<float4*>(dk) = va;
                        *reinterpret_cast<float4*>(dk + 4) = vb;
                        glg[ct] = gw_ * gs[ct][1];
                        glb[ct] = gw_ * gs[ct][2];
                    }
#if GCPX_STORE_WAIT == 1
                    asm volatile("" : "+v"(glg[0]), "+v"(glb[0]), "+v"(glg[1]), "+v"(glb[1]), "+v"(glg[2]), "+v"(glb[2]), "+v"(glg[3]), "+v"(glb[3]),
                                 "+v"(glg[4]), "+v"(glb[4]));
#endif
                    // packing.dlm_log_scale_slot: ct = 0, 1 -> lane group 2 h of tile 5, ct = 2, 3 -> 2 h + 1, ct = 4 -> slots 96 + 2 h
                    *reinterpret_cast<float4*>(drow + 80 + 8 * h) = make_float4(glg[0], glb[0], glg[1], glb[1]);
                    *reinterpret_cast<float4*>(drow + 84 + 8 * h) = make_float4(glg[2], glb[2], glg[3], glb[3]);
                    *reinterpret_cast<float2*>(drow + 96 + 2 * h) = make_float2(glg[4], glb[4]);
                    if (h == 0) {
#pragma unroll
                        for (int z = 100; z < 112; z += 4) *reinterpret_cast<float4*>(drow + z) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        }
        if constexpr (WN) {
            // @phase nll_reduce matched
            // the item's 64 pixels: lanes q = 0, 1 (16 columns each), two rows s2 per lane -> one value per item
            float v = q < 2 ? nll_item : 0.f;
            v = row16_sum_dpp(v);
            const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
            v = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);      // lane groups 0 + 1 (2, 3 hold zeros)
            const int it_in_f = (y0 >> 2) * ncb + (x0 >> 4);
            if (lane == 0) a.nll_partial[(size_t)it_in_f * a.nll_rows + orow] = v;
        }
    };
    // @phase finish_swap item
    auto epilogue = [&]() __attribute__((always_inline)) {
        const int f = p_f, y0 = p_y0, x0 = p_x0, orow = p_orow;
        const bool store_raw = NLL == 0 && (mode == GCPX_HEAD_RAW || mode == GCPX_HEAD_DLM_BOTH) && orow >= 0;
        // (the coefficient lanes: odd lane groups before the row swap)
        finish_tiles<0, 5, FOLD>(a, bias_l, acc, p_inv, (q & 1) ? p_inv * (2.f * L2E) : p_inv, store_raw, orow, y0, x0, j, q);
        if (mode == GCPX_HEAD_RAW) return;
        // kernel channel order and the lane exchange: see conv3x3_head_kernel (conv3x3.hip)
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
        if (NLL && orow >= 0) mixtures(std::true_type{}, f, y0, x0, orow);      // @callsite matched
        else mixtures(std::false_type{}, f, y0, x0, orow);                      // @callsite unmatched
    };

    // @phase bookkeeping item
    Pos cur = nxt;
    for (int it = 0; it <= n_iter; ++it) {                         // (one extra trip: the last item's epilogue — ONE copy of that code)
        const int item = item0 + it * istep;
        const bool valid = it < n_iter && item < item_end;
        // ======== VALU half-phase: epilogue of the previous item, staging of this one ========
        if (p_valid) epilogue();
        int f = 0, y0 = 0, x0 = 0, orow = -1;
        float inv = 1.f;
        if (valid) {
            cur = nxt;
            f = cur.f; y0 = cur.strip * 4; x0 = cur.cb * 16;
            orow = __builtin_amdgcn_readfirstlane(pre_orow);
            // @phase staging item
            // ---- staging: BatchNorm affine + LeakyReLU of the producer, the item's power-of-two scale, the two f16 pieces ----
            float amax = 0.f;
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                // (the zero padding of the conv stays exactly zero: rows outside the frame are wave-uniform, halo columns per lane)
                const bool rv = k == RH ? true : (!(k == 0 && y0 == 0) && !(k == RH - 1 && y0 + 4 == H));
                if (rv) {
                    float4 v = pre[k];
                    const f32x2 v01 = pk_fma(pk2(v.x, v.y), pk2(bn_s.x, bn_s.y), pk2(bn_t.x, bn_t.y));
                    const f32x2 v23 = pk_fma(pk2(v.z, v.w), pk2(bn_s.z, bn_s.w), pk2(bn_t.z, bn_t.w));
                    const f32x2 l01 = v01 * pk1(slope), l23 = v23 * pk1(slope);
                    v.x = fmaxf(v01.x, l01.x); v.y = fmaxf(v01.y, l01.y); v.z = fmaxf(v23.x, l23.x); v.w = fmaxf(v23.y, l23.y);
                    if (k == RH) {
                        const bool ok = halo_ok(y0, x0);
                        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
                    }
                    pre[k] = v;
                }
                amax = vmax3abs(amax, pre[k].x, pre[k].y);
                amax = vmax3abs(amax, pre[k].z, pre[k].w);
            }
            amax = wave_max_nonneg(amax);
            // amax 2^ex in [2^14, 2^15): below the f16 maximum, and every piece that matters is a normal f16
            int ex = 14 + 127 - (int)((__float_as_uint(amax) >> 23) & 0xff);
            ex = amax > 0.f ? max(-100, min(min(100, 126 - ew), ex)) : 0;   // (2^-(ex + ew) stays a normal f32)
            const float sx2 = __uint_as_float((unsigned)(127 + ex) << 23);
#pragma unroll
            for (int k = 0; k < RH; ++k) {
                h4 p1, p2;
                split4(pre[k], sx2, p1, p2);
                *reinterpret_cast<h4*>(lds_i + k * (RW * 32)) = p1;
                *reinterpret_cast<h4*>(lds_i + k * (RW * 32) + Cfg::PLANE_BYTES) = p2;
            }
            if (lane < 48) {
                h4 p1, p2;
                split4(pre[RH], sx2, p1, p2);
                *reinterpret_cast<h4*>(lds_h) = p1;
                *reinterpret_cast<h4*>(lds_h + Cfg::PLANE_BYTES) = p2;
            }
            inv = __uint_as_float((unsigned)(127 - ex - ew) << 23);      // undoes the two power-of-two scales (exact)
        }
        // @phase targets item
        if (NLL && valid && orow >= 0) {
            // the target pixels of this item's likelihood: requested now, read in the epilogue one full MFMA half-phase later
            const float* tp = a.nll_target + (size_t)orow * 3 * plane + (size_t)(y0 + 2 * (q & 1)) * W + (x0 + j);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int c = 0; c < 3; ++c) tx[s2][c] = tp[c * plane + (size_t)s2 * W];
        } else if (NLL) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int c = 0; c < 3; ++c) tx[s2][c] = 0.f;
        }
        flush_images();
        pend_ip = nullptr;
        // @phase bookkeeping item
        if (it + 1 < n_iter && item + istep < item_end) {                               // in flight during this item's MFMAs
            advance(nxt);
            issue_loads(nxt);
        }
        // @phase mfma_tiles56_stash matched
        // ======== MFMA half-phase ========
        if (valid) {
            const bool want_nll = NLL && orow >= 0;
            // fused likelihood, first half: the green / blue log-scales (channel tiles 5, 6).  A lane of the epilogue owns pixel
            // (row s2 + 2 (q & 1), column j) and mixtures 2 ct + (q >> 1), ct = 0..4.  (The kernel sits at the 256-register limit of
            // two wavefronts per SIMD: of the 20 log-scales a lane needs in the epilogue, 16 wait in a wave-private LDS stash and 4 in
            // registers — 22 spilled registers otherwise, and every scratch reload waits for all loads in flight.)
            if (want_nll) {
                f32x4 accb[2][4];
                mfma_tiles<5, 2>(wl, reg, tapoff, lane, accb);
                finish_tiles<5, 2>(a, bias_l, accb, inv, inv, false, orow, y0, x0, j, q);
                // the same row swap as for tiles 0..4: afterwards accb[c][s2] holds lane group q' = 2 (q >> 1) and accb[c][s2 + 2]
                // lane group q' + 1 of THIS lane's pixel.  Slot layout of tiles 5, 6: packing.dlm_log_scale_slot
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(accb[c][s2][r]), __float_as_uint(accb[c][s2 + 2][r]), false, false);
                            accb[c][s2][r] = __uint_as_float(sw[0]);
                            accb[c][s2 + 2][r] = __uint_as_float(sw[1]);
                        }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const f32x4 e5 = accb[0][s2], o5 = accb[0][s2 + 2], e6 = accb[1][s2];
                    // {ls_g, ls_b} of ct = 0, 1 (e5) and ct = 2, 3 (o5) of this lane's mixtures: to the stash
                    stash[(2 * s2) * 64] = make_float4(e5[0], e5[1], e5[2], e5[3]);
                    stash[(2 * s2 + 1) * 64] = make_float4(o5[0], o5[1], o5[2], o5[3]);
                    // mixtures 8, 9 sit in lane group 0 of tile 6 (slots 96..99): lanes 0..31 keep registers 0, 1 (mixture 8), lanes
                    // 32..63 take registers 2, 3 of lane - 32 (mixture 9) — v_permlane32_swap puts the source's lower half there
                    const auto g = __builtin_amdgcn_permlane32_swap(__float_as_uint(e6[0]), __float_as_uint(e6[2]), false, false);
                    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(e6[1]), __float_as_uint(e6[3]), false, false);
                    ls4[s2][0] = __uint_as_float(g[0]);
                    ls4[s2][1] = __uint_as_float(b[0]);
                }
            }

            // @phase mfma_passA item
            // pass A: channel tiles 0..4 = the 80 slots the mixture mean reads; their epilogue runs in the next VALU half-phase
            mfma_tiles<0, 5>(wl, reg, tapoff, lane, acc);
            // pass B for the stored-parameters modes: channel tiles 5, 6 = slots 80..99 (+ 12 empty), only ever stored raw
            if (NLL == 0 && (mode == GCPX_HEAD_RAW || mode == GCPX_HEAD_DLM_BOTH) && orow >= 0) {
                f32x4 accr[2][4];
                mfma_tiles<5, 2>(wl, reg, tapoff, lane, accr);
                finish_tiles<5, 2>(a, bias_l, accr, inv, inv, true, orow, y0, x0, j, q);
            }
        } else {
            // @phase invalid_item rare
            // (every path through this half-phase defines the accumulators: otherwise they count as live across the staging and the
            // prefetch of the next item — 80 registers — and the prefetch addresses get spilled instead)
#pragma unroll
            for (int c = 0; c < 5; ++c)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[c][pt] = f32x4{0, 0, 0, 0};
            ls4[0][0] = ls4[0][1] = ls4[1][0] = ls4[1][1] = 0.f;
        }
        // @phase bookkeeping item
        p_valid = valid; p_f = f; p_y0 = y0; p_x0 = x0; p_orow = orow; p_inv = inv;
    }
    flush_images();
}

}  // namespace

int gcpx_launch_head32(const gcpx_conv_args* a, hipStream_t stream);            // conv3x3_head32.hip

// Called by conv3x3_dispatch (conv3x3.hip) for the 100-channel mixture head when the caller supplies split-f16 weights.
int gcpx_launch_head_split(const gcpx_conv_args* a, hipStream_t stream) {
    using Cfg = SplitHeadCfg;
    if (a->split_layout == GCPX_SPLIT_HEAD32) return gcpx_launch_head32(a, stream);
    const int nll = a->head_mode == GCPX_HEAD_DLM_NLL ? 1 : (a->head_mode == GCPX_HEAD_DLM_NLL_GRAD ? 2 : 0);
    if (nll) {
        GCPX_CHECK_ARG(a->nll_target && a->nll_partial && a->nll_rows > 0 && a->raw_row_map, "GCPX_HEAD_DLM_NLL needs nll_target, nll_partial, nll_rows and raw_row_map");
        GCPX_CHECK_ARG(a->out_pitch == 112, "the fused likelihood is written for the 112-slot layout of the 10-mixture head");
        GCPX_CHECK_ARG(nll == 1 || a->out, "GCPX_HEAD_DLM_NLL_GRAD writes the parameter gradient to `out`");
    }
    typedef void (*kern_t)(const gcpx_conv_args, const int, const int);
    static const kern_t kerns[3] = {conv3x3_head_split_kernel<0>, conv3x3_head_split_kernel<1>, conv3x3_head_split_kernel<2>};
    const int lds = nll ? Cfg::LDS_BYTES_NLL : Cfg::LDS_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        for (int i = 0; i < 3; ++i) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[i]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               i ? Cfg::LDS_BYTES_NLL : Cfg::LDS_BYTES);
            if (e != hipSuccess) {
                gcpx_set_error("conv3x3 split head: hipFuncSetAttribute(%d B LDS): %s", i ? Cfg::LDS_BYTES_NLL : Cfg::LDS_BYTES, hipGetErrorString(e));
                return GCPX_ERR_HIP;
            }
        }
        attr_set = true;
    }
    const int nitems = a->F * (a->Hout / 4) * (a->Wout / 16);
    int grid = gcpx_conv_grid() / 2;
    if (grid * 8 > nitems) grid = (nitems + 7) / 8;
    const int ipw = (nitems + grid * 8 - 1) / (grid * 8);
    hipLaunchKernelGGL(kerns[nll], dim3(grid), dim3(512), lds, stream, *a, ipw, nitems);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
