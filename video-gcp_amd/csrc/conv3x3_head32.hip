// Output head of the decoder (gen_head 16 -> 100 channels @ full resolution + distr.mean + decoder.nll:
// /root/reference/gcp/prediction/models/tree/tree_dense_rec.py:42, frame_binding.py:88-99) in split-f16 arithmetic (split_mfma.h) on
// 32x32x16 MFMA tiles, with the epilogue of one half-item issued UNDER the MFMA pass of the next.
//
// STATUS: parity-green, measured NOT faster than the round-5 kernel (conv3x3_head_split.hip) and therefore opt-in (GCPX_HEAD32=1):
// profiles/r06_head32_study.txt has the probes, the two forms' timings and the PMC comparison.
//
// Why it was built (tools/r06/issue_probe, profiles/r06_issue_probe.txt): inside one wavefront 5 plain VALU instructions (or 4 + one
// transcendental) issue for free between two v_mfma_f32_32x32x16_f16 (13.6 ns each) but only 1 between two v_mfma_f32_16x16x32_f16
// (7.6 ns).  The round-5 kernel (16x16x32 tiles, an MFMA phase and a VALU phase per item) pays matrix time + VALU time.  Two forms:
//   PIPE = true   software pipeline inside the wavefront: while the 81 / 108 MFMAs of half-item n + 1 run, the wavefront's VALU slots
//                 carry the mixture mean / likelihood of half-item n and the staging arithmetic of the item after it (yield sites,
//                 MfmaStream below); two accumulator sets -> one wavefront per SIMD (512 registers).
//   PIPE = false  MFMA pass, then epilogue, per half-item, two wavefronts per SIMD.
// Why it does not pay: a matched item carries ~1650 VALU instructions for 216 MFMAs — 7.6 per MFMA where the pipe hides 5 — and with one
// wavefront per SIMD every LDS / dependency stall of the stream is exposed.
//
// Layout.  An item is 4 rows x 16 pixels of a frame, two half-items of 2 x 16 = 32 pixels = the N side of a 32x32 tile.  K = 16 input
// channels = ONE tap per MFMA: 9 k-steps, no padded tenth tap.  M = 32 channel slots per tile, 4 tiles (3 for frames without a
// likelihood).  D layout of the 32x32 tile: lane l holds rows 8 g + 4 (l >> 5) + r of column l & 31, i.e. 16 slots of ONE pixel per
// tile and lane; the slot -> channel map (head32_slot, packing.head32_slot) gives lane half h the mixtures k = 2 m + h, m = 0..4, of
// that pixel with all their parameters in its own registers: no row swap, no lane exchange except the sums over a pixel's lane pair.
// The 112-slot layout of the stored gradient rows (training variant) is the round-5 kernel's (packing.dlm_channel_perm).
//
// One wavefront per item at a time, weights resident in LDS ([9][4][2][64] x 16 B = 72 KiB), the item's haloed 6 x 18 x 16-channel
// region as two f16 planes with a 592 B row pitch (32 B per pixel + 16: the two pixel rows of a B fragment land on disjoint banks).
#include "common.h"
#include "split_mfma.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float L2E = 1.44269504088896341f, LN2 = 0.69314718055994531f;

struct Head32Cfg {
    static constexpr int CT = 4, KS = 9;
    static constexpr int RW = 18, RH = 6;
    static constexpr int ROW_BYTES = RW * 32 + 16;                         // 592
    static constexpr int PLANE_BYTES = RH * ROW_BYTES;                     // 3552
    static constexpr int REGION_BYTES = 2 * PLANE_BYTES;                   // 7104 per wavefront
    static constexpr int W_BYTES = KS * CT * 2 * 1024;                     // 73728
    static constexpr int BIAS_BYTES = 2 * 64 * 4;                          // per lane half: 64 slots
    static constexpr int lds_bytes(int nw) { return W_BYTES + nw * REGION_BYTES + BIAS_BYTES; }
};

// row i of tile c -> slot of the 112-slot layout, -1 = empty (packing.head32_slot)
__host__ __device__ constexpr int head32_slot(const int c, const int i) {
    const int g = i / 8, h = (i / 4) % 2, r = i % 4;
    if (c < 3) {
        const int m = 2 * c + g / 2, p = 4 * (g % 2) + r;
        return m < 5 ? 8 * (2 * m + h) + p : -1;
    }
    const int u = 4 * g + r;
    return u < 10 ? dlm_ls_slot(1 + u % 2, 2 * (u / 2) + h) : -1;
}

__device__ __forceinline__ f32x16 mfma32x32(const h8 a, const h8 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float exp2_hw(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float log2_hw(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float rcp_hw(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoid_fast_s(float x) { return rcp_hw(1.f + exp_hw(-x)); }
__device__ __forceinline__ float softplus_s(float x) { return x > 20.f ? x : log_hw(1.f + exp_hw(x)); }

// ---- how the VALU work gets under the MFMAs ----
// The VALU code of a block (epilogue, staging arithmetic) is written in program order with numbered YIELD SITES every ~15 instructions.
// A site emits the next unit(s) of the block's MFMA stream (one unit = the 3 MFMAs of a (tap, tile) + the LDS reads of the next unit's
// fragments), a sched_group_barrier pattern "1 MFMA, PER VALU" over the region since the previous site, and a sched_barrier fence.
// Site s of TOTAL emits units [s NU / TOTAL, (s + 1) NU / TOTAL): the stream is spread evenly over the block's sites whatever the two
// counts are.  Why not one pattern over the whole block: hipcc (ROCm 7.2) builds such a pipeline only for some blocks — the same
// epilogue is interleaved in one place and left behind the MFMAs in another, a DS-read group or per-class VALU / transcendental
// groups make it drop the pattern altogether (tools/r06/sgb_toy.hip); fenced regions of 3-9 MFMAs come out the same everywhere.
// PER = 6: one more than the pipe hides for free (tools/r06/issue_probe: 5 fillers cost nothing, the 6th-8th ~1.4 ns each, an
// exposed VALU instruction 2.1 ns) because the blocks carry more VALU work than 5 per MFMA.
#ifndef HEAD32_PER
#define HEAD32_PER 6
#endif
template <int NT, int PT, int TOTAL, int PER = HEAD32_PER>
struct MfmaStream {
    using Cfg = Head32Cfg;
    static constexpr int NU = Cfg::KS * NT;
    const char* wl_lane;        // weights + lane * 16
    const char* reg_lane;       // region + this lane's pixel / channel-half offset
    f32x16 (&acc)[4];
    // weight fragments WD units ahead (a ring of WD + 1), activation fragments one tap = NT units ahead.  One unit ahead is not
    // enough: a unit is ~100 cycles of matrix pipe, an LDS read under this kernel's load takes longer, and while one wavefront of
    // a SIMD is in its MFMA pass nobody else feeds the pipe (profiles/r06_head_pmc.txt: 23 % of the wave cycles parked in s_waitcnt)
    static constexpr int WD = 2;
    h8 w[WD + 1][2], b[2][2];

    __device__ __forceinline__ MfmaStream(const char* wl, const char* rg, f32x16 (&a)[4]) : wl_lane(wl), reg_lane(rg), acc(a) {
        if constexpr (NU > 0) {
            load_b(0, b[0]);
            static_for<0, WD>([&](auto i) __attribute__((always_inline)) {
                constexpr int U = decltype(i)::value;
                if constexpr (U < NU) load_w(U / NT, U % NT, w[U % (WD + 1)]);
            });
        }
    }
    __device__ __forceinline__ void load_w(const int t, const int c, h8 (&d)[2]) {
        const char* p = wl_lane + ((t * Cfg::CT + c) * 2) * 1024;
        d[0] = *reinterpret_cast<const h8*>(p);
        d[1] = *reinterpret_cast<const h8*>(p + 1024);
    }
    __device__ __forceinline__ void load_b(const int t, h8 (&d)[2]) {
        const char* p = reg_lane + (2 * PT + t / 3) * Cfg::ROW_BYTES + (t % 3) * 32;
        d[0] = *reinterpret_cast<const h8*>(p);
        d[1] = *reinterpret_cast<const h8*>(p + Cfg::PLANE_BYTES);
    }
    template <int U>
    __device__ __forceinline__ void unit() {
        constexpr int t = U / NT, c = U % NT;
#ifdef ABL_NOMFMA                      // (timing ablation: the block's VALU work alone)
        return;
#endif
        if constexpr (U + WD < NU) load_w((U + WD) / NT, (U + WD) % NT, w[(U + WD) % (WD + 1)]);
        if constexpr (c == 0 && t + 1 < Cfg::KS) load_b(t + 1, b[(t + 1) & 1]);
        const h8 w1 = w[U % (WD + 1)][0], w2 = w[U % (WD + 1)][1], b1 = b[t & 1][0], b2 = b[t & 1][1];
        // small terms first: they are added to the accumulator while it is still small
        if constexpr (t == 0) acc[c] = mfma32x32(w2, b1, f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0});
        else acc[c] = mfma32x32(w2, b1, acc[c]);
        acc[c] = mfma32x32(w1, b2, acc[c]);
        acc[c] = mfma32x32(w1, b1, acc[c]);
    }
    // yield site S (0 <= S < TOTAL, in program order)
    template <int S>
    __device__ __forceinline__ void operator()(std::integral_constant<int, S>) {
        static_assert(S >= 0 && S < TOTAL, "yield site out of range");
        constexpr int lo = S * NU / TOTAL, hi = (S + 1) * NU / TOTAL;
        if constexpr (hi > lo) {
            static_for<lo, hi>([&](auto u) __attribute__((always_inline)) { unit<decltype(u)::value>(); });
            if constexpr (PER > 0) {
                static_for<0, 3 * (hi - lo)>([&](auto) __attribute__((always_inline)) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x402, PER, 0);           // VALU | transcendental
                });
            } else {
                // a pass on its own (PER = 0, the two-phase kernel): the next unit's fragment reads ahead of every unit's three MFMAs
                static_for<lo, hi>([&](auto u) __attribute__((always_inline)) {
                    constexpr int U = decltype(u)::value;
                    constexpr int nld = (U + WD < NU ? 2 : 0) + ((U % NT) == 0 && U / NT + 1 < Cfg::KS ? 2 : 0);
                    if constexpr (nld > 0) __builtin_amdgcn_sched_group_barrier(0x100, nld, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};
// (the exact path's second run of an epilogue carries no MFMA stream)
struct NoYield {
    template <int S>
    __device__ __forceinline__ void operator()(std::integral_constant<int, S>) {}
};
#define YIELD(y, s) y(std::integral_constant<int, (s)>{})

// yield sites of the epilogue (EK = 1: mixture mean, EK = 2: + likelihood) and of the staging arithmetic
__host__ __device__ constexpr int epi_sites(const int ek) { return ek == 2 ? 3 + 5 * 5 + 3 : (ek == 1 ? 2 + 5 + 1 : 1); }
constexpr int STAGE_SITES = 11;

// What an epilogue leaves behind for the code between the blocks
struct HalfOut {
    float px[3];            // the pixel's mixture mean (both lanes of the pixel's pair hold it)
    float nll;              // -log likelihood of the pixel (matched frames; both lanes hold it)
};

// NLL = 1: forward with the likelihood of the matched frames in the epilogue (GCPX_HEAD_DLM_NLL; GCPX_HEAD_DLM_MEAN runs it with no
// matched frame).  NLL = 2: training forward, additionally the gradient rows (GCPX_HEAD_DLM_NLL_GRAD).  NW: wavefronts per workgroup
// (8 = two per SIMD with 256 registers each, 4 = one per SIMD with 512).
// PIPE: true = the software-pipelined form above (epilogue of half-item n under the MFMA pass of n + 1, two accumulator sets);
// false = two phases per half-item (MFMA pass, then its epilogue; one accumulator set) — the two wavefronts of a SIMD overlap each
// other's phases instead (tools/r06/cross_wave_probe: a wavefront's VALU work issues under the OTHER wavefront's 32x32x16 MFMAs).
template <int NLL, int NW, bool PIPE>
__global__ void __launch_bounds__(NW * 64, NW / 4) conv3x3_head32_kernel(const gcpx_conv_args a, const int nitems, const int fused_nll) {
    using Cfg = Head32Cfg;
    constexpr int RW = Cfg::RW, RH = Cfg::RH, ROWB = Cfg::ROW_BYTES;
    constexpr int NSL = 7;                                                          // staging slots per lane: six region rows + the halo columns
    extern __shared__ float4 smem4[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* wl = reinterpret_cast<const char*>(smem4);                          // [KS][CT][2][64] x 16 B
    char* reg = reinterpret_cast<char*>(smem4) + Cfg::W_BYTES + wave * Cfg::REGION_BYTES;
    float* bias_all = reinterpret_cast<float*>(reinterpret_cast<char*>(smem4) + Cfg::W_BYTES + NW * Cfg::REGION_BYTES);
    const int hh = lane >> 5, j32 = lane & 31, prow = j32 >> 4, pcol = j32 & 15;
    const int H = a.Hout, W = a.Wout;
    const int ncb = W / 16, nrp = H / 4;

    for (int i = tid; i < Cfg::W_BYTES / 16; i += NW * 64) smem4[i] = reinterpret_cast<const float4*>(a.wpk_split)[i];
    if (tid < 128) {
        // bias of lane half h, register u = 16 c + 4 g + r <-> tile c, row 8 g + 4 h + r; the colour coefficients (slots 8 k + 4 .. 6)
        // carry 2 log2(e): their scale-back fma feeds the tanh's v_exp_f32 directly
        const int h = tid >> 6, u = tid & 63, c = u >> 4, g = (u >> 2) & 3, r = u & 3;
        const int slot = head32_slot(c, 8 * g + 4 * h + r);
        float v = slot >= 0 ? a.bias[slot] : 0.f;
        if (slot >= 0 && slot < 80 && (slot & 7) >= 4 && (slot & 7) < 7) v *= 2.f * L2E;
        bias_all[tid] = v;
    }
    __syncthreads();
    const float* bias_l = bias_all + hh * 64;
    const char* wl_lane = wl + lane * 16;
    const char* reg_lane = reg + prow * ROWB + pcol * 32 + hh * 16;

    const gcpx_conv_src sr = a.src[0];
    const int ew = a.w_split_log2_dev ? __builtin_amdgcn_readfirstlane(*a.w_split_log2_dev) : a.w_split_log2;
    float4 bn_s = make_float4(1.f, 1.f, 1.f, 1.f), bn_t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sr.scale) {
        bn_s = *reinterpret_cast<const float4*>(sr.scale + (lane & 3) * 4);
        bn_t = *reinterpret_cast<const float4*>(sr.shift + (lane & 3) * 4);
    }
    const float slope = sr.act == GCPX_ACT_LRELU ? 0.2f : 1.f;                        // LeakyReLU as max(v, slope v) (slope <= 1)

    // ---- staging map (as conv3x3_head_split.hip): slot k < 6 = region row k, region column 1 + (lane >> 2), channels 4 (lane & 3) .. + 3 —
    // a source row is one contiguous 1 KiB read at lane * 16 B; slot 6 (lanes 0..47) = region row lane >> 3, column 0 / 17 ----
    const int hrow = lane >> 3, hside = (lane >> 2) & 1;
    const int hoff = (hrow * W + (hside ? 16 : -1)) * 16 + (lane & 3) * 4;           // floats from the item's row pointer
    char* const lds_i = reg + 32 + lane * 8;
    char* const lds_h = reg + hrow * ROWB + hside * 17 * 32 + (lane & 3) * 8;

    // items are dealt round-robin over the wavefronts: every wavefront sees the same mix of frames with / without a likelihood
    const int gw = blockIdx.x * NW + wave;
    const int istep = (int)gridDim.x * NW;
    const int n_iter = (nitems - gw + istep - 1) / istep;                           // items of this wavefront (>= 0)
    struct Pos { int f, cb, strip; };
    const int st_strip = istep % nrp, st_cb = (istep / nrp) % ncb, st_f = istep / nrp / ncb;
    auto advance = [&](Pos& p) __attribute__((always_inline)) {
        p.strip += st_strip;
        int c = p.strip >= nrp;
        p.strip -= c ? nrp : 0;
        p.cb += st_cb + c;
        c = p.cb >= ncb;
        p.cb -= c ? ncb : 0;
        p.f += st_f + c;
    };
    Pos nxt;
    nxt.strip = gw % nrp; nxt.cb = (gw / nrp) % ncb; nxt.f = gw / nrp / ncb;

    float4 pre[NSL];                                        // raw activations of the item whose loads are in flight
    int pre_orow = -1;
    auto halo_ok = [&](const int y0, const int x0) __attribute__((always_inline)) {
        return lane < 48 && (unsigned)(y0 - 1 + hrow) < (unsigned)H && (unsigned)(x0 + (hside ? 16 : -1)) < (unsigned)W;
    };
    auto issue_loads = [&](const Pos& p) __attribute__((always_inline)) {
        const int y0 = p.strip * 4, x0 = p.cb * 16;
        const float* rowp = sr.ptr + (((long long)p.f * H + (y0 - 1)) * W + x0) * 16;   // (row -1 of the frame: formed, never dereferenced)
        pre_orow = a.raw_row_map ? a.raw_row_map[p.f] : -1;
#pragma unroll
        for (int k = 0; k < RH; ++k) {
            const bool rv = !(k == 0 && y0 == 0) && !(k == RH - 1 && y0 + 4 == H);
            pre[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rv) pre[k] = *reinterpret_cast<const float4*>(rowp + (long long)k * W * 16 + lane * 4);
        }
        pre[RH] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (halo_ok(y0, x0)) pre[RH] = *reinterpret_cast<const float4*>(rowp + hoff);
    };

    // the staged item: its two f16 pieces per slot, waiting in registers for the region to be free
    h4 pc1[NSL], pc2[NSL];
    float st_inv = 1.f;
    // staging arithmetic (BatchNorm affine + LeakyReLU of the producer, the item's power-of-two scale, the two f16 pieces) — branch free:
    // it rides in the VALU slots of an MFMA pass
    auto stage_math = [&](const int y0, const int x0, auto&& Y, auto base_tag) __attribute__((always_inline)) {
        constexpr int YB = decltype(base_tag)::value;      // first yield site of the staging arithmetic in its block
        float amax = 0.f;
        const bool top = y0 == 0, bot = y0 + 4 == H, hok = halo_ok(y0, x0);
        static_for<0, NSL>([&](auto kt) __attribute__((always_inline)) {
            constexpr int k = decltype(kt)::value;
            float4 v = pre[k];
            v.x = fmaf(v.x, bn_s.x, bn_t.x); v.y = fmaf(v.y, bn_s.y, bn_t.y); v.z = fmaf(v.z, bn_s.z, bn_t.z); v.w = fmaf(v.w, bn_s.w, bn_t.w);
            v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope); v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
            // the zero padding of the conv stays exactly zero: rows outside the frame are wave-uniform, halo columns per lane
            const bool keep = k == RH ? hok : !((k == 0 && top) || (k == RH - 1 && bot));
            v.x = keep ? v.x : 0.f; v.y = keep ? v.y : 0.f; v.z = keep ? v.z : 0.f; v.w = keep ? v.w : 0.f;
            pre[k] = v;
            amax = vmax3abs(amax, v.x, v.y);
            amax = vmax3abs(amax, v.z, v.w);
            YIELD(Y, YB + k);
        });
        amax = wave_max_nonneg(amax);
        // amax 2^ex in [2^14, 2^15): below the f16 maximum, and every piece that matters is a normal f16
        int ex = 14 + 127 - (int)((__float_as_uint(amax) >> 23) & 0xff);
        ex = amax > 0.f ? max(-100, min(min(100, 126 - ew), ex)) : 0;       // (2^-(ex + ew) stays a normal f32)
        const float sx2 = __uint_as_float((unsigned)(127 + ex) << 23);
        YIELD(Y, YB + 7);
        static_for<0, NSL>([&](auto kt) __attribute__((always_inline)) {
            constexpr int k = decltype(kt)::value;
            split4(pre[k], sx2, pc1[k], pc2[k]);
            if constexpr (k == 1) YIELD(Y, YB + 8);
            if constexpr (k == 3) YIELD(Y, YB + 9);
        });
        st_inv = __uint_as_float((unsigned)(127 - ex - ew) << 23);          // undoes the two power-of-two scales (exact)
        YIELD(Y, YB + 10);
    };
    auto write_region = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < RH; ++k) {
            *reinterpret_cast<h4*>(lds_i + k * ROWB) = pc1[k];
            *reinterpret_cast<h4*>(lds_i + k * ROWB + Cfg::PLANE_BYTES) = pc2[k];
        }
        if (lane < 48) {
            *reinterpret_cast<h4*>(lds_h) = pc1[RH];
            *reinterpret_cast<h4*>(lds_h + Cfg::PLANE_BYTES) = pc2[RH];
        }
    };

    const size_t plane = (size_t)H * W;
    // ======== epilogue of ONE pixel per lane: mixture mean (EK >= 1) + likelihood (EK == 2) of mixtures k = 2 m + hh ========
    // acc: the half-item's accumulators; x: the target pixel (EK == 2); (y, xcol): the pixel.  The reference's vanishing-bin branch is a
    // rarely taken wave-level branch between two yield sites: the sites fence the schedule, a branch between them costs nothing
    // branch inline (the rare second run of a half-item whose main-path run flagged a lane)
    auto epilogue = [&](auto ek_tag, const f32x16 (&acc)[4], const float inv, const float (&x)[3], const int orow,
                        const int y, const int xcol, HalfOut& out, auto&& Y) __attribute__((always_inline)) {
#ifdef ABL_NOVALU                      // (timing ablation: the MFMA streams alone)
        constexpr int EK = 0;
#else
        constexpr int EK = decltype(ek_tag)::value;
#endif
        constexpr bool WN = EK == 2;
        out.nll = 0.f;
        if constexpr (EK == 0) {
            out.px[0] = out.px[1] = out.px[2] = 0.f;
            static_for<0, epi_sites(decltype(ek_tag)::value)>([&](auto st) __attribute__((always_inline)) { YIELD(Y, decltype(st)::value); });
            // (the accumulators count as used: the MFMA streams of the timing ablation stay)
            asm volatile("" : : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
            return;
        }
        const float inv3 = inv * (2.f * L2E);
        // scale back + bias, lazily: the five logits now (the softmax needs them first), a mixture's other parameters right before its
        // use — the accumulators of a tile die as its mixtures complete, nothing of the half-item stays live to the end of the block
        float lg[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) lg[m] = fmaf(acc[m >> 1][8 * (m & 1)], inv, bias_l[16 * (m >> 1) + 8 * (m & 1)]);
        YIELD(Y, 0);
        // ---- softmax weights of this lane's five mixtures (the other five: lane ^ 32) ----
        float mx = vmax(lg[0], lg[1]);
        mx = vmax(mx, lg[2]); mx = vmax(mx, lg[3]); mx = vmax(mx, lg[4]);
        mx = pair_max(mx);
        const float mL = mx * L2E;
        float w[5], S = 0.f;
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            w[m] = exp2_hw(fmaf(lg[m], L2E, -mL));
            S += w[m];
        }
        S = pair_sum(S);
        YIELD(Y, 1);
        float xp[3] = {0.f, 0.f, 0.f}, lp2[5], lse2 = 0.f;
        bool lo[3] = {false, false, false}, hi[3] = {false, false, false};
        constexpr int NG = NLL == 2 ? 5 : 1;           // gradient bookkeeping only in the training variant
        float gm[NG][3], gs[NG][3], cf[NG][3];
        if constexpr (WN) {
            lse2 = mL + log2_hw(S);                     // log2 sum exp of all ten logits
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                xp[c] = x[c] + 1.f / 255.f;
                lo[c] = x[c] < -0.999f; hi[c] = x[c] > 0.999f;
            }
            YIELD(Y, 2);
        }
        float Sr = 0.f, Sg = 0.f, Sb = 0.f;
        constexpr int S_MIX = WN ? 3 : 2, PER_MIX = WN ? 5 : 1;
        static_for<0, 5>([&](auto mt) __attribute__((always_inline)) {
            constexpr int m = decltype(mt)::value;
            constexpr int SB = S_MIX + PER_MIX * m;                      // first yield site of this mixture
            // {logit, mean r, g, b, coeff 0, 1, 2 (x 2 log2 e), log-scale r} and the green / blue log-scales of mixture 2 m + hh
            constexpr int tc = m >> 1, r0 = 8 * (m & 1);
            const float4 b0 = *reinterpret_cast<const float4*>(bias_l + 16 * tc + r0);
            const float4 b1 = *reinterpret_cast<const float4*>(bias_l + 16 * tc + r0 + 4);
            float e[8];
            e[0] = lg[m]; e[1] = fmaf(acc[tc][r0 + 1], inv, b0.y); e[2] = fmaf(acc[tc][r0 + 2], inv, b0.z); e[3] = fmaf(acc[tc][r0 + 3], inv, b0.w);
            e[4] = fmaf(acc[tc][r0 + 4], inv3, b1.x); e[5] = fmaf(acc[tc][r0 + 5], inv3, b1.y); e[6] = fmaf(acc[tc][r0 + 6], inv3, b1.z);
            e[7] = WN ? fmaf(acc[tc][r0 + 7], inv, b1.w) : 0.f;
            float lsg_m = 0.f, lsb_m = 0.f;
            if constexpr (WN) {
                const float2 bl = *reinterpret_cast<const float2*>(bias_l + 48 + 2 * m);
                lsg_m = fmaf(acc[3][2 * m], inv, bl.x);
                lsb_m = fmaf(acc[3][2 * m + 1], inv, bl.y);
            }
            // tanh of the colour coefficients, 1 - 2 / (exp(2 x) + 1)
            const float q0 = exp2_hw(e[4]) + 1.f, q1 = exp2_hw(e[5]) + 1.f, q2 = exp2_hw(e[6]) + 1.f;
            const float c0 = fmaf(rcp_hw(q0), -2.f, 1.f), c1 = fmaf(rcp_hw(q1), -2.f, 1.f), c2 = fmaf(rcp_hw(q2), -2.f, 1.f);
            const float mr = e[1];
            const float mg = fmaf(c0, mr, e[2]), mb0 = fmaf(c1, mr, e[3]);
            const float mb = fmaf(c2, mg, mb0);
            Sr = fmaf(w[m], mr, Sr); Sg = fmaf(w[m], mg, Sg); Sb = fmaf(w[m], mb, Sb);
            YIELD(Y, SB);
            if constexpr (WN) {
                // ---- likelihood of mixture 2 m + hh at this lane's pixel (formulas and ranges: conv3x3_head_split.hip).  With
                // e_p = exp(-plus_in), e_m = exp(-min_in):  cdf_plus - cdf_min = (e_m - e_p) / ((1 + e_p)(1 + e_m));
                // x < -0.999: cdf_plus = 1 / (1 + e_p);  x > 0.999: 1 - cdf_min = e_m / (1 + e_m), i.e. e_p := 0.
                constexpr int SC = NLL == 1 ? 10 : 0;
                constexpr float S1 = NLL == 1 ? 9.765625e-4f : 1.f, S2 = S1 * S1;       // 2^-SC, 2^-2SC
                constexpr float LIM = 40.f - SC;
                const float mean[3] = {e[1], fmaf(c0, x[0], e[2]), fmaf(c2, x[1], fmaf(c1, x[0], e[3]))};
                const float lsr[3] = {e[7], lsg_m, lsb_m};
                float tl[3], num[3], den[3], lsc[3], cdq[3], ap[3], am[3], ep[3], em[3], Pp[3], Mm[3];
                bool bad[3];
                if constexpr (NLL == 2) { cf[m][0] = c0; cf[m][1] = c1; cf[m][2] = c2; }
                static_for<0, 3>([&](auto ctag) __attribute__((always_inline)) {
                    constexpr int c = decltype(ctag)::value;
                    lsc[c] = vmax(lsr[c], -7.f);
                    // log2(e) / scale straight out of v_exp_f32 (log2 of log2(e) folded into its argument)
                    tl[c] = exp2_hw(fmaf(lsc[c], -L2E, 0.52876637294f));
                    ap[c] = fmaf(-tl[c], xp[c] - mean[c], -(float)SC);
                    am[c] = fminf(fmaf(tl[c], 2.f / 255.f, ap[c]), LIM);
                    ep[c] = exp2_hw(ap[c]);
                    em[c] = exp2_hw(am[c]);
                    ep[c] = hi[c] ? 0.f : ep[c];
                    Pp[c] = ep[c] + S1; Mm[c] = em[c] + S1;                 // 2^-SC (1 + e_p), 2^-SC (1 + e_m)
                    den[c] = Pp[c] * Mm[c];
                    const float dif = em[c] - ep[c], thr = den[c] * (1e-5f / S1);
                    num[c] = lo[c] ? Mm[c] : dif;
                    // bin probability > 1e-5 (as the reference's branch); false for every non-finite or out-of-range term
                    bad[c] = !(num[c] > thr);
                    if constexpr (NLL == 2) {
                        const float rd = rcp_hw(den[c]);
                        const float sp = Mm[c] * rd, sm = lo[c] ? 0.f : Pp[c] * rd;
                        const float pp_ = ep[c] * sp * sp, pm_ = em[c] * sm * sm;       // s (1 - s) as e s^2: no cancellation
                        cdq[c] = num[c] * rd;
                        const float rcd = rcp_hw(cdq[c]);
                        const float plus_in = ap[c] * -LN2, min_in = am[c] * -LN2;
                        gm[m][c] = -(tl[c] * LN2) * (pp_ - pm_) * rcd;
                        gs[m][c] = -(plus_in * pp_ - min_in * pm_) * rcd;
                    }
                    YIELD(Y, SB + 1 + c);
                });
                float extra2 = 0.f;                                       // log2 terms of the lanes off the main path
                if (__any(bad[0] || bad[1] || bad[2])) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if (__any(bad[c])) {
                            // The reference's branch for vanishing bins: log(pdf at the bin centre) - log(127.5) = log(2/255 t s'(mid)),
                            // s'(mid) = g / (1 + g)^2 with g = exp(-|mid|) — numerator 2/255 t, denominator (1 + g)^2, -|mid| as a log2 term
                            const float xc = x[c] - mean[c], t = tl[c] * LN2;          // t = 1 / scale
                            const float u = fabsf(tl[c] * xc);
                            const float g = exp2_hw(-u);
                            const float A = fmaf(g, S1, S1);
                            float vn = t * (2.f / 255.f * S1), vd = A * A, ve = -u, dm = 0.f, ds = 0.f;
                            if constexpr (NLL == 2) {
                                const float hq = (1.f - g) * rcp_hw(1.f + g);      // |1 - 2 sigmoid(mid)|
                                dm = xc < 0.f ? -t * hq : t * hq;
                                ds = fmaf(u * LN2, hq, -1.f);
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
                                gm[m][c] = bad[c] ? dm : gm[m][c];
                                gs[m][c] = bad[c] ? ds : gs[m][c];
                            }
                        }
                    }
                }
                if constexpr (NLL == 2) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) gs[m][c] = lsr[c] < -7.f ? 0.f : gs[m][c];     // clamp(min=-7) blocks the gradient
                }
                float prod;
                if constexpr (NLL == 2) prod = cdq[0] * cdq[1] * cdq[2];
                else prod = (num[0] * num[1] * num[2]) * rcp_hw(den[0] * den[1] * den[2]);
                // log2 of (mixture weight x the three bin probabilities)
                lp2[m] = fmaf(e[0], L2E, log2_hw(prod)) + (extra2 - lse2 - 3.f * SC);
                YIELD(Y, SB + 4);
            }
        });
        constexpr int S_FIN = S_MIX + 5 * PER_MIX;
        Sr = pair_sum(Sr); Sg = pair_sum(Sg); Sb = pair_sum(Sb);
        const float invS = rcp_hw(S);
        out.px[0] = fminf(fmaxf(Sr * invS, -1.f), 1.f);
        out.px[1] = fminf(fmaxf(Sg * invS, -1.f), 1.f);
        out.px[2] = fminf(fmaxf(Sb * invS, -1.f), 1.f);
        YIELD(Y, S_FIN);
        if constexpr (WN) {
            float mq = fmaxf(fmaxf(fmaxf(lp2[0], lp2[1]), fmaxf(lp2[2], lp2[3])), lp2[4]);
            mq = pair_max(mq);
            float r[5], se = 0.f;
#pragma unroll
            for (int m = 0; m < 5; ++m) { r[m] = exp2_hw(lp2[m] - mq); se += r[m]; }
            YIELD(Y, S_FIN + 1);
            se = pair_sum(se);
            out.nll = -(mq + log2_hw(se)) * LN2;
            if constexpr (NLL == 2) {
                // ---- gradient row of this pixel: slots 8 k .. 8 k + 7 of mixtures k = 2 m + hh, then their g / b log-scales (112-slot layout) ----
                const float coef = a.nll_scale * (a.nll_row_weight ? a.nll_row_weight[orow] : 1.f);
                const float inv_se = rcp_hw(se);
                float* drow = a.out + ((size_t)orow * plane + (size_t)y * W + xcol) * a.out_pitch;
                const float xr = x[0], xg = x[1];
                float glg[5], glb[5];
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    const float wk = r[m] * inv_se;                                // responsibility of the mixture
                    const float pik = exp2_hw(fmaf(lg[m], L2E, -lse2));
                    const float gw_ = -coef * wk;                                  // d (-logsumexp) / d s_k
                    const float g1 = gw_ * gm[m][1], g2 = gw_ * gm[m][2];
                    float* dk = drow + 8 * (2 * m + hh);
                    float4 va = make_float4(coef * (pik - wk), gw_ * gm[m][0], g1, g2);
                    float4 vb = make_float4(g1 * xr * (1.f - cf[m][0] * cf[m][0]), g2 * xr * (1.f - cf[m][1] * cf[m][1]),
                                            g2 * xg * (1.f - cf[m][2] * cf[m][2]), gw_ * gs[m][0]);
                    // (the row's products stay single registers under any flags: profiles/r05_head_store_hazard.txt)
                    asm volatile("" : "+v"(va.x), "+v"(va.y), "+v"(va.z), "+v"(va.w), "+v"(vb.x), "+v"(vb.y), "+v"(vb.z), "+v"(vb.w));
                    *reinterpret_cast<float4*>(dk) = va;
                    *reinterpret_cast<float4*>(dk + 4) = vb;
                    glg[m] = gw_ * gs[m][1];
                    glb[m] = gw_ * gs[m][2];
                }
                asm volatile("" : "+v"(glg[0]), "+v"(glb[0]), "+v"(glg[1]), "+v"(glb[1]), "+v"(glg[2]), "+v"(glb[2]), "+v"(glg[3]), "+v"(glb[3]),
                             "+v"(glg[4]), "+v"(glb[4]));
                // packing.dlm_log_scale_slot: m = 0, 1 -> slots 80 + 8 h .. + 3, m = 2, 3 -> 84 + 8 h .. + 3, m = 4 -> 96 + 2 h, + 1
                *reinterpret_cast<float4*>(drow + 80 + 8 * hh) = make_float4(glg[0], glb[0], glg[1], glb[1]);
                *reinterpret_cast<float4*>(drow + 84 + 8 * hh) = make_float4(glg[2], glb[2], glg[3], glb[3]);
                *reinterpret_cast<float2*>(drow + 96 + 2 * hh) = make_float2(glg[4], glb[4]);
                // slots 100..111 are empty: lane half 0 zeroes 100..105, half 1 106..111 (no exec masking inside the block)
                float* z = drow + 100 + 6 * hh;
                *reinterpret_cast<float2*>(z) = make_float2(0.f, 0.f);
                *reinterpret_cast<float2*>(z + 2) = make_float2(0.f, 0.f);
                *reinterpret_cast<float2*>(z + 4) = make_float2(0.f, 0.f);
            }
            YIELD(Y, S_FIN + 2);
        }
    };

    if constexpr (!PIPE) {
        // ======== two phases per half-item ========
#ifdef HEAD32_STAGGER                  // (experiment: the second wavefront of every SIMD starts half a phase late)
        if (wave >= NW / 2) __builtin_amdgcn_s_sleep(HEAD32_STAGGER);
#endif
        f32x16 acc[4];
        float txA[3] = {0.f, 0.f, 0.f}, txB[3] = {0.f, 0.f, 0.f};
        int c_orow = -1;
#ifdef HEAD32_TIMING                   // (phase clocks of every wavefront -> a.stats_partial[wave][4]: MFMA passes, epilogues, the rest, items)
        unsigned long long tm_mfma = 0, tm_epi = 0, tm_other = 0, tq_last = __builtin_readcyclecounter();
#endif
        if (n_iter > 0) {
            issue_loads(nxt);
            c_orow = __builtin_amdgcn_readfirstlane(pre_orow);
            stage_math(nxt.strip * 4, nxt.cb * 16, NoYield{}, std::integral_constant<int, 0>{});
        }
        for (int it = 0; it < n_iter; ++it) {
            const Pos cur = nxt;
            const int f = cur.f, y0 = cur.strip * 4, x0 = cur.cb * 16, orow = c_orow;
            const int ek = (fused_nll && orow >= 0) ? 2 : 1;
            const float inv = st_inv;
            write_region();                                        // (the region is free: the previous item's passes have read it)
            const bool have_next = it + 1 < n_iter;
            if (have_next) {
                advance(nxt);
                issue_loads(nxt);                                  // in flight during this item's passes; staged behind them
            }
            if (ek == 2) {
                const float* tp = a.nll_target + (size_t)orow * 3 * plane + (size_t)(y0 + prow) * W + (x0 + pcol);
#pragma unroll
                for (int c = 0; c < 3; ++c) { txA[c] = tp[c * plane]; txB[c] = tp[c * plane + 2 * (size_t)W]; }
            }
            HalfOut oa, ob;
#ifdef HEAD32_TIMING
            const unsigned long long tq0 = __builtin_readcyclecounter();
            tm_other += tq0 - tq_last;
#endif
            auto half = [&](auto nt_tag, auto ek_tag, auto pt_tag, const float (&tx)[3], const int y, HalfOut& o) __attribute__((always_inline)) {
                constexpr int NT = decltype(nt_tag)::value, PT = decltype(pt_tag)::value;
#ifdef HEAD32_TIMING
                const unsigned long long t0 = __builtin_readcyclecounter();
#endif
                {
                    MfmaStream<NT, PT, 1, 0> pass(wl_lane, reg_lane, acc);
                    pass(std::integral_constant<int, 0>{});
                }
#ifdef HEAD32_TIMING
                asm volatile("s_nop 0" : : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]));
                const unsigned long long t1 = __builtin_readcyclecounter();
                tm_mfma += t1 - t0;
#endif
                epilogue(ek_tag, acc, inv, tx, orow, y, x0 + pcol, o, NoYield{});
#ifdef HEAD32_TIMING
                asm volatile("s_nop 0" : : "v"(o.px[0]), "v"(o.px[1]), "v"(o.px[2]), "v"(o.nll));
                tm_epi += __builtin_readcyclecounter() - t1;
#endif
            };
            if (ek == 2) {
                half(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, txA, y0 + prow, oa);
                half(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, txB, y0 + 2 + prow, ob);
            } else {
                half(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, txA, y0 + prow, oa);
                half(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, txB, y0 + 2 + prow, ob);
            }
#ifdef HEAD32_TIMING
            tq_last = __builtin_readcyclecounter();
#endif
            if (have_next) stage_math(nxt.strip * 4, nxt.cb * 16, NoYield{}, std::integral_constant<int, 0>{});
            // image pixels: lane half 0 stores half-item A's pixel, lane half 1 half-item B's (both lanes of a pair hold both)
            float* ip = a.images + (size_t)f * 3 * plane + (size_t)(y0 + 2 * hh + prow) * W + (x0 + pcol);
            const float v0 = hh ? ob.px[0] : oa.px[0], v1 = hh ? ob.px[1] : oa.px[1], v2 = hh ? ob.px[2] : oa.px[2];
            ip[0] = v0; ip[plane] = v1; ip[2 * plane] = v2;
            if (a.images_rows != nullptr && orow >= 0) {
                float* ip2 = a.images_rows + (size_t)orow * 3 * plane + (size_t)(y0 + 2 * hh + prow) * W + (x0 + pcol);
                ip2[0] = v0; ip2[plane] = v1; ip2[2 * plane] = v2;
                if (a.images_rows_dup != 0) {
                    float* ip3 = ip2 + a.images_rows_dup;
                    ip3[0] = v0; ip3[plane] = v1; ip3[2 * plane] = v2;
                }
            }
            if (ek == 2) {
                float v = hh == 0 ? oa.nll + ob.nll : 0.f;
                v = row16_sum_dpp(v);
                const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                v = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);      // lane groups 0 + 1 (2, 3 hold zeros)
                const int it_in_f = (y0 >> 2) * ncb + (x0 >> 4);
                if (lane == 0) a.nll_partial[(size_t)it_in_f * a.nll_rows + orow] = v;
            }
            c_orow = __builtin_amdgcn_readfirstlane(pre_orow);
        }
#ifdef HEAD32_TIMING
        if (lane == 0 && a.stats_partial) {
            float* tp = a.stats_partial + (size_t)gw * 4;
            tp[0] = (float)tm_mfma; tp[1] = (float)tm_epi; tp[2] = (float)tm_other; tp[3] = (float)n_iter;
        }
#endif
    } else {
        // ---- pipeline state ----
        f32x16 accA[4], accB[4];
    #pragma unroll
        for (int c = 0; c < 4; ++c)
    #pragma unroll
            for (int r = 0; r < 16; ++r) { accA[c][r] = 0.f; accB[c][r] = 0.f; }
        // item whose half B waits for its epilogue (wave-uniform), and the values of its half A
        int p_valid = 0, p_f = 0, p_y0 = 0, p_x0 = 0, p_orow = -1, p_ek = 1;
        float p_inv = 1.f;
        float pendA[3] = {0.f, 0.f, 0.f}, nllA = 0.f;
        float txA[3] = {0.f, 0.f, 0.f}, txB[3] = {0.f, 0.f, 0.f}, p_txB[3] = {0.f, 0.f, 0.f};

        // block 1: MFMA pass of half A of the current item (NT tiles; 0 = none) beside the epilogue of half B of the previous item
        auto block1 = [&](auto nt_tag, auto ek_tag, HalfOut& ob) __attribute__((always_inline)) {
            constexpr int NT = decltype(nt_tag)::value, EK = decltype(ek_tag)::value;
            MfmaStream<NT, 0, epi_sites(EK)> Y(wl_lane, reg_lane, accA);
            epilogue(ek_tag, accB, p_inv, p_txB, p_orow, p_y0 + 2 + prow, p_x0 + pcol, ob, Y);
        };
        // block 2: MFMA pass of half B beside the epilogue of half A of the same item and the staging arithmetic of the next item
        auto block2 = [&](auto nt_tag, auto ek_tag, const float inv, const int orow, const int y0, const int x0, const int ny0, const int nx0,
                          HalfOut& oa) __attribute__((always_inline)) {
            constexpr int NT = decltype(nt_tag)::value, EK = decltype(ek_tag)::value;
            MfmaStream<NT, 1, epi_sites(EK) + STAGE_SITES> Y(wl_lane, reg_lane, accB);
            epilogue(ek_tag, accA, inv, txA, orow, y0 + prow, x0 + pcol, oa, Y);
            stage_math(ny0, nx0, Y, std::integral_constant<int, epi_sites(EK)>{});
        };

        // ---- prologue: the first item is loaded and staged outside the pipeline ----
        Pos cur = nxt;
        int c_orow = -1;
        if (n_iter > 0) {
            issue_loads(nxt);
            c_orow = __builtin_amdgcn_readfirstlane(pre_orow);
            stage_math(nxt.strip * 4, nxt.cb * 16, NoYield{}, std::integral_constant<int, 0>{});
        }
        for (int it = 0; it <= n_iter; ++it) {                         // (one extra trip: the last item's half B)
            const bool valid = it < n_iter;
            int f = 0, y0 = 0, x0 = 0, orow = -1, ek = 1;
            float inv = 1.f;
            bool have_next = false;
            if (valid) {
                cur = nxt;
                f = cur.f; y0 = cur.strip * 4; x0 = cur.cb * 16;
                orow = c_orow;
                ek = (fused_nll && orow >= 0) ? 2 : 1;
                inv = st_inv;
                write_region();                                        // (the region is free: the previous item's passes have read it)
                have_next = it + 1 < n_iter;
                if (have_next) {
                    advance(nxt);
                    issue_loads(nxt);                                  // in flight during block 1; staged in block 2
                }
                if (ek == 2) {
                    // the target pixels of this item's two half-items: requested now, read one and two blocks later
                    const float* tp = a.nll_target + (size_t)orow * 3 * plane + (size_t)(y0 + prow) * W + (x0 + pcol);
    #pragma unroll
                    for (int c = 0; c < 3; ++c) { txA[c] = tp[c * plane]; txB[c] = tp[c * plane + 2 * (size_t)W]; }
                }
            }
            // ======== block 1 ========
            HalfOut ob;
            const int nt = valid ? (ek == 2 ? 4 : 3) : 0;
            const int pek = p_valid ? p_ek : 1;                        // (no previous item: a mean epilogue over zero accumulators, discarded)
            if (nt == 4) {
                if (pek == 2) block1(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{}, ob);
                else block1(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{}, ob);
            } else if (nt == 3) {
                if (pek == 2) block1(std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{}, ob);
                else block1(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, ob);
            } else {
                if (pek == 2) block1(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, ob);
                else block1(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, ob);
            }
            // ---- between the blocks: the previous item is complete ----
            if (p_valid) {
                // image pixels: lane half 0 stores half-item A's pixel, lane half 1 half-item B's (both lanes of a pair hold both)
                float* ip = a.images + (size_t)p_f * 3 * plane + (size_t)(p_y0 + 2 * hh + prow) * W + (p_x0 + pcol);
                const float v0 = hh ? ob.px[0] : pendA[0], v1 = hh ? ob.px[1] : pendA[1], v2 = hh ? ob.px[2] : pendA[2];
                ip[0] = v0; ip[plane] = v1; ip[2 * plane] = v2;
                if (a.images_rows != nullptr && p_orow >= 0) {
                    float* ip2 = a.images_rows + (size_t)p_orow * 3 * plane + (size_t)(p_y0 + 2 * hh + prow) * W + (p_x0 + pcol);
                    ip2[0] = v0; ip2[plane] = v1; ip2[2 * plane] = v2;
                    if (a.images_rows_dup != 0) {
                        float* ip3 = ip2 + a.images_rows_dup;
                        ip3[0] = v0; ip3[plane] = v1; ip3[2 * plane] = v2;
                    }
                }
                if (p_ek == 2) {
                    // the item's 64 pixels: lanes 0..31 hold one pixel of each half-item
                    float v = hh == 0 ? nllA + ob.nll : 0.f;
                    v = row16_sum_dpp(v);
                    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                    v = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);      // lane groups 0 + 1 (2, 3 hold zeros)
                    const int it_in_f = (p_y0 >> 2) * ncb + (p_x0 >> 4);
                    if (lane == 0) a.nll_partial[(size_t)it_in_f * a.nll_rows + p_orow] = v;
                }
            }
            if (!valid) break;
            // ======== block 2 ========
            HalfOut oa;
            const int ny0 = have_next ? nxt.strip * 4 : 0, nx0 = have_next ? nxt.cb * 16 : 0;
            if (ek == 2) block2(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{}, inv, orow, y0, x0, ny0, nx0, oa);
            else block2(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, inv, orow, y0, x0, ny0, nx0, oa);
            pendA[0] = oa.px[0]; pendA[1] = oa.px[1]; pendA[2] = oa.px[2];
            nllA = oa.nll;
            p_valid = 1; p_f = f; p_y0 = y0; p_x0 = x0; p_orow = orow; p_ek = ek; p_inv = inv;
            p_txB[0] = txB[0]; p_txB[1] = txB[1]; p_txB[2] = txB[2];
            c_orow = __builtin_amdgcn_readfirstlane(pre_orow);
        }
    }
}

}  // namespace

// Called by gcpx_launch_head_split (conv3x3_head_split.hip) when the caller's split weights are in the 32x32 layout
// (a->split_layout == GCPX_SPLIT_HEAD32, packing.pack_head32_split / head32_index).
int gcpx_launch_head32(const gcpx_conv_args* a, hipStream_t stream) {
    using Cfg = Head32Cfg;
    const int mode = a->head_mode;
    GCPX_CHECK_ARG(mode == GCPX_HEAD_DLM_MEAN || mode == GCPX_HEAD_DLM_NLL || mode == GCPX_HEAD_DLM_NLL_GRAD,
                   "the 32x32 head runs the mean-only and the fused-likelihood modes (stored parameters: GCPX_SPLIT_PLAIN)");
    const int fused = mode != GCPX_HEAD_DLM_MEAN;
    if (fused) {
        GCPX_CHECK_ARG(a->nll_target && a->nll_partial && a->nll_rows > 0 && a->raw_row_map, "GCPX_HEAD_DLM_NLL needs nll_target, nll_partial, nll_rows and raw_row_map");
        GCPX_CHECK_ARG(mode != GCPX_HEAD_DLM_NLL_GRAD || (a->out && a->out_pitch == 112), "GCPX_HEAD_DLM_NLL_GRAD writes the parameter gradient to `out` (112-slot rows)");
    }
    GCPX_CHECK_ARG(a->images, "images is NULL");
    typedef void (*kern_t)(const gcpx_conv_args, const int, const int);
    // forms: "seq8" two phases per half-item, two wavefronts per SIMD (default); "pipe4" software-pipelined, one wavefront per SIMD
    static const int form_env = []() { const char* e = getenv("GCPX_HEAD32_FORM"); return e && !strcmp(e, "pipe4") ? 1 : 0; }();
    const bool grad = mode == GCPX_HEAD_DLM_NLL_GRAD;
    struct Form { kern_t k; int nw; };
    static const Form forms[2][2] = {{{conv3x3_head32_kernel<1, 8, false>, 8}, {conv3x3_head32_kernel<1, 4, true>, 4}},
                                     {{conv3x3_head32_kernel<2, 8, false>, 8}, {conv3x3_head32_kernel<2, 4, true>, 4}}};
    const Form fm = forms[grad ? 1 : 0][form_env];
    const kern_t k = fm.k;
    const int nw = fm.nw;
    const int lds = Cfg::lds_bytes(nw);
    static bool attr_set = false;
    if (!attr_set) {
        for (int g = 0; g < 2; ++g)
            for (int i = 0; i < 2; ++i) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(forms[g][i].k), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::lds_bytes(forms[g][i].nw));
                if (e != hipSuccess) {
                    gcpx_set_error("conv3x3 head32: hipFuncSetAttribute(%d B LDS): %s", Cfg::lds_bytes(forms[g][i].nw), hipGetErrorString(e));
                    return GCPX_ERR_HIP;
                }
            }
        attr_set = true;
    }
    const int nitems = a->F * (a->Hout / 4) * (a->Wout / 16);
    int grid = gcpx_conv_grid() / 2;
    if (grid * nw > nitems) grid = (nitems + nw - 1) / nw;
    hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), lds, stream, *a, nitems, fused);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
