// Row GEMM of gemm.hip on the f16 matrix pipes of gfx950 with f32-equivalent arithmetic ("split-f16"), for problems with enough
// rows to be bound by the f32 MFMA rate (tree levels with >= 128 nodes, the planner's 65 k-node trees):
//
//   out[r, n] = epi( sum_s sum_k X_s[map_s(r), k] * W[n, koff_s + k] + bias[n] )
//
// Replaces the same Linear / LSTMCell launches as gemm.hip:
//   /root/reference/gcp/prediction/models/tree/tree_lstm.py:43-49   (split_linear merge, HiddenStatePredictorModel: embed, LSTMCell x 3)
//
// Arithmetic: as in conv3x3_split.hip — both operands as two f16 pieces, three v_mfma_f32_16x16x32_f16 per f32 product, every
// partial product exact in the f32 accumulator.  The weights are split once (packing.pack_gemm_split, one power-of-two scale per
// tensor).  The activations are split while they are staged, with a power-of-two scale PER ROW that follows the row along K: the
// four threads that stage a row keep the row's running exponent (largest magnitude seen so far lands in [2^14, 2^15)), publish it
// with every stage, and the wavefronts that accumulate the row rescale their sums (exactly) when it drops.  A row's error is then
// a few f32 roundings of its largest term whatever the magnitude of the data — gradients of 1e-6 as well as states of order one.
//
// Tiling: one 256-thread workgroup computes 64 rows x TN columns (TN = 128 or 64) — or, for problems with thousands of rows, one
// 512-thread workgroup 128 rows x 128 columns (template parameter NW) —; a stage is 32 NSUB k.  Weights travel global -> registers
// -> LDS in fragment order (the packed layout IS the LDS layout), activations global (gathered / shifted / masked rows, producer's
// affine + activation applied) -> registers -> two f16 planes in fragment order.  Two LDS stages, the next stage's global loads in
// flight during this stage's 48 MFMAs per wavefront; wavefront (wr, wc) owns row tiles 2 wr, 2 wr + 1 x column tiles of half wc.
// A lane ends with 4 consecutive columns of one row, exactly as in gemm.hip: the LSTM-cell epilogue is unchanged.
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* gptr4;
__device__ __forceinline__ float4 gload4(const float* p) {      // explicitly global: a flat load would tie up both memory counters
    const f32x4 t = *(gptr4)p;
    return make_float4(t[0], t[1], t[2], t[3]);
}

__device__ __forceinline__ f32x4 mfma32h(h8 a, h8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// NW = wavefronts of the workgroup: 4 (64 rows) or 8 (128 rows, 512 threads: the same roles per wavefront — a wavefront stages the 16
// rows of "its" row tile and accumulates two row tiles x half the column tiles —, 1.55x the FLOPs per operand byte pulled)
template <int TN, int NSUB, int NW = 4>
struct GsCfg {
    static constexpr int TM = 16 * NW, THREADS = 64 * NW, KS = 32 * NSUB;   // a stage = NSUB sub-steps of 32 k
    static constexpr int NCT = TN / 16;                         // column tiles of the workgroup
    static constexpr int W_SUB = NCT * 2048;                    // one 32-k sub-step: [NCT][2 pieces][64 lanes] x 16 B
    static constexpr int W_BYTES = NSUB * W_SUB;
    static constexpr int X_SUB = NW * 2048;                     // [NW row tiles][2 pieces][64 lanes] x 16 B
    static constexpr int X_BYTES = NSUB * X_SUB;
    static constexpr int STAGE = W_BYTES + X_BYTES + 4 * TM;    // + the TM row exponents
    static constexpr int LDS_BYTES = 2 * STAGE;
    static constexpr int WLD = W_BYTES / (THREADS * 16);        // 16-byte weight loads per thread and stage
    static constexpr int WSTEP = THREADS * 16;                  // bytes the workgroup moves per weight load
};

// XF: some source carries an affine / activation on load (its per-channel loads are conditional: the memory counter is then drained
// at every conversion; the plain variant keeps the newer register set in flight)
template <int TN, int NSUB, bool LSTM, bool XF, int NW = 4>
__global__ void __launch_bounds__(64 * NW) gemm_split_kernel(const gcpx_gemm_args a) {
    using Cfg = GsCfg<TN, NSUB, NW>;
    constexpr int KS = Cfg::KS;
    constexpr int NCT = Cfg::NCT, WLD = Cfg::WLD, CPW = NCT / 2;          // column tiles per wavefront
    extern __shared__ float4 smem4[];
    char* smem = reinterpret_cast<char*>(smem4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int zb = blockIdx.z;
    const int NT = a.N / 16;
    const int ct0 = blockIdx.y * NCT;
    const int M = a.M, rpb = a.rpb;
    const int NK = a.K / KS;

    // ---- staging role: thread t stages k = 8 kq .. + 7 and 32 + 8 kq .. + 7 of row 16 (t >> 6) + (t & 15), kq = (t & 63) >> 4, of every
    //      stage: its two 16-byte pieces are lane t & 63 of row tile t >> 6 (a wavefront writes 1 KiB contiguous: no bank conflicts) ----
    const int srow = (wave << 4) + j, kq = q;
    const int sr_ = blockIdx.x * Cfg::TM + srow;
    const bool srv = sr_ < M;
    const int srs = srv ? sr_ : 0;
    const int srb = srs / rpb, srj = srs % rpb;
    const int ew = a.w_split_log2_dev ? a.w_split_log2_dev[zb] : a.w_split_log2;
    const int ex_cap = min(100, 126 - ew);
    int ex_run = ex_cap;                                          // the row's running exponent (identical in its four threads)
    // walking state over the concatenated sources (the loads run two stages ahead of the conversion: every register set carries
    // the source and the k offset it was loaded from)
    int cur_s = -1, krem = 0, kloc = 0;
    const float* xp = nullptr;
    float xmask = 0.f;
    // every source's row pointer and mask up front (the row gather's index load must not sit inside the pipelined loop: a load
    // whose result is needed at once drains the memory counter, prefetched register sets included)
    const float* sp[6];
    float sm[6];
    int sw[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        sp[i] = a.src[0].ptr; sm[i] = 0.f; sw[i] = 0;
        if (i < a.nsrc) {
            sw[i] = a.src[i].width / KS;
            const gcpx_row_src& src = a.src[i];
            bool ok = srv;
            size_t off;
            if (src.rowidx) {
                off = (size_t)src.rowidx[srs] * src.sr;
            } else {
                const int jj = srj + src.shift;
                ok = ok && jj >= 0 && jj < rpb;
                off = (size_t)srb * src.sb + (size_t)(ok ? jj : 0) * src.sr;
            }
            sp[i] = src.ptr + (size_t)zb * a.z_src_off + off + kq * 8;
            sm[i] = ok ? 1.f : 0.f;
        }
    }
    auto next_source = [&]() {
        ++cur_s;
        xp = sp[0]; xmask = sm[0]; krem = sw[0];
#pragma unroll
        for (int i = 1; i < 6; ++i)
            if (cur_s == i) { xp = sp[i]; xmask = sm[i]; krem = sw[i]; }
        kloc = 0;
    };
    const char* wsrc = reinterpret_cast<const char*>(a.wpk_split) + (size_t)zb * a.z_w_off * 4 + (size_t)ct0 * 2048 + tid * 16;

    struct RegSet {
        float4 x[2 * NSUB], w[WLD];
        float mask;
        int src, k0;
    };
    RegSet rs[2];
    // Always issued, also past the last stage (the last stage is then loaded again and never used): the number of loads in flight
    // at every wait is static, so the compiler's s_waitcnt before a conversion leaves the NEWER register set in flight.
    const float* last_x = sp[0];
    int last_s = 0;
    auto issue = [&](const int s, RegSet& r) __attribute__((always_inline)) {      // global loads of stage s into registers
        if (s < NK) {
            if (krem == 0) next_source();
            last_x = xp + kloc;
            last_s = s;
            r.mask = xmask; r.src = cur_s; r.k0 = kloc;
            kloc += KS;
            --krem;
        }
#pragma unroll
        for (int h = 0; h < NSUB; ++h) {
            r.x[2 * h] = gload4(last_x + 32 * h);
            r.x[2 * h + 1] = gload4(last_x + 32 * h + 4);
        }
#pragma unroll
        for (int i = 0; i < WLD; ++i) {
            const int sub = i / (WLD / NSUB), c = i % (WLD / NSUB);
            r.w[i] = *reinterpret_cast<const float4*>(wsrc + ((size_t)(NSUB * last_s + sub) * NT) * 2048 + c * Cfg::WSTEP);
        }
    };
    auto commit = [&](const int st, RegSet& r) __attribute__((always_inline)) {    // registers -> LDS stage st (affine, scale, split)
        char* base = smem + st * Cfg::STAGE;
#pragma unroll
        for (int i = 0; i < WLD; ++i) {
            const int sub = i / (WLD / NSUB), c = i % (WLD / NSUB);
            *reinterpret_cast<float4*>(base + sub * Cfg::W_SUB + c * Cfg::WSTEP + tid * 16) = r.w[i];
        }
        if constexpr (XF) {
#pragma unroll
            for (int si = 0; si < 6; ++si) {
                if (r.src == si && (a.src[si].scale || a.src[si].act)) {
#pragma unroll
                    for (int i = 0; i < 2 * NSUB; ++i)
                        r.x[i] = affine_act4(r.x[i], a.src[si].scale, a.src[si].shiftv,
                                             (r.k0 + 32 * (i >> 1) + 4 * (i & 1) + kq * 8) & (a.src[si].cmod - 1), a.src[si].act);
                }
            }
        }
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 2 * NSUB; ++i) {
            float4 v = r.x[i];
            v.x *= r.mask; v.y *= r.mask; v.z *= r.mask; v.w *= r.mask;
            r.x[i] = v;
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        // largest magnitude of the row's 64 values: over its four threads (lanes j, j + 16, j + 32, j + 48 of this wavefront)
        {
            unsigned u = __float_as_uint(amax);
            auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            u = max(s16[0], s16[1]);                               // non-negative floats order like their bit patterns
            auto s32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            amax = __uint_as_float(max(s32[0], s32[1]));
        }
        if (amax > 0.f) {
            const int ec = 14 + 127 - (int)((__float_as_uint(amax) >> 23) & 0xff);     // amax 2^ec in [2^14, 2^15)
            ex_run = min(ex_run, max(-100, ec));
        }
        const float sc = __uint_as_float((unsigned)(127 + ex_run) << 23);
        char* xb = base + Cfg::W_BYTES + (wave * 2) * 1024 + lane * 16;
#pragma unroll
        for (int h = 0; h < NSUB; ++h) {
            const float4 v0 = r.x[2 * h], v1 = r.x[2 * h + 1];
            const float f[8] = {v0.x * sc, v0.y * sc, v0.z * sc, v0.w * sc, v1.x * sc, v1.y * sc, v1.z * sc, v1.w * sc};
            h8 p1, p2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                p1[e] = (_Float16)f[e];
                p2[e] = (_Float16)fmaf((float)p1[e], -1.f, f[e]);
            }
            *reinterpret_cast<h8*>(xb + h * Cfg::X_SUB) = p1;
            *reinterpret_cast<h8*>(xb + h * Cfg::X_SUB + 1024) = p2;
        }
        if (kq == 0) *reinterpret_cast<int*>(base + Cfg::W_BYTES + Cfg::X_BYTES + srow * 4) = ex_run;
    };

    // ---- accumulate role ----
    f32x4 acc[CPW][2];
#pragma unroll
    for (int c = 0; c < CPW; ++c) { acc[c][0] = f32x4{0, 0, 0, 0}; acc[c][1] = f32x4{0, 0, 0, 0}; }
    int excur[2] = {ex_cap, ex_cap};
    auto compute = [&](const int st) __attribute__((always_inline)) {
        const char* base = smem + st * Cfg::STAGE;
        // the rows' exponents of this stage; sums accumulated under a larger exponent are scaled down (exact)
        int en[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) en[i] = *reinterpret_cast<const int*>(base + Cfg::W_BYTES + Cfg::X_BYTES + ((2 * wr + i) * 16 + j) * 4);
        if (__any(en[0] != excur[0] || en[1] != excur[1])) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int d = en[i] - excur[i];                     // <= 0
                const float r = d < -126 ? 0.f : __uint_as_float((unsigned)(127 + d) << 23);
#pragma unroll
                for (int c = 0; c < CPW; ++c) acc[c][i] *= r;
                excur[i] = en[i];
            }
        }
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            h8 b[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* bp = base + Cfg::W_BYTES + sub * Cfg::X_SUB + ((2 * wr + i) * 2) * 1024 + lane * 16;
                b[i][0] = *reinterpret_cast<const h8*>(bp);
                b[i][1] = *reinterpret_cast<const h8*>(bp + 1024);
            }
            // column tiles four at a time (the 256-column form would hold 64 registers of weight fragments otherwise)
            constexpr int CH = CPW <= 4 ? CPW : 2;
#pragma unroll
            for (int c0 = 0; c0 < CPW; c0 += CH) {
                h8 w[CH][2];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const char* wp = base + sub * Cfg::W_SUB + ((wc * CPW + c0 + c) * 2) * 1024 + lane * 16;
                    w[c][0] = *reinterpret_cast<const h8*>(wp);
                    w[c][1] = *reinterpret_cast<const h8*>(wp + 1024);
                }
                // small terms first: they are added to the accumulator while it is still small
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[c0 + c][i] = mfma32h(w[c][1], b[i][0], acc[c0 + c][i]);
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[c0 + c][i] = mfma32h(w[c][0], b[i][1], acc[c0 + c][i]);
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[c0 + c][i] = mfma32h(w[c][0], b[i][0], acc[c0 + c][i]);
            }
        }
    };

    // stage s is computed from LDS buffer s & 1 while the registers hold stages s + 1 and s + 2
    issue(0, rs[0]);
    issue(1, rs[1]);
    commit(0, rs[0]);
    issue(2, rs[0]);
    __syncthreads();
    for (int s = 0; s < NK; s += 2) {
        compute(0);
        __builtin_amdgcn_sched_barrier(0);                         // (the conversion's waits stay behind the MFMAs)
        commit(1, rs[1]);                                          // (stage s + 1; past the end: a copy of the last stage, never read)
        issue(s + 3, rs[1]);
        __syncthreads();
        if (s + 1 >= NK) break;
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
        commit(0, rs[0]);
        issue(s + 4, rs[0]);
        __syncthreads();
    }

    // ---- epilogue (scale back: exact; then as gemm.hip) ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = blockIdx.x * Cfg::TM + (2 * wr + i) * 16 + j;
        const bool rv = r < M;
        const int rs = rv ? r : 0;
        const int rb = rs / rpb, rj = rs % rpb;
        const float inv = __uint_as_float((unsigned)(127 - excur[i] - ew) << 23);
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int nt = ct0 + wc * CPW + c;
            const int n = nt * 16 + q * 4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + (size_t)zb * a.z_bias_off + n);
            f32x4 v = acc[c][i];
            v[0] = fmaf(v[0], inv, bv.x); v[1] = fmaf(v[1], inv, bv.y); v[2] = fmaf(v[2], inv, bv.z); v[3] = fmaf(v[3], inv, bv.w);
            if (!rv) continue;
            if constexpr (LSTM) {
                const int u = nt * 4 + q;                          // hidden unit of this lane; regs = gates i, f, g, o
                const float cp = a.c_prev[(size_t)r * a.c_prev_stride + u];
                const float ig = sigmoidf_(v[0]), fg = sigmoidf_(v[1]), gg = tanhf(v[2]), og = sigmoidf_(v[3]);
                const float cn = fg * cp + ig * gg;
                const float h = og * tanhf(cn);
                const size_t o = (size_t)rb * a.hb + (size_t)rj * a.hrow + u;
                a.h_out[o] = h;
                a.c_out[o] = cn;
                if (a.h_copy) a.h_copy[(size_t)r * (a.N / 4) + u] = h;
                if (a.gates_out)                                   // (training forward: the activated gates, kept for the backward pass)
                    *reinterpret_cast<float4*>(a.gates_out + ((size_t)r * (a.N / 4) + u) * 4) = make_float4(ig, fg, gg, og);
            } else {
                if (a.epi == GCPX_EPI_LRELU) {
                    v[0] = lrelu(v[0], 0.2f); v[1] = lrelu(v[1], 0.2f); v[2] = lrelu(v[2], 0.2f); v[3] = lrelu(v[3], 0.2f);
                }
                float* op = a.out + (size_t)zb * a.z_out_off + (size_t)rb * a.ob + (size_t)rj * a.orow + n;
                *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <int TN, int NSUB, int NW = 4>
int launch_split(const gcpx_gemm_args* a, hipStream_t stream) {
    using Cfg = GsCfg<TN, NSUB, NW>;
    static bool attr_set = false;
    if (!attr_set) {
        for (const void* k : {reinterpret_cast<const void*>(gemm_split_kernel<TN, NSUB, false, false, NW>), reinterpret_cast<const void*>(gemm_split_kernel<TN, NSUB, true, false, NW>),
                              reinterpret_cast<const void*>(gemm_split_kernel<TN, NSUB, false, true, NW>), reinterpret_cast<const void*>(gemm_split_kernel<TN, NSUB, true, true, NW>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
            if (e != hipSuccess) {
                gcpx_set_error("gemm split: hipFuncSetAttribute(%d B LDS): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
                return GCPX_ERR_HIP;
            }
        }
        attr_set = true;
    }
    const int nb = a->nbatch > 1 ? a->nbatch : 1;
    const dim3 grid((a->M + Cfg::TM - 1) / Cfg::TM, a->N / TN, nb);
    bool xf = false;
    for (int s = 0; s < a->nsrc; ++s) xf = xf || a->src[s].scale || a->src[s].act;
    const bool lstm = a->epi == GCPX_EPI_LSTM;
    auto kern = lstm ? (xf ? gemm_split_kernel<TN, NSUB, true, true, NW> : gemm_split_kernel<TN, NSUB, true, false, NW>)
                     : (xf ? gemm_split_kernel<TN, NSUB, false, true, NW> : gemm_split_kernel<TN, NSUB, false, false, NW>);
    hipLaunchKernelGGL(kern, grid, dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, *a);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

}  // namespace

// gemm.hip asks: does this problem have a split-f16 form, and is it worth it?  Every source a multiple of 64 wide, columns a multiple
// of 64, no statistics / saved gates (those stay on the exact kernel), and at least min_rows rows (below that the launch is bound
// by the rate at which a CU pulls its operand bytes, ~10 B / cycle, not by the f32 MFMA rate: measured 24 us at 128 rows against 21 us).
bool gcpx_gemm_split_applies(const gcpx_gemm_args* a) {
    static const int min_rows = [] { const char* e = getenv("GCPX_GEMM_SPLIT_MIN_ROWS"); return e ? atoi(e) : 512; }();
    if (!a->wpk_split || a->M < min_rows || a->N % 64 || a->stats_partial || a->lstm_bwd) return false;
    for (int s = 0; s < a->nsrc; ++s)
        if (a->src[s].width % 64) return false;
    return true;
}

int gcpx_launch_gemm_split(const gcpx_gemm_args* a, hipStream_t stream) {
    const long nb = a->nbatch > 1 ? a->nbatch : 1;
    const long rbk = (a->M + 63) / 64;
    // 128-column tiles read fewer operand bytes per MFMA; 64-column tiles when those would leave most of the chip without a workgroup
    static const int force = [] { const char* e = getenv("GCPX_GEMM_SPLIT_CFG"); return e ? atoi(e) : 0; }();     // tuning aid: TN * 10 + NSUB
    if (force == 1282) return launch_split<128, 2>(a, stream);
    if (force == 642) return launch_split<64, 2>(a, stream);
    if (force == 641) return launch_split<64, 1>(a, stream);
    if (force == 1281) return launch_split<128, 1>(a, stream);
    if (force == 2561 && a->N % 256 == 0) return launch_split<256, 1>(a, stream);
    if (force == 1288 && a->N % 128 == 0) return launch_split<128, 1, 8>(a, stream);      // 128 x 128 tiles, 512 threads
    if (force == 2568 && a->N % 256 == 0) return launch_split<256, 1, 8>(a, stream);      // 128 x 256 tiles, 512 threads
    // Stages of 32 k: 24 / 16 KB of LDS per stage and 208 / 144 registers per thread (128- / 64-column tiles), so two / three workgroups
    // share a CU — the registers, not the LDS, set that number — and one's barrier and conversion hide behind the others' loads (64-k
    // stages: 264 / 176 registers, one or two workgroups per CU: 1024 x 2048 x 1024 38 us, 32768 rows 1082 us; 32-k stages 35 / 754).
    // 128-column tiles (fewer operand bytes per MFMA) once they still give every CU three workgroups, 64-column tiles below.
    // Many rows (the planner's 65 k-node trees): 128 x 128 tiles in 512-thread workgroups once they still give every CU two of them —
    // 1.55x the FLOPs per operand byte pulled, the same number of wavefronts per CU: 32768 / 8192 / 4096 x 2048 x 1024 802 / 199 / 94 us
    // against 872 / 214 / 107 (at 1024 rows 44 against 35: half the chip without a workgroup).
    // ... and 128 x 256 tiles (2.1x) once those give every CU a workgroup: 640 / 155 / 77 us, the batched merge 813 against 1 013 us
    // (one workgroup of eight wavefronts per CU in both forms: 256 / 184 registers per thread).
    if (a->N % 256 == 0 && ((a->M + 127) / 128) * (long)(a->N / 256) * nb >= 256) return launch_split<256, 1, 8>(a, stream);
    if (a->N % 128 == 0 && ((a->M + 127) / 128) * (long)(a->N / 128) * nb >= 512) return launch_split<128, 1, 8>(a, stream);
    if (a->N % 128 == 0 && rbk * (a->N / 128) * nb >= 768) return launch_split<128, 1>(a, stream);
    return launch_split<64, 1>(a, stream);
}
