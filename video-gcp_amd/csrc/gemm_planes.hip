// Row GEMM of gemm.hip / gemm_split.hip for problems with many rows, as TWO launches: the activations are split into their two f16 pieces
// ONCE (split_rows_kernel: every source's gather / shift / mask / affine + activation applied, one power-of-two scale per row over the
// whole K extent, pieces written in MFMA fragment order), then a GEMM whose operands BOTH arrive in LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no conversion VALU, no ds_write pass in the loop) and whose pipeline is counted by hand:
//
//   out[r, n] = epi( 2^-(Ex[r] + Ew) * sum_k ( x1 w1 + x1 w2 + x2 w1 )[r, n] + bias[n] )          x 2^Ex[r] = x1 + x2, w 2^Ew = w1 + w2
//
// Replaces the same Linear / LSTMCell launches as gemm.hip:
//   /root/reference/gcp/prediction/models/tree/tree_lstm.py:43-49   (split_linear merge, HiddenStatePredictorModel: embed, LSTMCell x 3)
// at the planner's population sizes (cem_simulator.py:29-31: 512 candidates x 127 nodes) and at the wide levels of a training batch.
//
// Why two launches: gemm_split_kernel converts its activation tile in every one of the N / TN workgroups that share it and carries the
// conversion, the register staging of BOTH operands and a barrier per 32 k inside its loop; its co-residency is set by the registers of
// that staging.  Here the conversion costs one pass over the activations (M K 8 bytes of HBM traffic against the
// GEMM's 6 M N K f16 FLOP), the GEMM's registers hold accumulators and fragments only, and a workgroup can afford 256 x 256 / 128 x 256
// tiles with every stage's bytes in flight one to two stages ahead.
//
// Arithmetic: conv3x3_split.hip (norm-wise bound per scaled unit; the unit here is a whole ROW of the activations — one exponent per row
// from the row's largest magnitude over all of K — and a whole weight tensor).
//
// Layouts (both are what a wavefront's ds_read_b128 fragment read wants, so LDS-DMA copies 1 KiB pieces verbatim):
//   weights     [K / 32][N / 16][2 pieces][64 lanes][8 f16]   packing.pack_gemm_split (lane = 16 (k % 32) / 8 + n % 16)
//   activations [ceil16(M) / 16][K / 32][2 pieces][64 lanes][8 f16]   (lane = 16 (k % 32) / 8 + r % 16), row tiles padded to a multiple of 16
//   exponents   int32 [padded rows]
#include "common.h"

#include <cstdlib>

#ifndef GP_ABLATE
#define GP_ABLATE 0
#endif

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef const f32x4 __attribute__((address_space(1)))* gptr4;
__device__ __forceinline__ float4 gload4(const float* p) {
    const f32x4 t = *(gptr4)p;
    return make_float4(t[0], t[1], t[2], t[3]);
}

__device__ __forceinline__ f32x4 mfma32h(h8 a, h8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__host__ __device__ __forceinline__ int padded_row_tiles(const int M) { return ((M + 255) >> 8) << 4; }

// LSTM gate functions on the hardware's base-2 exponential and reciprocal (1 ulp each): |error| ~ 1e-7 absolute, saturating correctly
// (exp2 -> inf gives rcp = 0).  libm's tanhf / an IEEE division cost ~10x the instructions, and this kernel's epilogue is not hidden
// behind another workgroup's loop (one workgroup per CU).
__device__ __forceinline__ float sigmoid_hw(const float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
__device__ __forceinline__ float tanh_hw(const float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)); }

// 4 x 4 transpose between the registers x[0..3] and the four 16-lane rows of a wavefront (row q = lane >> 4): afterwards x[c] of row q
// holds what x[q] of row c held.  Two swap stages (v_permlane32_swap: rows {2, 3} of the first operand <-> rows {0, 1} of the second;
// v_permlane16_swap: odd rows of the first <-> even rows of the second), four instructions.
__device__ __forceinline__ void transpose4_rows(float (&x)[4]) {
    unsigned u[4] = {__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        auto s = __builtin_amdgcn_permlane32_swap(u[c], u[c + 2], false, false);
        u[c] = s[0]; u[c + 2] = s[1];
    }
#pragma unroll
    for (int c = 0; c < 4; c += 2) {
        auto s = __builtin_amdgcn_permlane16_swap(u[c], u[c + 1], false, false);
        u[c] = s[0]; u[c + 1] = s[1];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = __uint_as_float(u[c]);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// activations -> two f16 planes in fragment order + one exponent per row.  One workgroup of NWV wavefronts per 16 rows; wavefront w takes
// the 32-k steps ks = w (mod NWV); lane (q, j) holds k = 8 q .. 8 q + 7 of row j of a step.  Two passes over the row (the second one hits
// L2): largest magnitude, then scale / split / store.  NWV = 4 when the rows alone fill the chip, 16 for a few hundred row tiles (one
// wavefront then holds two or three steps: the pass is a latency chain, not a stream).
template <bool XF, int NWV>
__global__ void __launch_bounds__(64 * NWV) split_rows_kernel(const gcpx_gemm_args a) {
    __shared__ unsigned wmax[NWV][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int zb = blockIdx.z, rt = blockIdx.x;
    const int M = a.M, rpb = a.rpb;
    const int NKS = a.K >> 5;
    const int r = rt * 16 + j;
    const bool rv = r < M;
    const int rs = rv ? r : 0;
    const int rb = rs / rpb, rj = rs % rpb;

    auto for_each_chunk = [&](auto&& f) __attribute__((always_inline)) {
        int s0 = 0;                                                  // first 32-k step of the source
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i < a.nsrc) {
                const gcpx_row_src& src = a.src[i];
                bool ok = rv;
                size_t off;
                if (src.rowidx) {
                    off = (size_t)src.rowidx[rs] * src.sr;
                } else {
                    const int jj = rj + src.shift;
                    ok = ok && jj >= 0 && jj < rpb;
                    off = (size_t)rb * src.sb + (size_t)(ok ? jj : 0) * src.sr;
                }
                const float* p = src.ptr + (size_t)zb * a.z_src_off + off + q * 8;
                const float mask = ok ? 1.f : 0.f;
                const int n = src.width >> 5;
                for (int l = (wave - s0) & (NWV - 1); l < n; l += NWV) {
                    float4 v0 = gload4(p + 32 * l), v1 = gload4(p + 32 * l + 4);
                    if constexpr (XF) {
                        if (src.scale || src.act) {
                            const int c = (32 * l + q * 8) & (src.cmod - 1);
                            v0 = affine_act4(v0, src.scale, src.shiftv, c, src.act);
                            v1 = affine_act4(v1, src.scale, src.shiftv, (c + 4) & (src.cmod - 1), src.act);
                        }
                    }
                    v0.x *= mask; v0.y *= mask; v0.z *= mask; v0.w *= mask;
                    v1.x *= mask; v1.y *= mask; v1.z *= mask; v1.w *= mask;
                    f(s0 + l, v0, v1);
                }
                s0 += n;
            }
        }
    };

    float amax = 0.f;
    for_each_chunk([&](const int, const float4 v0, const float4 v1) {
        amax = fmaxf(amax, fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))),
                                 fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)))));
    });
    {
        unsigned u = __float_as_uint(amax);                          // non-negative floats order like their bit patterns
        auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        u = max(s16[0], s16[1]);
        auto s32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        u = max(s32[0], s32[1]);
        if (q == 0) wmax[wave][j] = u;
        __syncthreads();
        u = wmax[0][j];
#pragma unroll
        for (int w = 1; w < NWV; ++w) u = max(u, wmax[w][j]);
        amax = __uint_as_float(u);
    }
    const int ew = a.w_split_log2_dev ? a.w_split_log2_dev[zb] : a.w_split_log2;
    const int ex_cap = min(100, 126 - ew);
    int ex = ex_cap;
    if (amax > 0.f) ex = min(ex_cap, max(-100, 14 + 127 - (int)((__float_as_uint(amax) >> 23) & 0xff)));     // amax 2^ex in [2^14, 2^15)
    if (!(amax < 3.0e38f)) ex = -100;                                // inf / nan rows: finite scale, the result carries the inf / nan
    const float sc = __uint_as_float((unsigned)(127 + ex) << 23);
    const int RTP = padded_row_tiles(M);
    char* dst = reinterpret_cast<char*>(a.x_planes) + (((size_t)zb * RTP + rt) * NKS) * 2048 + lane * 16;
    for_each_chunk([&](const int ks, const float4 v0, const float4 v1) {
        const float f[8] = {v0.x * sc, v0.y * sc, v0.z * sc, v0.w * sc, v1.x * sc, v1.y * sc, v1.z * sc, v1.w * sc};
        h8 p1, p2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            p1[e] = (_Float16)f[e];
            p2[e] = (_Float16)fmaf((float)p1[e], -1.f, f[e]);
        }
        *reinterpret_cast<h8*>(dst + (size_t)ks * 2048) = p1;
        *reinterpret_cast<h8*>(dst + (size_t)ks * 2048 + 1024) = p2;
    });
    if (wave == 0 && q == 0) a.x_exp[(size_t)zb * RTP * 16 + r] = ex;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// One LDS-DMA piece: 64 lanes x 16 B from (uniform base + lane offset) to the LDS byte address in M0 (+ 16 lane).  hipcc neither counts
// nor orders what is inside an asm statement: every wait on these loads is the counted s_waitcnt below (MI355X guide, section 5.7).
__device__ __forceinline__ void glds16(const char* base, const unsigned voff, const unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// RT x CTN tiles of 16 rows x 16 columns per workgroup of NW wavefronts arranged WR (rows) x NW / WR (columns); a stage = KSUB steps of
// 32 k; STAGES LDS buffers.  Wavefront (wr, wc) accumulates RT / WR row tiles x CTN / (NW / WR) column tiles.  Per iteration ONE barrier:
//   wait for this wavefront's pieces of stage s (the newer STAGES - 2 stages stay in flight) -> barrier (every wavefront's pieces of stage
//   s have landed; every wavefront has finished reading stage s - 1) -> issue stage s + STAGES - 1 into the buffer stage s - 1 occupied
//   -> fragments + MFMAs of stage s.
template <int RT, int CTN, int NW, int WR, int KSUB, int STAGES, bool LSTM>
__global__ void __launch_bounds__(64 * NW) gemm_planes_kernel(const gcpx_gemm_args a, const int nbx, const int nby) {
    constexpr int WC = NW / WR, RPW = RT / WR, CPW = CTN / WC;
    constexpr int SUB_PIECES = (RT + CTN) * 2;                    // 1 KiB pieces of one 32-k step
    constexpr int SUB_BYTES = SUB_PIECES * 1024, STAGE_BYTES = KSUB * SUB_BYTES;
    // pieces per wavefront and stage: LW of the weights, LX of the activations; a wavefront's pieces of one kind lie inside ONE 32-k
    // step and are consecutive, so their addresses are one uniform base per kind + constants (scalar registers only)
    constexpr int LW = KSUB * CTN * 2 / NW, LX = KSUB * RT * 2 / NW, LPS = LW + LX;
    static_assert((KSUB * CTN * 2) % NW == 0 && (KSUB * RT * 2) % NW == 0 && (CTN * 2) % LW == 0 && (RT * 2) % LX == 0 && LX % 2 == 0 &&
                  RT % WR == 0 && CTN % WC == 0, "tile");
    extern __shared__ float4 smem4[];
    char* smem = reinterpret_cast<char*>(smem4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int j = lane & 15, q = lane >> 4;
    // consecutive blocks land on consecutive XCDs: give every XCD a contiguous range of (row block, column block) pairs, column blocks
    // fastest, so the workgroups that share an activation row block share an L2 (bijective for any block count)
    int id = blockIdx.x;
    {
        const int nblk = nbx * nby, qn = nblk >> 3, rn = nblk & 7, xcd = id & 7, idx = id >> 3;
        id = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
    }
    const int by = id % nby, bx = id / nby;
    const int zb = blockIdx.z;
    const int NT = a.N >> 4, NKS = a.K >> 5, NK = NKS / KSUB;
    const int M = a.M, rpb = a.rpb;
    const int RTP = padded_row_tiles(M);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned voff = lane * 16;
    const int gw = wave * LW, gx = wave * LX;                      // first piece of this wavefront within a stage, per kind
    const int sub_w = gw / (CTN * 2), rem_w = gw % (CTN * 2), sub_x = gx / (RT * 2), rem_x = gx % (RT * 2);
    const char* wv0 = reinterpret_cast<const char*>(a.wpk_split) + (size_t)zb * a.z_w_off * 4 + ((size_t)sub_w * NT + by * CTN) * 2048 + rem_w * 1024;
    const char* xv0 = reinterpret_cast<const char*>(a.x_planes) +
                      ((((size_t)zb * RTP + (size_t)bx * RT + (rem_x >> 1)) * NKS) + sub_x) * 2048;
    const size_t wstride = (size_t)KSUB * NT * 2048, xrow = (size_t)NKS * 2048;
    const unsigned dw0 = lds0 + sub_w * SUB_BYTES + rem_w * 1024, dx0 = lds0 + sub_x * SUB_BYTES + (CTN * 2 + rem_x) * 1024;

    auto issue = [&](const int s, const int b) __attribute__((always_inline)) {
        const char* wp = wv0 + (size_t)s * wstride;
        const char* xp = xv0 + (size_t)s * (KSUB * 2048);
        const unsigned db = b * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < LW; ++i) glds16(wp + i * 1024, voff, dw0 + db + i * 1024);
#pragma unroll
        for (int i = 0; i < LX; ++i) glds16(xp + (size_t)(i >> 1) * xrow + (i & 1) * 1024, voff, dx0 + db + i * 1024);
    };

    f32x4 acc[CPW][RPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int i = 0; i < RPW; ++i) acc[c][i] = f32x4{0, 0, 0, 0};

    // Fragments + MFMAs of one stage.  A unit = one row tile x the wavefront's CPW column tiles (3 CPW MFMAs); the weight fragments of
    // the 32-k step are held for all units, the activation fragments rotate through three register sets and are requested TWO units
    // ahead of their use (an LDS read beside seven other wavefronts' takes a few hundred cycles; a unit is ~200), the order pinned by
    // sched_group_barrier: [2 LDS reads][3 CPW MFMAs] per unit.
    auto compute = [&](const int b) __attribute__((always_inline)) {
        const char* base = smem + b * STAGE_BYTES + lane * 16;
#pragma unroll
        for (int sub = 0; sub < KSUB; ++sub) {
            const char* sb = base + sub * SUB_BYTES;
            const char* xb = sb + CTN * 2048 + (wr * RPW * 2) * 1024;
            h8 w[CPW][2], x[3][2];
#pragma unroll
            for (int i = 0; i < 2 && i < RPW; ++i) {
                x[i][0] = *reinterpret_cast<const h8*>(xb + (i * 2) * 1024);
                x[i][1] = *reinterpret_cast<const h8*>(xb + (i * 2 + 1) * 1024);
            }
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const char* wp = sb + ((wc * CPW + c) * 2) * 1024;
                w[c][0] = *reinterpret_cast<const h8*>(wp);
                w[c][1] = *reinterpret_cast<const h8*>(wp + 1024);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * CPW + (RPW < 2 ? 2 : 4), 0);
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                if (i + 2 < RPW) {
                    x[(i + 2) % 3][0] = *reinterpret_cast<const h8*>(xb + ((i + 2) * 2) * 1024);
                    x[(i + 2) % 3][1] = *reinterpret_cast<const h8*>(xb + ((i + 2) * 2 + 1) * 1024);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                // small terms first: they are added to the accumulator while it is still small
#pragma unroll
                for (int c = 0; c < CPW; ++c) acc[c][i] = mfma32h(w[c][1], x[i % 3][0], acc[c][i]);
#pragma unroll
                for (int c = 0; c < CPW; ++c) acc[c][i] = mfma32h(w[c][0], x[i % 3][1], acc[c][i]);
#pragma unroll
                for (int c = 0; c < CPW; ++c) acc[c][i] = mfma32h(w[c][0], x[i % 3][0], acc[c][i]);
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * CPW, 0);
            }
        }
    };

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < NK) issue(s, s);
    int buf = 0;                                                   // buffer of stage s
    for (int s = 0; s < NK; ++s) {
        // pieces of this wavefront still allowed in flight: those of the min(STAGES - 2, NK - 1 - s) stages behind stage s
        if (s + STAGES - 2 < NK) wait_vm<(STAGES - 2) * LPS>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
#if GP_ABLATE != 1                                                 // (tuning aid: 1 = no loads in the loop, 2 = no fragments / MFMAs)
        if (s + STAGES - 1 < NK) issue(s + STAGES - 1, buf == 0 ? STAGES - 1 : buf - 1);
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifdef GP_TRACE                                                    // (tuning aid, with GP_ABLATE=1: a.h_copy is a long long [8 waves][NK][2] trace of block 0)
        const long long tr0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
#if GP_ABLATE != 2
        compute(buf);
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifdef GP_TRACE
        if (blockIdx.x == 0 && zb == 0 && lane == 0) {
            long long* tr = reinterpret_cast<long long*>(a.h_copy) + ((size_t)wave * NK + s) * 2;
            tr[0] = tr0;
            tr[1] = __builtin_amdgcn_s_memtime();
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
        buf = buf + 1 == STAGES ? 0 : buf + 1;
    }

    // ---- epilogue (scale back: exact; then as gemm.hip) ----
    const int ew = a.w_split_log2_dev ? a.w_split_log2_dev[zb] : a.w_split_log2;
    const int nt0 = by * CTN + wc * CPW;
    if constexpr (LSTM && CPW == 4) {
        // A lane ends with the four gates of ONE hidden unit per column tile (units 4 (nt0 + c) + q, c = 0..3): 4-byte accesses 16 B apart.
        // Transposing registers against the 16-lane rows hands lane (q, j) the four consecutive units of tile nt0 + q instead: the previous
        // cell state comes in and h / c go out as 16-byte accesses, 64 B contiguous per row (needs 16-byte aligned rows; else the scalar form)
        const bool al = ((reinterpret_cast<uintptr_t>(a.c_prev) | reinterpret_cast<uintptr_t>(a.h_out) | reinterpret_cast<uintptr_t>(a.c_out) |
                          reinterpret_cast<uintptr_t>(a.h_copy)) & 15) == 0 && ((a.c_prev_stride | a.hb | a.hrow) & 3) == 0;
        if (al) {
            float4 bv[CPW];
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                bv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias) bv[c] = *reinterpret_cast<const float4*>(a.bias + (size_t)zb * a.z_bias_off + (nt0 + c) * 16 + q * 4);
            }
            const int u4 = (nt0 + q) * 4;
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int r = (bx * RT + wr * RPW + i) * 16 + j;
                const bool rv = r < M;
                const int rs = rv ? r : 0;
                const int rb = rs / rpb, rj = rs % rpb;
                const int ex = a.x_exp[(size_t)zb * RTP * 16 + rs];
                const float inv = __uint_as_float((unsigned)(127 - ex - ew) << 23);
                const float4 cp4 = *reinterpret_cast<const float4*>(a.c_prev + (size_t)rs * a.c_prev_stride + u4);
                float cp[4] = {cp4.x, cp4.y, cp4.z, cp4.w}, hv[4], cv[4];
                transpose4_rows(cp);                               // cp[c] = cell state of unit 4 (nt0 + c) + q
#pragma unroll
                for (int c = 0; c < CPW; ++c) {
                    const f32x4 v = acc[c][i];
                    const float ig = sigmoid_hw(fmaf(v[0], inv, bv[c].x)), fg = sigmoid_hw(fmaf(v[1], inv, bv[c].y));
                    const float gg = tanh_hw(fmaf(v[2], inv, bv[c].z)), og = sigmoid_hw(fmaf(v[3], inv, bv[c].w));
                    cv[c] = fg * cp[c] + ig * gg;
                    hv[c] = og * tanh_hw(cv[c]);
                    if (a.gates_out && rv)                        // (training forward: the activated gates of unit 4 (nt0 + c) + q, kept for the backward)
                        *reinterpret_cast<float4*>(a.gates_out + ((size_t)r * (a.N / 4) + (nt0 + c) * 4 + q) * 4) = make_float4(ig, fg, gg, og);
                }
                transpose4_rows(hv);
                transpose4_rows(cv);
                if (rv) {
                    const size_t o = (size_t)rb * a.hb + (size_t)rj * a.hrow + u4;
                    *reinterpret_cast<float4*>(a.h_out + o) = make_float4(hv[0], hv[1], hv[2], hv[3]);
                    *reinterpret_cast<float4*>(a.c_out + o) = make_float4(cv[0], cv[1], cv[2], cv[3]);
#ifndef GP_TRACE
                    if (a.h_copy) *reinterpret_cast<float4*>(a.h_copy + (size_t)r * (a.N / 4) + u4) = make_float4(hv[0], hv[1], hv[2], hv[3]);
#endif
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = (bx * RT + wr * RPW + i) * 16 + j;
        const bool rv = r < M;
        const int rs = rv ? r : 0;
        const int rb = rs / rpb, rj = rs % rpb;
        const int ex = a.x_exp[(size_t)zb * RTP * 16 + rs];
        const float inv = __uint_as_float((unsigned)(127 - ex - ew) << 23);
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int nt = nt0 + c;
            const int n = nt * 16 + q * 4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + (size_t)zb * a.z_bias_off + n);
            f32x4 v = acc[c][i];
            v[0] = fmaf(v[0], inv, bv.x); v[1] = fmaf(v[1], inv, bv.y); v[2] = fmaf(v[2], inv, bv.z); v[3] = fmaf(v[3], inv, bv.w);
            if (!rv) continue;
            if constexpr (LSTM) {
                const int u = nt * 4 + q;                          // hidden unit of this lane; regs = gates i, f, g, o
                const float cp = a.c_prev[(size_t)r * a.c_prev_stride + u];
                const float ig = sigmoid_hw(v[0]), fg = sigmoid_hw(v[1]), gg = tanh_hw(v[2]), og = sigmoid_hw(v[3]);
                const float cn = fg * cp + ig * gg;
                const float h = og * tanh_hw(cn);
                const size_t o = (size_t)rb * a.hb + (size_t)rj * a.hrow + u;
                a.h_out[o] = h;
                a.c_out[o] = cn;
                if (a.gates_out) *reinterpret_cast<float4*>(a.gates_out + ((size_t)r * (a.N / 4) + u) * 4) = make_float4(ig, fg, gg, og);
#ifndef GP_TRACE
                if (a.h_copy) a.h_copy[(size_t)r * (a.N / 4) + u] = h;
#endif
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

template <int RT, int CTN, int NW, int WR, int KSUB, int STAGES>
int launch_planes(const gcpx_gemm_args* a, hipStream_t stream) {
    constexpr int LDS = STAGES * KSUB * (RT + CTN) * 2048;
    static bool attr_set = false;
    if (!attr_set) {
        for (const void* k : {reinterpret_cast<const void*>(gemm_planes_kernel<RT, CTN, NW, WR, KSUB, STAGES, false>),
                              reinterpret_cast<const void*>(gemm_planes_kernel<RT, CTN, NW, WR, KSUB, STAGES, true>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) {
                gcpx_set_error("gemm planes: hipFuncSetAttribute(%d B LDS): %s", LDS, hipGetErrorString(e));
                return GCPX_ERR_HIP;
            }
        }
        attr_set = true;
    }
    const int nb = a->nbatch > 1 ? a->nbatch : 1;
    const int nbx = (a->M + RT * 16 - 1) / (RT * 16), nby = a->N / (CTN * 16);
    const dim3 grid(nbx * nby, 1, nb);
    if (a->epi == GCPX_EPI_LSTM)
        hipLaunchKernelGGL((gemm_planes_kernel<RT, CTN, NW, WR, KSUB, STAGES, true>), grid, dim3(64 * NW), LDS, stream, *a, nbx, nby);
    else
        hipLaunchKernelGGL((gemm_planes_kernel<RT, CTN, NW, WR, KSUB, STAGES, false>), grid, dim3(64 * NW), LDS, stream, *a, nbx, nby);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

}  // namespace

// bytes of the activation planes / number of row exponents a problem needs (row tiles padded to 256 rows per batch entry: a workgroup
// of the largest tile reads whole row blocks)
extern "C" int gcpx_gemm_planes_workspace(int32_t M, int32_t K, int32_t nbatch, int64_t* planes_bytes, int64_t* n_exp) {
    const int64_t nb = nbatch > 1 ? nbatch : 1;
    const int64_t rtp = padded_row_tiles(M);
    if (planes_bytes) *planes_bytes = nb * rtp * (K / 32) * 2048;
    if (n_exp) *n_exp = nb * rtp * 16;
    return GCPX_OK;
}

// gemm.hip asks: planes workspace given, split weights given, enough rows that the conversion pass pays, shapes the tiles cover
bool gcpx_gemm_planes_applies(const gcpx_gemm_args* a) {
    static const int min_rows = [] { const char* e = getenv("GCPX_GEMM_PLANES_MIN_ROWS"); return e ? atoi(e) : 512; }();
    if (!a->x_planes || !a->x_exp || !a->wpk_split || a->M < min_rows || a->N % 128 || a->K % 64 || a->stats_partial || a->lstm_bwd) return false;
    for (int s = 0; s < a->nsrc; ++s)
        if (a->src[s].width % 32) return false;
    int64_t need = 0;
    gcpx_gemm_planes_workspace(a->M, a->K, a->nbatch, &need, nullptr);
    return a->x_planes_bytes >= need;
}

int gcpx_launch_gemm_planes(const gcpx_gemm_args* a, hipStream_t stream) {
    const int nb = a->nbatch > 1 ? a->nbatch : 1;
    bool xf = false;
    for (int s = 0; s < a->nsrc; ++s) xf = xf || a->src[s].scale || a->src[s].act;
    const dim3 sgrid((a->M + 15) / 16, 1, nb);
    if ((long)sgrid.x * nb >= 1024) {
        if (xf) hipLaunchKernelGGL((split_rows_kernel<true, 4>), sgrid, dim3(256), 0, stream, *a);
        else hipLaunchKernelGGL((split_rows_kernel<false, 4>), sgrid, dim3(256), 0, stream, *a);
    } else {
        if (xf) hipLaunchKernelGGL((split_rows_kernel<true, 16>), sgrid, dim3(1024), 0, stream, *a);
        else hipLaunchKernelGGL((split_rows_kernel<false, 16>), sgrid, dim3(1024), 0, stream, *a);
    }
    GCPX_CHECK_LAUNCH();
    static const int force = [] { const char* e = getenv("GCPX_GEMM_PLANES_CFG"); return e ? atoi(e) : 0; }();     // tuning aid
    const long tiles_a = (long)((a->M + 255) / 256) * (a->N / 256) * nb, tiles_b = (long)((a->M + 127) / 128) * (a->N / 256) * nb;
    const long tiles_d = (long)((a->M + 127) / 128) * (a->N / 128) * nb;
    const bool n256 = a->N % 256 == 0;
    if (force == 1 && n256) return launch_planes<16, 16, 8, 2, 1, 2>(a, stream);
    if (force == 2 && n256) return launch_planes<8, 16, 8, 2, 1, 3>(a, stream);
    if (force == 3) return launch_planes<8, 8, 8, 4, 2, 2>(a, stream);
    if (force == 4) return launch_planes<4, 8, 4, 2, 2, 3>(a, stream);
    if (force == 5) return launch_planes<16, 8, 8, 4, 1, 3>(a, stream);                    // 256 x 128, 3 x 48 KB
    if (force == 6) return launch_planes<8, 8, 8, 4, 1, 4>(a, stream);                     // 128 x 128, 32-k stages, 4 x 32 KB
    if (force == 7) return launch_planes<4, 4, 4, 4, 2, 4>(a, stream);                     // 64 x 64, 64-k stages, 4 x 32 KB
    if (n256 && tiles_a >= 256) return launch_planes<16, 16, 8, 2, 1, 2>(a, stream);       // 256 x 256, 2 x 64 KB
    if (n256 && tiles_b >= 256) return launch_planes<8, 16, 8, 2, 1, 3>(a, stream);        // 128 x 256, 3 x 48 KB
    if (tiles_d >= 256) return launch_planes<8, 8, 8, 4, 2, 2>(a, stream);                 // 128 x 128, 64-k stages, 2 x 64 KB
    const long tiles_c = (long)((a->M + 63) / 64) * (a->N / 128) * nb;
    if (tiles_c < 192) return launch_planes<4, 4, 4, 4, 2, 4>(a, stream);                  // 64 x 64, 64-k stages, 4 x 32 KB (a few hundred rows)
    return launch_planes<4, 8, 4, 2, 2, 3>(a, stream);                                     // 64 x 128, 64-k stages, 3 x 48 KB
}
