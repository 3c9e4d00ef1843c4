// gemm_tile: the workgroup body of the row GEMM of gemm.hip (see there for layout and call sites), shared with the launches that
// carry GEMM workgroups next to other work (gemm_group_kernel; level_pre_kernel in mlp.hip).
#pragma once
#include "common.h"

#include <type_traits>

namespace {

// KS (K split inside the workgroup): with few rows a wavefront streams its whole weight column alone and the launch is bound by
// the bytes ONE wavefront keeps in flight (8 KiB per ~1.5 us HBM round trip: 13 us for K = 1024 whatever M).  There the four
// wavefronts of a workgroup share one 16 x 16 output tile, take every 4th batch of k-groups and are summed through LDS in a
// fixed order — 4x the bytes in flight, a quarter of the dependent round trips.
template <int PR, int CR, bool LSTM, bool KS = false>
__device__ __forceinline__ void gemm_tile(const gcpx_gemm_args& a, const int bx, const int by, const int bz) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int NT = a.N / 16;
    const int nt0 = KS ? by * CR : (by * 4 + wave) * CR;
    if (!KS && nt0 >= NT) return;
    const int rowblk = bx;
    const int M = a.M, rpb = a.rpb;

    int rr[PR], rb[PR], rj[PR];
    bool rv[PR];
#pragma unroll
    for (int pt = 0; pt < PR; ++pt) {
        rr[pt] = (rowblk * PR + pt) * 16 + j;
        rv[pt] = rr[pt] < M;
        const int r = rv[pt] ? rr[pt] : 0;
        rb[pt] = r / rpb;
        rj[pt] = r % rpb;
    }

    f32x4 acc[CR][PR];
#pragma unroll
    for (int ct = 0; ct < CR; ++ct)
#pragma unroll
        for (int pt = 0; pt < PR; ++pt) acc[ct][pt] = f32x4{0, 0, 0, 0};

    const int zb = bz;
    // What the epilogue reads — bias, and the LSTM's previous cell state — is requested NOW: a few-row launch is one latency chain
    // (arguments -> operands -> dependent MFMAs -> combine -> epilogue, profiles/r06_gemm_rows_pmc.txt) and these two loads sat at its end
    // as a round trip of their own.  (Split K: only the wavefront that runs the epilogue needs them.)
    float4 bias_pre[CR];
    float cprev_pre[CR][PR];
#pragma unroll
    for (int ct = 0; ct < CR; ++ct) {
        bias_pre[ct] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int pt = 0; pt < PR; ++pt) cprev_pre[ct][pt] = 0.f;
    }
    if (!KS || wave == 0) {
#pragma unroll
        for (int ct = 0; ct < CR; ++ct) {
            const int n = (nt0 + ct) * 16 + q * 4;
            if (a.bias) bias_pre[ct] = *reinterpret_cast<const float4*>(a.bias + (size_t)zb * a.z_bias_off + n);
            if constexpr (LSTM) {
#pragma unroll
                for (int pt = 0; pt < PR; ++pt)
                    if (rv[pt]) cprev_pre[ct][pt] = a.c_prev[(size_t)rr[pt] * a.c_prev_stride + (nt0 + ct) * 4 + q];
            }
        }
    }
    const float4* wbase = reinterpret_cast<const float4*>(a.wpk + (size_t)zb * a.z_w_off) + (size_t)nt0 * 64 + lane;
    int kg0 = 0;
    for (int s = 0; s < a.nsrc; ++s) {
        const gcpx_row_src src = a.src[s];
        const float* bp[PR];
        float mask[PR];
#pragma unroll
        for (int pt = 0; pt < PR; ++pt) {
            bool ok = rv[pt];
            size_t off = 0;
            if (src.rowidx) {
                off = (size_t)src.rowidx[rv[pt] ? rr[pt] : 0] * src.sr;
            } else {
                const int jj = rj[pt] + src.shift;
                ok = ok && jj >= 0 && jj < rpb;
                off = (size_t)rb[pt] * src.sb + (size_t)(ok ? jj : 0) * src.sr;
            }
            // rows that do not exist read a valid (clamped) row and are zeroed by the mask: no divergent loads
            mask[pt] = ok ? 1.f : 0.f;
            bp[pt] = src.ptr + (size_t)zb * a.z_src_off + off + q * 4;
        }
        const int nkg = src.width / 16;
        const bool xf = src.scale || src.act;
        // k-groups whose loads are issued together = one register set.  Two sets in ping-pong (explicit, no register copies: a
        // `set = next` after the MFMA block would let the scheduler sink the NEXT batch's wait in front of THIS batch's MFMAs): the
        // loads of one set are in flight during the MFMAs of the other.  A set's MFMAs last UK * PR * CR * 4 * 32 cycles — sized to
        // about one L2 / HBM round trip, because with 1024+ wave tiles there is ONE wavefront per SIMD and nothing else hides it.
        constexpr int UK = KS ? ((PR * CR >= 8) ? 2 : (PR * CR >= 4) ? 4 : 8)
                              : ((PR * CR >= 16) ? 2 : (PR * CR >= 8) ? 4 : (PR * CR >= 2) ? 4 : 8);
        const int kstep = (KS ? 4 : 1) * UK;
        auto load_set = [&](const int kg, float4 (&b)[UK][PR], float4 (&w)[UK][CR]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < UK; ++u) {
                const int k = (kg + u < nkg) ? kg + u : nkg - 1;
#pragma unroll
                for (int pt = 0; pt < PR; ++pt) b[u][pt] = *reinterpret_cast<const float4*>(bp[pt] + k * 16);
                const float4* wp = wbase + (size_t)(kg0 + k) * NT * 64;
#pragma unroll
                for (int ct = 0; ct < CR; ++ct) w[u][ct] = wp[ct * 64];
            }
        };
        // XF: affine + activation of the producer on load; MK: some row of this wavefront is masked (tail rows, conv1d edge taps).
        // Both are wave-uniform and constant over the source: they select one of four loop bodies up front, so that the steady
        // state is loads + MFMAs only — the per-row uniform branches used to cost ~200 cycles per k-group next to 1024 of MFMA.
        auto mfma_set = [&](const int kg, float4 (&b)[UK][PR], float4 (&w)[UK][CR], auto xf_tag, auto mk_tag) __attribute__((always_inline)) {
            constexpr bool XF = decltype(xf_tag)::value, MK = decltype(mk_tag)::value;
#pragma unroll
            for (int u = 0; u < UK; ++u) {
                if (kg + u < nkg) {
                    if constexpr (XF || MK) {
#pragma unroll
                        for (int pt = 0; pt < PR; ++pt) {
                            float4 bb = b[u][pt];
                            if constexpr (XF) bb = affine_act4(bb, src.scale, src.shiftv, ((kg + u) * 16 + q * 4) & (src.cmod - 1), src.act);
                            if constexpr (MK) { bb.x *= mask[pt]; bb.y *= mask[pt]; bb.z *= mask[pt]; bb.w *= mask[pt]; }
                            b[u][pt] = bb;
                        }
                    }
#pragma unroll
                    for (int ct = 0; ct < CR; ++ct) {
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
        };
        float4 b0[UK][PR], w0[UK][CR], b1[UK][PR], w1[UK][CR];
        // loads are issued unconditionally (load_set clamps its k index: a redundant re-read of the last k-group at the end):
        // the number of loads in flight at every wait is then static, and the compiler's s_waitcnt keeps the next set in flight
        auto run = [&](auto xf_tag, auto mk_tag) __attribute__((always_inline)) {
            int kg = KS ? wave * UK : 0;
            if (kg >= nkg) return;
            load_set(kg, b0, w0);
            for (; kg < nkg; kg += 2 * kstep) {
                load_set(kg + kstep, b1, w1);
                mfma_set(kg, b0, w0, xf_tag, mk_tag);
                if (kg + kstep >= nkg) break;
                load_set(kg + 2 * kstep, b0, w0);
                mfma_set(kg + kstep, b1, w1, xf_tag, mk_tag);
            }
        };
        bool anymask = false;
#pragma unroll
        for (int pt = 0; pt < PR; ++pt) anymask = anymask || (mask[pt] == 0.f);
        const bool mk = __any(anymask);
        if (xf) {
            if (mk) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{});
        } else {
            if (mk) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{});
        }
        kg0 += nkg;
    }

    if constexpr (KS) {
        __shared__ float4 part[3][CR * PR][64];
        if (wave > 0) {
#pragma unroll
            for (int ct = 0; ct < CR; ++ct)
#pragma unroll
                for (int pt = 0; pt < PR; ++pt)
                    part[wave - 1][ct * PR + pt][lane] = make_float4(acc[ct][pt][0], acc[ct][pt][1], acc[ct][pt][2], acc[ct][pt][3]);
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w_ = 0; w_ < 3; ++w_)
#pragma unroll
            for (int ct = 0; ct < CR; ++ct)
#pragma unroll
                for (int pt = 0; pt < PR; ++pt) {
                    const float4 v = part[w_][ct * PR + pt][lane];
                    acc[ct][pt][0] += v.x; acc[ct][pt][1] += v.y; acc[ct][pt][2] += v.z; acc[ct][pt][3] += v.w;
                }
    }

    // ---- epilogue ----
#pragma unroll
    for (int ct = 0; ct < CR; ++ct) {
        const int n = (nt0 + ct) * 16 + q * 4;
        const float4 bv = bias_pre[ct];
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pt = 0; pt < PR; ++pt) {
            f32x4 v = acc[ct][pt];
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            if constexpr (LSTM) {
                if (rv[pt]) {
                    const int u = (nt0 + ct) * 4 + q;    // hidden unit of this lane; regs = gates i, f, g, o
                    const float cp = cprev_pre[ct][pt];
                    const float ig = sigmoidf_(v[0]), fg = sigmoidf_(v[1]), gg = tanhf(v[2]), og = sigmoidf_(v[3]);
                    const float c = fg * cp + ig * gg;
                    const float h = og * tanhf(c);
                    const size_t o = (size_t)rb[pt] * a.hb + (size_t)rj[pt] * a.hrow + u;
                    a.h_out[o] = h;
                    a.c_out[o] = c;
                    if (a.h_copy) a.h_copy[(size_t)rr[pt] * (a.N / 4) + u] = h;
                    if (a.gates_out)
                        *reinterpret_cast<float4*>(a.gates_out + ((size_t)rr[pt] * (a.N / 4) + u) * 4) = make_float4(ig, fg, gg, og);
                }
            } else {
                if (a.epi == GCPX_EPI_LRELU) {
                    v[0] = lrelu(v[0], 0.2f); v[1] = lrelu(v[1], 0.2f); v[2] = lrelu(v[2], 0.2f); v[3] = lrelu(v[3], 0.2f);
                }
                if (rv[pt]) {
                    float* op = a.out + (size_t)zb * a.z_out_off + (size_t)rb[pt] * a.ob + (size_t)rj[pt] * a.orow + n;
                    *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                    if (a.lstm_bwd) {                        // the cell backward of the layer this gradient feeds (gcpx_gemm_args.lstm_bwd)
                        const gcpx_lstm_bwd_args& L = *a.lstm_bwd;
                        if (n < L.H) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) lstm_bwd_cell(L, rr[pt], n + e, v[e]);
                        }
                    }
                    if (a.stats_partial) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { s1[r] += v[r]; s2[r] += v[r] * v[r]; }
                    }
                }
            }
        }
        if constexpr (!LSTM) {
            if (a.stats_partial) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t1 = row16_sum(s1[r]);
                    const float t2 = row16_sum(s2[r]);
                    if (j == 0) {
                        a.stats_partial[((size_t)rowblk * 2 + 0) * a.N + n + r] = t1;
                        a.stats_partial[((size_t)rowblk * 2 + 1) * a.N + n + r] = t2;
                    }
                }
            }
        }
    }
}

}  // namespace
