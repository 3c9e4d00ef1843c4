// Row GEMM with gathered / concatenated / shifted inputs and fused epilogues, on f32 MFMA (gfx950).
//
//   out[r, n] = epi( sum_s sum_k X_s[map_s(r), k] * W[n, koff_s + k] + bias[n] )
//
// Replaces the Linear / LSTMCell / Conv1d / 1x1->4x4 ConvTranspose / 4x4-valid Conv launches of
//   /root/reference/gcp/prediction/models/tree/tree_lstm.py:43-49   (split_linear merge, HiddenStatePredictorModel)
//   /root/reference/gcp/prediction/models/base_gcp.py:199           (ConvSeqEncodingModule, conv over time)
//   encoder head / decoder input block (blox, absent; spec in DESIGN.md).
//
// Layout: output columns n sit on the MFMA i side (A operand = weights, pre-packed in fragment order so a
// wavefront's load of 16 columns x 16 k is one coalesced 1 KiB read), rows sit on the j side (B operand: lane
// (j, kk) reads 16 B = 4 consecutive k of its row, with the producer's BatchNorm affine + LeakyReLU applied on
// load).  A lane ends with 4 consecutive columns of one row: with gate-interleaved LSTM weights (n = 4u + gate)
// the whole cell update for (row, unit u) is lane-local.  No LDS, no barriers: wavefronts are independent.
#include "common.h"

#include <cstdio>
#include <cstdlib>

namespace {

// KS (K split inside the workgroup): with few rows a wavefront streams its whole weight column alone and the launch is bound by
// the bytes ONE wavefront keeps in flight (8 KiB per ~1.5 us HBM round trip: 13 us for K = 1024 whatever M).  There the four
// wavefronts of a workgroup share one 16 x 16 output tile, take every 4th batch of k-groups and are summed through LDS in a
// fixed order — 4x the bytes in flight, a quarter of the dependent round trips.
template <int PR, int CR, bool LSTM, bool KS = false>
__global__ void __launch_bounds__(256) gemm_kernel(const gcpx_gemm_args a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int NT = a.N / 16;
    const int nt0 = KS ? blockIdx.y * CR : (blockIdx.y * 4 + wave) * CR;
    if (!KS && nt0 >= NT) return;
    const int rowblk = blockIdx.x;
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

    const int zb = blockIdx.z;
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
        constexpr int UK = (PR * CR >= 8) ? 2 : (PR * CR >= 4) ? 4 : 8;   // k-groups whose loads are issued together
        for (int kg = KS ? wave * UK : 0; kg < nkg; kg += (KS ? 4 : 1) * UK) {
            float4 b[UK][PR], w[UK][CR];
#pragma unroll
            for (int u = 0; u < UK; ++u) {
                const int k = (kg + u < nkg) ? kg + u : nkg - 1;
#pragma unroll
                for (int pt = 0; pt < PR; ++pt) b[u][pt] = *reinterpret_cast<const float4*>(bp[pt] + k * 16);
                const float4* wp = wbase + (size_t)(kg0 + k) * NT * 64;
#pragma unroll
                for (int ct = 0; ct < CR; ++ct) w[u][ct] = wp[ct * 64];
            }
#pragma unroll
            for (int u = 0; u < UK; ++u) {
                if (kg + u < nkg) {
#pragma unroll
                    for (int pt = 0; pt < PR; ++pt) {
                        float4 bb = b[u][pt];
                        if (xf) bb = affine_act4(bb, src.scale, src.shiftv, ((kg + u) * 16 + q * 4) & (src.cmod - 1), src.act);
                        bb.x *= mask[pt]; bb.y *= mask[pt]; bb.z *= mask[pt]; bb.w *= mask[pt];
                        b[u][pt] = bb;
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
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + (size_t)zb * a.z_bias_off + n);
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pt = 0; pt < PR; ++pt) {
            f32x4 v = acc[ct][pt];
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            if constexpr (LSTM) {
                if (rv[pt]) {
                    const int u = (nt0 + ct) * 4 + q;    // hidden unit of this lane; regs = gates i, f, g, o
                    const float cp = a.c_prev[(size_t)rr[pt] * a.c_prev_stride + u];
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

struct TileChoice { int pr, cr; };

TileChoice choose_tile(int M, int N, int nb = 1) {
    if (const char* ov = getenv("GCPX_GEMM_TILE")) {           // tuning aid: "pr,cr,minM" (also seen by gcpx_gemm_row_blocks)
        int pr = 0, cr = 0, mm = 0;
        if (sscanf(ov, "%d,%d,%d", &pr, &cr, &mm) == 3 && M >= mm && N % (16 * cr) == 0) return TileChoice{pr, cr};
    }
    const int prs[3] = {4, 2, 1}, crs[3] = {4, 2, 1};
    TileChoice best{1, 1};
    long best_wg = -1;
    static const long wg_min = getenv("GCPX_GEMM_WGMIN") ? atol(getenv("GCPX_GEMM_WGMIN")) : 200;
    // candidates by decreasing tile area; take the first that fills the chip, else the one with most workgroups
    for (int area = 16; area >= 1; area /= 2) {
        for (int pi = 0; pi < 3; ++pi)
            for (int ci = 0; ci < 3; ++ci) {
                const int pr = prs[pi], cr = crs[ci];
                if (pr * cr != area) continue;
                if (N % (16 * cr)) continue;
                if (pr > 1 && 16 * pr > ((M + 15) & ~15)) continue;      // row tiles that would be entirely masked
                const long rbk = (M + 16 * pr - 1) / (16 * pr);
                const long cbk = (N / 16 + 4 * cr - 1) / (4 * cr);
                const long wg = rbk * cbk * nb;
                if (wg >= wg_min) return TileChoice{pr, cr};
                if (wg > best_wg) { best_wg = wg; best = TileChoice{pr, cr}; }
            }
    }
    return best;
}

template <int PR, int CR>
void launch_t(const gcpx_gemm_args* a, hipStream_t stream) {
    const int rbk = (a->M + 16 * PR - 1) / (16 * PR);
    const int cbk = (a->N / 16 + 4 * CR - 1) / (4 * CR);
    const int nb = a->nbatch > 1 ? a->nbatch : 1;
    if (a->epi == GCPX_EPI_LSTM)
        hipLaunchKernelGGL((gemm_kernel<PR, CR, true>), dim3(rbk, cbk, nb), dim3(256), 0, stream, *a);
    else
        hipLaunchKernelGGL((gemm_kernel<PR, CR, false>), dim3(rbk, cbk, nb), dim3(256), 0, stream, *a);
}

}  // namespace

extern "C" int gcpx_gemm_row_blocks(int32_t M, int32_t N) {
    const TileChoice t = choose_tile(M, N);
    return (M + 16 * t.pr - 1) / (16 * t.pr);
}

extern "C" int gcpx_gemm(const gcpx_gemm_args* a, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(a != nullptr, "null args");
    GCPX_CHECK_ARG(a->nsrc >= 1 && a->nsrc <= 6, "nsrc out of range");
    GCPX_CHECK_ARG(a->M > 0 && a->N > 0 && a->N % 16 == 0 && a->rpb > 0, "bad M/N/rpb");
    int ksum = 0;
    for (int s = 0; s < a->nsrc; ++s) {
        GCPX_CHECK_ARG(a->src[s].ptr != nullptr, "source pointer is NULL");
        GCPX_CHECK_ARG(a->src[s].width > 0 && a->src[s].width % 16 == 0, "source width must be a multiple of 16");
        GCPX_CHECK_ARG(!(a->src[s].scale || a->src[s].act) ||
                           (a->src[s].cmod > 0 && (a->src[s].cmod & (a->src[s].cmod - 1)) == 0),
                       "cmod must be a power of two when scale/act is set");
        ksum += a->src[s].width;
    }
    GCPX_CHECK_ARG(ksum == a->K, "K != sum of source widths");
    GCPX_CHECK_ARG(a->wpk != nullptr, "weights missing");
    if (a->epi == GCPX_EPI_LSTM) {
        GCPX_CHECK_ARG(a->c_prev && a->h_out && a->c_out, "LSTM epilogue needs c_prev/h_out/c_out");
    } else {
        GCPX_CHECK_ARG(a->out != nullptr, "out is NULL");
    }
    GCPX_CHECK_ARG(a->nbatch <= 1 || (a->epi != GCPX_EPI_LSTM && !a->stats_partial), "nbatch > 1 only for plain epilogues");
    const TileChoice t = choose_tile(a->M, a->N, a->nbatch > 1 ? a->nbatch : 1);
    if (t.pr == 1 && t.cr == 1 && a->K >= 256 && !getenv("GCPX_GEMM_NOKS")) {
        // few rows: split K over the wavefronts of a workgroup when that still leaves the launch small
        const long nb = a->nbatch > 1 ? a->nbatch : 1;
        const long rbk = (a->M + 15) / 16, nt = a->N / 16;
        if (rbk * nt * nb <= 1024) {
            if (a->epi == GCPX_EPI_LSTM)
                hipLaunchKernelGGL((gemm_kernel<1, 1, true, true>), dim3(rbk, nt, nb), dim3(256), 0, stream, *a);
            else
                hipLaunchKernelGGL((gemm_kernel<1, 1, false, true>), dim3(rbk, nt, nb), dim3(256), 0, stream, *a);
            GCPX_CHECK_LAUNCH();
            return GCPX_OK;
        }
    }
    switch (t.pr * 10 + t.cr) {
        case 44: launch_t<4, 4>(a, stream); break;
        case 42: launch_t<4, 2>(a, stream); break;
        case 41: launch_t<4, 1>(a, stream); break;
        case 24: launch_t<2, 4>(a, stream); break;
        case 22: launch_t<2, 2>(a, stream); break;
        case 21: launch_t<2, 1>(a, stream); break;
        case 14: launch_t<1, 4>(a, stream); break;
        case 12: launch_t<1, 2>(a, stream); break;
        default: launch_t<1, 1>(a, stream); break;
    }
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
