// Fused backward of the Predictor MLP on gfx950 (data-gradient chain): the mirror image of mlp.hip.
//   d head output -> (head^T) -> [GroupNorm + LReLU backward -> (mid_l^T)] x n_mid -> LReLU backward -> (input^T, per input split)
// GroupNorm is per row, so the whole chain is row-local: one 256-thread workgroup carries 16 rows through every stage, the
// four wavefronts split the 128 (32) columns of a hidden layer, d(activation) tiles go through LDS between the GEMMs.  The
// intermediate gradients du_l (inputs of the weight-gradient GEMMs, which stay separate launches on the side lanes) and the
// per-workgroup GroupNorm parameter sums are written on the way.  Replaces 10 dependent launches per Predictor
// (gcpx_gemm x 6, gcpx_gn_lrelu_bwd x 3, gcpx_lrelu_bwd) on the latency-bound chain of a tree level:
//   backward of /root/reference/gcp/prediction/models/tree/tree_module.py:77 (prior), inference.py:27-35 (posterior),
//   which the reference gets from torch autograd (train.py:157-160).
#include "common.h"

namespace {

template <int MID>
__device__ __forceinline__ void mlp_bwd_body(const gcpx_mlp_bwd_args& a, const int bx) {
    constexpr int NTM = MID / 16;
    constexpr int NW = NTM >= 4 ? 4 : NTM;
    constexpr int TPW = NTM / NW;
    constexpr int PITCH = MID + 4;
    constexpr int CPG = MID / 8;                     // channels per GroupNorm group (gn_groups = 8)
    static_assert(CPG == 4 || CPG == 16, "GroupNorm group must be one lane (4) or one 16-channel tile");
    __shared__ float4 hid4[16 * PITCH / 4];
    float* hid = reinterpret_cast<float*>(hid4);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int r = bx * 16 + j;
    const bool rv = r < a.M;
    const int rs = rv ? r : 0;
    const int rb = rs / a.rpb, rj = rs % a.rpb;
    const float slope = a.lrelu_slope;
    const bool owner = wave < NW;
    const int nt0 = wave * TPW;
    const float mask = rv ? 1.f : 0.f;

    f32x4 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0, 0, 0, 0};

    // ---- d(last hidden activation) = dout @ W_out  (pack of W_out^T: [out_pad / 16][NTM][64] float4) ----
    if (owner) {
        const float4* wbase = reinterpret_cast<const float4*>(a.wT_out) + (size_t)nt0 * 64 + lane;
        const float* bp = a.dout + (size_t)rs * a.ldo + q * 4;
        const int nkg = a.out_pad / 16;
        for (int kg = 0; kg < nkg; kg += 4) {
            float4 b[4], w[4][TPW];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = (kg + u < nkg) ? kg + u : nkg - 1;
                b[u] = *reinterpret_cast<const float4*>(bp + k * 16);
#pragma unroll
                for (int t = 0; t < TPW; ++t) w[u][t] = wbase[((size_t)k * NTM + t) * 64];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (kg + u < nkg) {
                    float4 bb = b[u];
                    bb.x *= mask; bb.y *= mask; bb.z *= mask; bb.w *= mask;
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        acc[t] = mfma16(w[u][t].x, bb.x, acc[t]);
                        acc[t] = mfma16(w[u][t].y, bb.y, acc[t]);
                        acc[t] = mfma16(w[u][t].z, bb.z, acc[t]);
                        acc[t] = mfma16(w[u][t].w, bb.w, acc[t]);
                    }
                }
            }
        }
    }

    // ---- hidden layers, last to first: GroupNorm + LReLU backward on the accumulator tile, then du @ W_mid_l ----
    for (int l = a.n_mid - 1; l >= 0; --l) {
        if (owner) {
            const float* ul = a.save + (size_t)(1 + 2 * l) * a.M * MID + (size_t)rs * MID;
            float* dul = a.du[1 + l];
            float* part = a.gn_partial[l] + (size_t)bx * 2 * MID;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int c = (nt0 + t) * 16 + q * 4;
                const float4 uv = *reinterpret_cast<const float4*>(ul + c);
                const float4 gv = *reinterpret_cast<const float4*>(a.gn_gamma[l] + c);
                const float4 be = *reinterpret_cast<const float4*>(a.gn_beta[l] + c);
                float sum = (uv.x + uv.y) + (uv.z + uv.w);
                if (CPG == 16) { sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32); }
                const float mean = sum * (1.f / CPG);
                const float d0 = uv.x - mean, d1 = uv.y - mean, d2 = uv.z - mean, d3 = uv.w - mean;
                float ss = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                if (CPG == 16) { ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32); }
                const float rstd = rsqrtf(ss * (1.f / CPG) + a.gn_eps);
                const float xh[4] = {d0 * rstd, d1 * rstd, d2 * rstd, d3 * rstd};
                const float g[4] = {gv.x, gv.y, gv.z, gv.w}, bt[4] = {be.x, be.y, be.z, be.w};
                float dxh[4], gs[4], bs[4];
                float m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float y = xh[i] * g[i] + bt[i];
                    const float dy = acc[t][i] * (y > 0.f ? 1.f : slope);      // rows >= M carry acc = 0
                    gs[i] = dy * xh[i];
                    bs[i] = dy;
                    dxh[i] = dy * g[i];
                    m1 += dxh[i];
                    m2 += dxh[i] * xh[i];
                }
                if (CPG == 16) {
                    m1 += __shfl_xor(m1, 16); m1 += __shfl_xor(m1, 32);
                    m2 += __shfl_xor(m2, 16); m2 += __shfl_xor(m2, 32);
                }
                m1 *= (1.f / CPG); m2 *= (1.f / CPG);
                const float4 du = make_float4(rstd * (dxh[0] - m1 - xh[0] * m2), rstd * (dxh[1] - m1 - xh[1] * m2),
                                              rstd * (dxh[2] - m1 - xh[2] * m2), rstd * (dxh[3] - m1 - xh[3] * m2));
                *reinterpret_cast<float4*>(hid + j * PITCH + c) = du;
                if (rv) *reinterpret_cast<float4*>(dul + (size_t)r * MID + c) = du;
                // d gamma / d beta: sums over the 16 rows of the workgroup (lanes j = 0..15 of a 16-lane row)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int m = 1; m < 16; m <<= 1) {
                        gs[i] += __shfl_xor(gs[i], m);
                        bs[i] += __shfl_xor(bs[i], m);
                    }
                }
                if (j == 0) {
                    *reinterpret_cast<float4*>(part + c) = make_float4(gs[0], gs[1], gs[2], gs[3]);
                    *reinterpret_cast<float4*>(part + MID + c) = make_float4(bs[0], bs[1], bs[2], bs[3]);
                }
            }
        }
        __syncthreads();
        if (owner) {
            const float4* wbase = reinterpret_cast<const float4*>(a.wT_mid[l]) + (size_t)nt0 * 64 + lane;
            float4 w[NTM][TPW], b[NTM];
#pragma unroll
            for (int kg = 0; kg < NTM; ++kg) {
#pragma unroll
                for (int t = 0; t < TPW; ++t) w[kg][t] = wbase[(kg * NTM + t) * 64];
                b[kg] = *reinterpret_cast<const float4*>(hid + j * PITCH + kg * 16 + q * 4);
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kg = 0; kg < NTM; ++kg) {
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    acc[t] = mfma16(w[kg][t].x, b[kg].x, acc[t]);
                    acc[t] = mfma16(w[kg][t].y, b[kg].y, acc[t]);
                    acc[t] = mfma16(w[kg][t].z, b[kg].z, acc[t]);
                    acc[t] = mfma16(w[kg][t].w, b[kg].w, acc[t]);
                }
            }
        }
        __syncthreads();
    }

    // ---- LReLU backward of the input layer ----
    if (owner) {
        const float* a0 = a.save + (size_t)rs * MID;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int c = (nt0 + t) * 16 + q * 4;
            const float4 av = *reinterpret_cast<const float4*>(a0 + c);
            const float4 du = make_float4(acc[t][0] * (av.x > 0.f ? 1.f : slope), acc[t][1] * (av.y > 0.f ? 1.f : slope),
                                          acc[t][2] * (av.z > 0.f ? 1.f : slope), acc[t][3] * (av.w > 0.f ? 1.f : slope));
            *reinterpret_cast<float4*>(hid + j * PITCH + c) = du;
            if (rv) *reinterpret_cast<float4*>(a.du[0] + (size_t)r * MID + c) = du;
        }
    }
    __syncthreads();

    // ---- input splits: dx_i = du0 @ W_in[:, split i]; column tiles dealt round-robin over the four wavefronts ----
    float4 b[NTM];
#pragma unroll
    for (int kg = 0; kg < NTM; ++kg) b[kg] = *reinterpret_cast<const float4*>(hid + j * PITCH + kg * 16 + q * 4);
    for (int i = 0; i < a.ndx; ++i) {
        const int NTO = a.dx[i].width / 16;
        const float4* wbase = reinterpret_cast<const float4*>(a.dx[i].wT) + lane;
        float* orow = a.dx[i].out + (size_t)rb * a.dx[i].ob + (size_t)rj * a.dx[i].orow;
        for (int nt = wave; nt < NTO; nt += 4) {
            float4 w[NTM];
#pragma unroll
            for (int kg = 0; kg < NTM; ++kg) w[kg] = wbase[(kg * NTO + nt) * 64];
            f32x4 ao = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kg = 0; kg < NTM; ++kg) {
                ao = mfma16(w[kg].x, b[kg].x, ao);
                ao = mfma16(w[kg].y, b[kg].y, ao);
                ao = mfma16(w[kg].z, b[kg].z, ao);
                ao = mfma16(w[kg].w, b[kg].w, ao);
            }
            if (rv) *reinterpret_cast<float4*>(orow + nt * 16 + q * 4) = make_float4(ao[0], ao[1], ao[2], ao[3]);
        }
    }
}

template <int MID>
__global__ void __launch_bounds__(256) mlp_bwd_kernel(const gcpx_mlp_bwd_args a) { mlp_bwd_body<MID>(a, blockIdx.x); }

// several Predictors whose chains are independent (a tree level's posterior and prior) side by side: blockIdx.y = the Predictor.
// The same function per workgroup: the same bits as one launch each.
struct MlpBwdGroup { gcpx_mlp_bwd_args p[GCPX_MLP_BWD_GROUP_MAX]; };
template <int MID>
__global__ void __launch_bounds__(256) mlp_bwd_group_kernel(const MlpBwdGroup g) {
    const gcpx_mlp_bwd_args& a = g.p[blockIdx.y];
    if ((int)blockIdx.x * 16 >= a.M) return;
    mlp_bwd_body<MID>(a, blockIdx.x);
}

int check_mlp_bwd(const gcpx_mlp_bwd_args* a, const char* fn) {
#define CHK(cond, msg)                                     \
    do {                                                   \
        if (!(cond)) {                                     \
            gcpx_set_error("%s: %s", fn, msg);             \
            return GCPX_ERR_INVALID_ARG;                   \
        }                                                  \
    } while (0)
    CHK(a != nullptr, "null args");
    CHK(a->M > 0 && a->rpb > 0, "bad M / rpb");
    CHK(a->dout && a->save && a->wT_out && a->du[0], "missing pointer");
    CHK(a->out_pad > 0 && a->out_pad % 16 == 0 && a->ldo >= a->out_pad && a->ldo % 4 == 0, "out_pad % 16, ldo");
    CHK(a->n_mid >= 0 && a->n_mid <= 4 && a->ndx >= 0 && a->ndx <= 4, "n_mid / ndx out of range");
    for (int l = 0; l < a->n_mid; ++l)
        CHK(a->wT_mid[l] && a->gn_gamma[l] && a->gn_beta[l] && a->du[1 + l] && a->gn_partial[l], "hidden-layer pointer missing");
    for (int i = 0; i < a->ndx; ++i)
        CHK(a->dx[i].wT && a->dx[i].out && a->dx[i].width > 0 && a->dx[i].width % 16 == 0 && a->dx[i].orow % 4 == 0 && a->dx[i].ob % 4 == 0,
            "input split: pointers, width % 16, 16-byte aligned rows");
#undef CHK
    return GCPX_OK;
}

}  // namespace

extern "C" int gcpx_mlp_bwd_group(const gcpx_mlp_bwd_args* tab, int32_t nprob, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(tab && nprob >= 1 && nprob <= GCPX_MLP_BWD_GROUP_MAX, "1 .. GCPX_MLP_BWD_GROUP_MAX problems");
    MlpBwdGroup g;
    int mx = 0;
    for (int i = 0; i < nprob; ++i) {
        const int st = check_mlp_bwd(tab + i, __func__);
        if (st != GCPX_OK) return st;
        GCPX_CHECK_ARG(tab[i].mid == tab[0].mid, "one hidden width per group");
        g.p[i] = tab[i];
        mx = tab[i].M > mx ? tab[i].M : mx;
    }
    for (int i = nprob; i < GCPX_MLP_BWD_GROUP_MAX; ++i) g.p[i] = tab[0];
    const dim3 grid((mx + 15) / 16, nprob);
    if (tab[0].mid == 128) hipLaunchKernelGGL(mlp_bwd_group_kernel<128>, grid, dim3(256), 0, stream, g);
    else if (tab[0].mid == 32) hipLaunchKernelGGL(mlp_bwd_group_kernel<32>, grid, dim3(256), 0, stream, g);
    else {
        gcpx_set_error("gcpx_mlp_bwd_group: unsupported mid=%d (128 or 32)", tab[0].mid);
        return GCPX_ERR_UNSUPPORTED;
    }
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_mlp_bwd_blocks(int32_t M) { return (M + 15) / 16; }

extern "C" int gcpx_mlp_bwd(const gcpx_mlp_bwd_args* a, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int st = check_mlp_bwd(a, __func__);
    if (st != GCPX_OK) return st;
    const int gx = (a->M + 15) / 16;
    if (a->mid == 128) hipLaunchKernelGGL(mlp_bwd_kernel<128>, dim3(gx), dim3(256), 0, stream, *a);
    else if (a->mid == 32) hipLaunchKernelGGL(mlp_bwd_kernel<32>, dim3(gx), dim3(256), 0, stream, *a);
    else {
        gcpx_set_error("gcpx_mlp_bwd: unsupported mid=%d (128 or 32)", a->mid);
        return GCPX_ERR_UNSUPPORTED;
    }
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
