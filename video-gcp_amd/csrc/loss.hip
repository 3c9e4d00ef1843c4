// Losses of the gcp_tree forward on gfx950 (all reductions deterministic: fixed trees, no atomics).
//
// Replaces, for matching_type='balanced':
//   decoder.nll(matched distr, traj_seq, weights=pad_mask)          /root/reference/gcp/prediction/models/tree/frame_binding.py:88-99
//   KLDivLoss2(q_z, p_z) over all tree nodes                        /root/reference/gcp/prediction/models/tree/inference.py:38-43
//   CELogitsLoss (length), BCELogitsLoss (existence), L2Loss (state) misc.py:53-56, frame_binding.py:80-86, base_gcp.py:281-286
//   get_total_loss                                                  /root/reference/gcp/prediction/models/base_gcp.py:294-304
// The loss formulas (blox.torch.losses is absent) follow DESIGN.md "Model spec": value = sum over non-batch dims of
// error * weight, mean over the batch.
#include "common.h"

namespace {

__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }
// hardware exp / reciprocal (1 - 2 ulp) for the VALU-bound mixture likelihood (same helpers as the fused value + gradient kernel in
// backward.hip, so the two paths return the same loss)
__device__ __forceinline__ float sigmoid_fast(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __frcp_rn(__expf(2.f * x) + 1.f); }

// ---------------------------------------------------------------------------------------------------
// discretised-logistic-mixture NLL of one frame per workgroup.
// params: [rows][H*W][pitch] in the head kernel's channel order (packing.dlm_channel_perm):
//   slot 8k..8k+7 = logit_k, mean_r, mean_g, mean_b, coeff0, coeff1, coeff2, log_scale_{r,k} ; slot dlm_ls_slot(c, k) (80..99, common.h) = log_scale_{c,k}, c = 1, 2
// A wavefront stages 16 pixels x pitch floats in LDS with coalesced 16-byte loads; lane (j = pixel, q) evaluates
// mixtures q, q+4, q+8 and the 4-lane column combines them with a logsumexp.
// ---------------------------------------------------------------------------------------------------
template <int NMIX, int PITCH>
__global__ void __launch_bounds__(256) dlm_nll_kernel(const float* __restrict__ params, const float* __restrict__ target,
                                                      const float* __restrict__ row_weight, float* __restrict__ nll_out,
                                                      const int npix) {
    __shared__ float4 stage4[4 * 16 * PITCH / 4];
    __shared__ float red[4];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    float acc = 0.f;
    if (row_weight == nullptr || row_weight[row] != 0.f) {
        float* st = reinterpret_cast<float*>(stage4) + wave * 16 * PITCH;
        const float* prow = params + (size_t)row * npix * PITCH;
        const float* trow = target + (size_t)row * 3 * npix;
        constexpr int F4 = 16 * PITCH / 4;
        for (int p0 = wave * 16; p0 < npix; p0 += 64) {
            const float4* src = reinterpret_cast<const float4*>(prow + (size_t)p0 * PITCH);
            for (int i = lane; i < F4; i += 64) reinterpret_cast<float4*>(st)[i] = src[i];
            __builtin_amdgcn_wave_barrier();
            const float* pp = st + j * PITCH;
            const float xr = trow[p0 + j], xg = trow[npix + p0 + j], xb = trow[2 * npix + p0 + j];
            // log-softmax denominators over all mixtures (every lane reads the 10 logits of its pixel)
            float lmax = pp[0];
#pragma unroll
            for (int k = 1; k < NMIX; ++k) lmax = fmaxf(lmax, pp[8 * k]);
            float lsum = 0.f;
#pragma unroll
            for (int k = 0; k < NMIX; ++k) lsum += expf(pp[8 * k] - lmax);
            const float lse_logits = lmax + logf(lsum);
            float lp[3];
            int nk = 0;
            for (int k = q; k < NMIX; k += 4, ++nk) {
                const float* m = pp + 8 * k;
                const float c0 = tanh_fast(m[4]), c1 = tanh_fast(m[5]), c2 = tanh_fast(m[6]);
                const float mean[3] = {m[1], m[2] + c0 * xr, m[3] + c1 * xr + c2 * xg};
                const float x[3] = {xr, xg, xb};
                float s = m[0] - lse_logits;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float ls = fmaxf(pp[dlm_ls_slot(c, k)], -7.f);
                    const float xc = x[c] - mean[c];
                    const float inv = __expf(-ls);
                    const float plus_in = inv * (xc + 1.f / 255.f), min_in = inv * (xc - 1.f / 255.f);
                    const float cdf_delta = sigmoid_fast(plus_in) - sigmoid_fast(min_in);
                    const float mid_in = inv * xc;
                    float v;
                    if (x[c] < -0.999f) v = plus_in - softplusf_(plus_in);
                    else if (x[c] > 0.999f) v = -softplusf_(min_in);
                    else if (cdf_delta > 1e-5f) v = logf(fmaxf(cdf_delta, 1e-12f));
                    else v = mid_in - ls - 2.f * softplusf_(mid_in) - 4.8481163864f;   // log(127.5)
                    s += v;
                }
                lp[nk] = s;
            }
            float mx = lp[0];
            for (int i = 1; i < nk; ++i) mx = fmaxf(mx, lp[i]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float se = 0.f;
            for (int i = 0; i < nk; ++i) se += expf(lp[i] - mx);
            se += __shfl_xor(se, 16);
            se += __shfl_xor(se, 32);
            if (q == 0) acc -= mx + logf(se);
            __builtin_amdgcn_wave_barrier();
        }
    }
    // deterministic reduction: 16 pixel lanes of q == 0, then the 4 waves
    acc = row16_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) nll_out[row] = (red[0] + red[1]) + (red[2] + red[3]);
}

// gaussian decoder: 0.5 * ((x - mu) / sigma)^2 + log_sigma + 0.5 * log(2 pi), summed over one frame per workgroup
__global__ void __launch_bounds__(256) gauss_nll_kernel(const float* __restrict__ mu, const float* __restrict__ target,
                                                        const float* __restrict__ log_sigma, float* __restrict__ nll_out,
                                                        const int nelem) {
    __shared__ float red[256];
    const int row = blockIdx.x;
    const float ls = log_sigma[0];
    const float inv = expf(-ls);
    float acc = 0.f;
    for (int i = threadIdx.x; i < nelem; i += 256) {
        const float d = (target[(size_t)row * nelem + i] - mu[(size_t)row * nelem + i]) * inv;
        acc += 0.5f * d * d + ls + 0.9189385332f;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) nll_out[row] = red[0];
}

// analytic KL(q || p) of diagonal Gaussians, clamped below at free_nats per dimension, summed per batch element
// one 1024-thread workgroup per sequence (deterministic sum), four dimensions per thread and trip: the 256-thread scalar
// version took 39 - 76 us at the serial tail of the forward with B workgroups on the chip
__device__ __forceinline__ void kl_sequence(const int b, const float* __restrict__ qz, const float* __restrict__ pz, const int N, const int nz,
                                            const long long batch_stride, const long long node_stride, const float free_nats,
                                            const float* __restrict__ node_weight, const long long weight_bstride,
                                            float* __restrict__ kl_out, float* red) {
    const int nz4 = nz / 4;
    float acc = 0.f;
    for (int i = threadIdx.x; i < N * nz4; i += 1024) {
        const int n = i / nz4, d = (i % nz4) * 4;
        const float* q = qz + (size_t)b * batch_stride + (size_t)n * node_stride;
        const float* p = pz + (size_t)b * batch_stride + (size_t)n * node_stride;
        const float4 mq = *reinterpret_cast<const float4*>(q + d), lq = *reinterpret_cast<const float4*>(q + nz + d);
        const float4 mp = *reinterpret_cast<const float4*>(p + d), lp = *reinterpret_cast<const float4*>(p + nz + d);
        const float wgt = node_weight ? node_weight[(size_t)b * weight_bstride + n] : 1.f;
        const float mqa[4] = {mq.x, mq.y, mq.z, mq.w}, lqa[4] = {lq.x, lq.y, lq.z, lq.w};
        const float mpa[4] = {mp.x, mp.y, mp.z, mp.w}, lpa[4] = {lp.x, lp.y, lp.z, lp.w};
        float s4 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float diff = mqa[k] - mpa[k];
            const float kl = lpa[k] - lqa[k] + (expf(2.f * lqa[k]) + diff * diff) / (2.f * expf(2.f * lpa[k])) - 0.5f;
            s4 += fmaxf(kl, free_nats);
        }
        acc += s4 * wgt;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) kl_out[b] = red[0];
}

__global__ void __launch_bounds__(1024) kl_kernel(const float* __restrict__ qz, const float* __restrict__ pz, const int N,
                                                  const int nz, const long long batch_stride, const long long node_stride,
                                                  const float free_nats, const float* __restrict__ node_weight,
                                                  const long long weight_bstride, float* __restrict__ kl_out) {
    __shared__ float red[1024];
    kl_sequence(blockIdx.x, qz, pz, N, nz, batch_stride, node_stride, free_nats, node_weight, weight_bstride, kl_out, red);
}

__device__ float block_sum(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// ---- the small loss terms, each a block-wide reduction over blockDim.x threads (red: blockDim.x floats) ----
// length prediction: cross entropy of seq_len_logits [B,T] against end_ind (misc.py:53-56)
__device__ float term_len_ce(const gcpx_loss_args& a, float* red) {
    if (!a.len_logits) return 0.f;
    const int B = a.B, T = a.T, tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
    float v = 0.f;
    for (int b = tid >> 6; b < B; b += nw) {            // one wavefront per sequence, lanes over the T logits
        const float* l = a.len_logits + (size_t)b * T;
        float m = -INFINITY;
        for (int t = lane; t < T; t += 64) m = fmaxf(m, l[t]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float s = 0.f;
        for (int t = lane; t < T; t += 64) s += expf(l[t] - m);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) v += m + logf(s) - l[a.end_ind[b]];
    }
    return block_sum(v, red) / B;
}
// existence: BCE with logits against the keep mask in depth-first order (frame_binding.py:80-86)
__device__ float term_exist_bce(const gcpx_loss_args& a, float* red) {
    if (!a.existence) return 0.f;
    const int B = a.B, N = a.N;
    float v = 0.f;
    for (int i = threadIdx.x; i < B * N; i += blockDim.x) {
        const float x = a.existence[i], y = a.leave[i] ? 1.f : 0.f;
        v += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
    }
    return block_sum(v, red) / (B * N);
}
// state regression: mean over [B, max_len, state_dim] of pad_mask * (pred - target)^2 (base_gcp.py:281-286)
__device__ float term_state_reg(const gcpx_loss_args& a, float* red) {
    if (!(a.regressed_state && a.state_target)) return 0.f;
    const int B = a.B, T = a.T;
    int maxlen = 0;
    for (int b = 0; b < B; ++b) maxlen = max(maxlen, a.seq_len[b]);
    float v = 0.f;
    const int sd = a.state_dim;
    const float* mask = a.state_mask ? a.state_mask : a.pad_mask;
    for (int i = threadIdx.x; i < B * maxlen * sd; i += blockDim.x) {
        const int d = i % sd, t = (i / sd) % maxlen, b = i / (sd * maxlen);
        const float e = a.regressed_state[((size_t)b * T + t) * sd + d] - a.state_target[((size_t)b * T + t) * sd + d];
        v += mask[b * T + t] * e * e;
    }
    return block_sum(v, red) / (B * maxlen * sd);
}
// inverse model, sampled pair: mean over [B, n_actions] of (pred - actions[b, t0[b]])^2 (inverse_mdl.py:181-191, weights = 1)
__device__ float term_action_reg(const gcpx_loss_args& a, float* red) {
    if (!a.action_pred) return 0.f;
    const int B = a.B, T = a.T, na = a.n_actions;
    float v = 0.f;
    for (int i = threadIdx.x; i < B * na; i += blockDim.x) {
        const int b = i / na, d = i % na;
        const float e = a.action_pred[i] - a.action_seq[((size_t)b * (T - 1) + (int)a.inv_t0[b]) * na + d];
        v += e * e;
    }
    return block_sum(v, red) / (B * na);
}
// cost model: mean over [B, 1] of (cost - gt_cost)^2 (cost_mdl.py:59-62)
__device__ float term_cost_reg(const gcpx_loss_args& a, float* red) {
    if (!a.cost_pred) return 0.f;
    const int B = a.B;
    float v = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float e = a.cost_pred[i] - a.cost_target[i];
        v += e * e;
    }
    return block_sum(v, red) / B;
}

// PRE = false: one workgroup, every term and the weighted total.  PRE = true: the reconstruction / KL sums and the total only; the
// five latent-side terms were left in out[2], [3], [4], [7], [8] by loss_pre_kernel (they do not depend on the decoder, so they are
// off the serial tail of the forward)
template <bool PRE>
__global__ void __launch_bounds__(256) loss_combine_kernel(const gcpx_loss_args a) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int B = a.B, T = a.T;
    // dense_img_rec: sum_bt pad_mask * nll / B
    float v = 0.f;
    for (int i = tid; i < B * T; i += 256) v += a.nll_bt[i] * a.pad_mask[i];
    const float rec = block_sum(v, red) / B;
    v = 0.f;
    if (a.kl_b)
        for (int i = tid; i < B; i += 256) v += a.kl_b[i];
    const float kl = a.kl_b ? block_sum(v, red) / B : 0.f;
    float ce, bce, sreg, areg, creg;
    if constexpr (PRE) {
        ce = a.out[2]; bce = a.out[3]; sreg = a.out[4]; areg = a.out[7]; creg = a.out[8];
    } else {
        ce = term_len_ce(a, red); bce = term_exist_bce(a, red); sreg = term_state_reg(a, red);
        areg = term_action_reg(a, red); creg = term_cost_reg(a, red);
    }
    if (tid == 0) {
        a.out[7] = areg; a.out[8] = creg;
        a.out[0] = rec; a.out[1] = kl; a.out[2] = ce; a.out[3] = bce; a.out[4] = sreg;
        float total = 0.f;
        if (a.w_rec > 0.f) total += a.w_rec * rec;
        const float w_kl = a.w_kl_dev ? *a.w_kl_dev : a.w_kl;       // (burn-in schedule: the current weight lives in device memory)
        if (w_kl > 0.f) total += w_kl * kl;
        if (a.w_len > 0.f) total += a.w_len * ce;
        if (a.w_exist > 0.f) total += a.w_exist * bce;
        if (a.w_state > 0.f) total += a.w_state * sreg;
        if (a.w_action > 0.f) total += a.w_action * areg;
        if (a.w_cost > 0.f) total += a.w_cost * creg;
        a.out[5] = total / a.total_div;      // base_gcp.py:299-301: / prod(traj_seq.shape[1:])
        a.out[6] = rec + kl;                 // nll upper bound (base_gcp.py:289-290)
    }
}

// Everything of the loss that needs no decoded frame, in ONE launch in front of the decoder: workgroups 0 .. B-1 the KL of one
// sequence each (kl_kernel), workgroups B .. B+4 one latent-side term each
__global__ void __launch_bounds__(1024) loss_pre_kernel(const gcpx_loss_args a, const float* __restrict__ qz, const float* __restrict__ pz,
                                                        const int N, const int nz, const long long batch_stride, const long long node_stride,
                                                        const float free_nats, const float* __restrict__ node_weight,
                                                        const long long weight_bstride, float* __restrict__ kl_out) {
    __shared__ float red[1024];
    const int blk = blockIdx.x;
    if (blk < a.B) {
        kl_sequence(blk, qz, pz, N, nz, batch_stride, node_stride, free_nats, node_weight, weight_bstride, kl_out, red);
        return;
    }
    float r;
    int slot;
    switch (blk - a.B) {
        case 0: r = term_len_ce(a, red); slot = 2; break;
        case 1: r = term_exist_bce(a, red); slot = 3; break;
        case 2: r = term_state_reg(a, red); slot = 4; break;
        case 3: r = term_action_reg(a, red); slot = 7; break;
        default: r = term_cost_reg(a, red); slot = 8; break;
    }
    if (threadIdx.x == 0) a.out[slot] = r;
}

}  // namespace

extern "C" int gcpx_dlm_nll(const float* params, const float* target, const float* row_weight, float* nll_out,
                            int32_t rows, int32_t npix, int32_t pitch, int32_t n_mix, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(params && target && nll_out && rows > 0, "null pointer / rows <= 0");
    GCPX_CHECK_ARG(npix % 64 == 0, "pixels per frame must be a multiple of 64");
    if (n_mix != 10 || pitch != 112) {
        gcpx_set_error("gcpx_dlm_nll: only n_mix=10 with the 112-slot head layout is built");
        return GCPX_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL((dlm_nll_kernel<10, 112>), dim3(rows), dim3(256), 0, stream, params, target, row_weight, nll_out, npix);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_gauss_nll(const float* mu, const float* target, const float* log_sigma, float* nll_out, int32_t rows,
                              int32_t nelem, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(mu && target && log_sigma && nll_out && rows > 0 && nelem > 0, "null pointer / bad sizes");
    hipLaunchKernelGGL(gauss_nll_kernel, dim3(rows), dim3(256), 0, stream, mu, target, log_sigma, nll_out, nelem);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_kl_gauss(const float* qz, const float* pz, int32_t B, int32_t N, int32_t nz, int64_t batch_stride,
                             int64_t node_stride, float free_nats, const float* node_weight, int64_t weight_bstride,
                             float* kl_out, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(qz && pz && kl_out && B > 0 && N > 0 && nz > 0, "null pointer / bad sizes");
    GCPX_CHECK_ARG(nz % 4 == 0 && batch_stride % 4 == 0 && node_stride % 4 == 0, "nz and the strides must be multiples of 4");
    hipLaunchKernelGGL(kl_kernel, dim3(B), dim3(1024), 0, stream, qz, pz, N, nz, (long long)batch_stride,
                       (long long)node_stride, free_nats, node_weight, (long long)weight_bstride, kl_out);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_loss_combine(const gcpx_loss_args* a, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(a && a->nll_bt && a->pad_mask && a->out, "null pointer");
    GCPX_CHECK_ARG(a->B > 0 && a->T > 0 && a->total_div > 0, "bad sizes");
    hipLaunchKernelGGL(loss_combine_kernel<false>, dim3(1), dim3(256), 0, stream, *a);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_loss_pre(const gcpx_loss_args* a, const float* qz, const float* pz, int32_t N, int32_t nz, int64_t batch_stride,
                             int64_t node_stride, float free_nats, const float* node_weight, int64_t weight_bstride, float* kl_out,
                             void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(a && a->out && a->B > 0 && a->T > 0, "null pointer / bad sizes");
    GCPX_CHECK_ARG(qz && pz && kl_out && N > 0 && nz > 0, "null pointer / bad sizes");
    GCPX_CHECK_ARG(nz % 4 == 0 && batch_stride % 4 == 0 && node_stride % 4 == 0, "nz and the strides must be multiples of 4");
    hipLaunchKernelGGL(loss_pre_kernel, dim3(a->B + 5), dim3(1024), 0, stream, *a, qz, pz, N, nz, (long long)batch_stride,
                       (long long)node_stride, free_nats, node_weight, (long long)weight_bstride, kl_out);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_loss_final(const gcpx_loss_args* a, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(a && a->nll_bt && a->pad_mask && a->out, "null pointer");
    GCPX_CHECK_ARG(a->B > 0 && a->T > 0 && a->total_div > 0, "bad sizes");
    hipLaunchKernelGGL(loss_combine_kernel<true>, dim3(1), dim3(256), 0, stream, *a);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
