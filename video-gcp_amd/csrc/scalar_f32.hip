// The kernels of the training step that must not carry packed-f32 VALU instructions: the optimizer updates (gcp_builder.py:174-186 over
// train.py:155-163), the likelihood backward from stored parameters (decoder.nll, frame_binding.py:88-99) and the mixture-mean backward.
// This file is built with -fno-slp-vectorize (csrc/build.sh).  Round 4 found ~1e-7 of the head's gradient values stored as +-0, every one
// the LOW result of a v_pk_mul_f32 whose low lane read the HIGH half of a register pair (profiles/r05_head_store_hazard.txt; no isolated
// reproducer, root cause open: hipcc of ROCm 7.2.0, AMD clang 20.0.0git).  Under SLP vectorisation these five kernels compiled to the
// same form (56 such instructions); they write parameters, moments and gradients on every step, so they are kept scalar instead of
// watched.  tests/test_isa_scan_cpu.py asserts that the whole library holds no such instruction.
#include "common.h"
#include <algorithm>

namespace {

// hardware exp / reciprocal (1 - 2 ulp): the mixture-likelihood kernels are VALU bound and full-precision expf / IEEE division cost
// ~10 instructions each (16 division sequences and 37 exponentials in the loop body of dlm_nll_bwd_kernel)
__device__ __forceinline__ float sigmoid_fast(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __frcp_rn(__expf(2.f * x) + 1.f); }

// Mirror of dlm_nll_kernel (loss.hip): a wavefront stages 16 pixels x PITCH parameters in LDS, lane (j = pixel, q)
// owns mixtures q, q+4, q+8, overwrites its slots with the gradients, and the tile goes out with coalesced stores.
template <int NMIX, int PITCH>
__global__ void __launch_bounds__(256) dlm_nll_bwd_kernel(const float* __restrict__ params, const float* __restrict__ target,
                                                          const float* __restrict__ row_weight, const float scale,
                                                          float* __restrict__ dparams, float* __restrict__ colsum,
                                                          float* __restrict__ nll_out, const int npix) {
    __shared__ float4 stage4[4 * 16 * PITCH / 4];
    __shared__ float csum[4][2][64];
    __shared__ float nred[4];
    float cs0 = 0.f, cs1 = 0.f;                    // column sums of this row's gradients: columns lane, lane + 64
    float nacc = 0.f;                              // the row's negative log-likelihood (training: forward and backward in one pass)
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    constexpr int F4 = 16 * PITCH / 4;
    float* st = reinterpret_cast<float*>(stage4) + wave * 16 * PITCH;
    float* drow = dparams + (size_t)row * npix * PITCH;
    const float coef = (row_weight ? row_weight[row] : 1.f) * scale;
    if (coef == 0.f) {
        for (int p0 = wave * 16; p0 < npix; p0 += 64) {
            float4* dst = reinterpret_cast<float4*>(drow + (size_t)p0 * PITCH);
            for (int i = lane; i < F4; i += 64) dst[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (colsum)
            for (int i = tid; i < PITCH; i += 256) colsum[(size_t)row * PITCH + i] = 0.f;
        if (nll_out && tid == 0) nll_out[row] = 0.f;
        return;
    }
    const float* prow = params + (size_t)row * npix * PITCH;
    const float* trow = target + (size_t)row * 3 * npix;
    for (int p0 = wave * 16; p0 < npix; p0 += 64) {
        const float4* src = reinterpret_cast<const float4*>(prow + (size_t)p0 * PITCH);
        for (int i = lane; i < F4; i += 64) reinterpret_cast<float4*>(st)[i] = src[i];
        __builtin_amdgcn_wave_barrier();
        float* pp = st + j * PITCH;
        const float xr = trow[p0 + j], xg = trow[npix + p0 + j], xb = trow[2 * npix + p0 + j];
        float lmax = pp[0];
#pragma unroll
        for (int k = 1; k < NMIX; ++k) lmax = fmaxf(lmax, pp[8 * k]);
        float lsum = 0.f;
#pragma unroll
        for (int k = 0; k < NMIX; ++k) lsum += expf(pp[8 * k] - lmax);
        const float lse_logits = lmax + logf(lsum);
        float lp[3], gm[3][3], gs[3][3], cf[3][3], lg[3];
        int nk = 0;
        for (int k = q; k < NMIX; k += 4, ++nk) {
            const float* m = pp + 8 * k;
            const float c0 = tanh_fast(m[4]), c1 = tanh_fast(m[5]), c2 = tanh_fast(m[6]);
            cf[nk][0] = c0; cf[nk][1] = c1; cf[nk][2] = c2;
            const float mean[3] = {m[1], m[2] + c0 * xr, m[3] + c1 * xr + c2 * xg};
            const float x[3] = {xr, xg, xb};
            lg[nk] = m[0];
            float s = m[0] - lse_logits;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float raw = pp[dlm_ls_slot(c, k)];     // packing.dlm_log_scale_slot
                const float ls = fmaxf(raw, -7.f);
                const float xc = x[c] - mean[c];
                const float inv = __expf(-ls);
                const float plus_in = inv * (xc + 1.f / 255.f), min_in = inv * (xc - 1.f / 255.f);
                const float sp = sigmoid_fast(plus_in), sm = sigmoid_fast(min_in);
                const float cdf_delta = sp - sm;
                const float mid_in = inv * xc;
                float v, dm, ds;     // value, d v / d mean, d v / d log_scale
                if (x[c] < -0.999f) {
                    v = plus_in - (plus_in > 20.f ? plus_in : log1pf(expf(plus_in)));
                    dm = -inv * (1.f - sp);
                    ds = -plus_in * (1.f - sp);
                } else if (x[c] > 0.999f) {
                    v = -(min_in > 20.f ? min_in : log1pf(expf(min_in)));
                    dm = inv * sm;
                    ds = min_in * sm;
                } else if (cdf_delta > 1e-5f) {
                    v = logf(fmaxf(cdf_delta, 1e-12f));
                    const float pp_ = sp * (1.f - sp), pm_ = sm * (1.f - sm);
                    const float rcd = __frcp_rn(cdf_delta);
                    dm = -inv * (pp_ - pm_) * rcd;
                    ds = -(plus_in * pp_ - min_in * pm_) * rcd;
                } else {
                    const float smid = sigmoid_fast(mid_in);
                    v = mid_in - ls - 2.f * (mid_in > 20.f ? mid_in : log1pf(expf(mid_in))) - 4.8481163864f;
                    dm = -inv * (1.f - 2.f * smid);
                    ds = -mid_in * (1.f - 2.f * smid) - 1.f;
                }
                if (raw < -7.f) ds = 0.f;          // clamp(min=-7) blocks the gradient
                s += v;
                gm[nk][c] = dm;
                gs[nk][c] = ds;
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
        const float inv_se = __frcp_rn(se);
        if (q == 0) nacc -= mx + logf(se);
        __builtin_amdgcn_wave_barrier();           // every lane has read the logits of its pixel
        nk = 0;
        for (int k = q; k < NMIX; k += 4, ++nk) {
            const float w = expf(lp[nk] - mx) * inv_se;            // responsibility of mixture k
            const float pik = expf(lg[nk] - lse_logits);
            float* m = pp + 8 * k;
            const float gw = -coef * w;                             // d(-logsumexp)/d s_k
            m[0] = coef * (pik - w);
            m[1] = gw * gm[nk][0];
            m[2] = gw * gm[nk][1];
            m[3] = gw * gm[nk][2];
            m[4] = gw * gm[nk][1] * xr * (1.f - cf[nk][0] * cf[nk][0]);
            m[5] = gw * gm[nk][2] * xr * (1.f - cf[nk][1] * cf[nk][1]);
            m[6] = gw * gm[nk][2] * xg * (1.f - cf[nk][2] * cf[nk][2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) pp[dlm_ls_slot(c, k)] = gw * gs[nk][c];
        }
        if (q == 0)
            for (int s = 80 + 2 * NMIX; s < PITCH; ++s) pp[s] = 0.f;
        __builtin_amdgcn_wave_barrier();
        float4* dst = reinterpret_cast<float4*>(drow + (size_t)p0 * PITCH);
        for (int i = lane; i < F4; i += 64) dst[i] = reinterpret_cast<float4*>(st)[i];
        if (colsum) {
#pragma unroll
            for (int px = 0; px < 16; ++px) {
                cs0 += st[px * PITCH + lane];
                if (lane + 64 < PITCH) cs1 += st[px * PITCH + lane + 64];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (colsum) {
        csum[wave][0][lane] = cs0;
        csum[wave][1][lane] = cs1;
        __syncthreads();
        for (int i = tid; i < PITCH; i += 256) {
            const int h = i >> 6, l = i & 63;
            colsum[(size_t)row * PITCH + i] = (csum[0][h][l] + csum[1][h][l]) + (csum[2][h][l] + csum[3][h][l]);
        }
    }
    if (nll_out) {                                  // same reduction order as dlm_nll_kernel (loss.hip)
        nacc = row16_sum(nacc);
        if (lane == 0) nred[wave] = nacc;
        __syncthreads();
        if (tid == 0) nll_out[row] = (nred[0] + nred[1]) + (nred[2] + nred[3]);
    }
}


// Backward of the mixture MEAN (images of the discrete-logistic-mixture head; oracle dlm_mean): params [F][npix][PITCH] in the
// head's slot order, dimg NCHW [F][3][npix] -> dparams [F][npix][PITCH] (+ per-frame column sums for the bias gradient).
// A wavefront stages 16 pixels x PITCH parameters in LDS; lane (j = pixel, q) owns mixtures q, q + 4, q + 8.
// USED: slots of a pixel that are read and written.  The mean depends on slots 0 .. 8 NMIX - 1 only (logit, means, colour coefficients of
// every mixture; the green / blue log-scales live behind them, packing.dlm_log_scale_slot): USED = 8 NMIX moves 80 of 112 floats per
// pixel each way and leaves slots >= USED of dparams UNWRITTEN — for callers whose data / weight gradient kernels walk the leading USED
// channels only (the adaptive model's head backward); colsum's slots >= USED are written as 0.
template <int NMIX, int PITCH, int USED>
__global__ void __launch_bounds__(256) dlm_mean_bwd_kernel(const float* __restrict__ params, const float* __restrict__ dimg,
                                                           float* __restrict__ dparams, float* __restrict__ colsum, const int npix) {
    static_assert(USED == PITCH || USED == 8 * NMIX, "all slots or the 8 NMIX slots of the mean");
    __shared__ float4 stage4[4 * 16 * PITCH / 4];
    __shared__ float csum[4][2][64];
    float cs0 = 0.f, cs1 = 0.f;
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    constexpr int F4 = 16 * PITCH / 4;
    float* st = reinterpret_cast<float*>(stage4) + wave * 16 * PITCH;
    const float* prow = params + (size_t)row * npix * PITCH;
    const float* grow = dimg + (size_t)row * 3 * npix;
    float* drow = dparams + (size_t)row * npix * PITCH;
    for (int p0 = wave * 16; p0 < npix; p0 += 64) {
        const float4* src = reinterpret_cast<const float4*>(prow + (size_t)p0 * PITCH);
        if constexpr (USED == PITCH) {
            for (int i = lane; i < F4; i += 64) reinterpret_cast<float4*>(st)[i] = src[i];
        } else {
            constexpr int U4 = USED / 4;
            for (int i = lane; i < 16 * U4; i += 64) {
                const int px = i / U4, c4 = i - px * U4;
                reinterpret_cast<float4*>(st)[px * (PITCH / 4) + c4] = src[px * (PITCH / 4) + c4];
            }
        }
        __builtin_amdgcn_wave_barrier();
        float* pp = st + j * PITCH;
        float lmax = pp[0];
#pragma unroll
        for (int k = 1; k < NMIX; ++k) lmax = fmaxf(lmax, pp[8 * k]);
        float lsum = 0.f;
#pragma unroll
        for (int k = 0; k < NMIX; ++k) lsum += __expf(pp[8 * k] - lmax);
        const float rls = __frcp_rn(lsum);
        float pi[3], Mr[3], Mg[3], Mb[3], cf[3][3];
        float Sr = 0.f, Sg = 0.f, Sb = 0.f;
        int nk = 0;
        for (int k = q; k < NMIX; k += 4, ++nk) {
            const float* m = pp + 8 * k;
            const float c0 = tanh_fast(m[4]), c1 = tanh_fast(m[5]), c2 = tanh_fast(m[6]);   // the head's forward uses the same
            cf[nk][0] = c0; cf[nk][1] = c1; cf[nk][2] = c2;
            pi[nk] = __expf(m[0] - lmax) * rls;
            Mr[nk] = m[1];
            Mg[nk] = m[2] + c0 * Mr[nk];
            Mb[nk] = m[3] + c1 * Mr[nk] + c2 * Mg[nk];
            Sr += pi[nk] * Mr[nk]; Sg += pi[nk] * Mg[nk]; Sb += pi[nk] * Mb[nk];
        }
        Sr += __shfl_xor(Sr, 16); Sr += __shfl_xor(Sr, 32);
        Sg += __shfl_xor(Sg, 16); Sg += __shfl_xor(Sg, 32);
        Sb += __shfl_xor(Sb, 16); Sb += __shfl_xor(Sb, 32);
        // clamp(-1, 1) of the forward blocks the gradient outside the interval
        const float gr = (Sr >= -1.f && Sr <= 1.f) ? grow[p0 + j] : 0.f;
        const float gg = (Sg >= -1.f && Sg <= 1.f) ? grow[npix + p0 + j] : 0.f;
        const float gb = (Sb >= -1.f && Sb <= 1.f) ? grow[2 * npix + p0 + j] : 0.f;
        float dpi[3], dot = 0.f;
        for (int i = 0; i < nk; ++i) {
            dpi[i] = gr * Mr[i] + gg * Mg[i] + gb * Mb[i];
            dot += pi[i] * dpi[i];
        }
        dot += __shfl_xor(dot, 16);
        dot += __shfl_xor(dot, 32);
        __builtin_amdgcn_wave_barrier();           // every lane has read the logits of its pixel
        nk = 0;
        for (int k = q; k < NMIX; k += 4, ++nk) {
            float* m = pp + 8 * k;
            const float dMb = gb * pi[nk];
            const float dMg = gg * pi[nk] + dMb * cf[nk][2];
            const float dMr = gr * pi[nk] + dMb * cf[nk][1] + dMg * cf[nk][0];
            m[0] = pi[nk] * (dpi[nk] - dot);
            m[1] = dMr;
            m[2] = dMg;
            m[3] = dMb;
            m[4] = dMg * Mr[nk] * (1.f - cf[nk][0] * cf[nk][0]);
            m[5] = dMb * Mr[nk] * (1.f - cf[nk][1] * cf[nk][1]);
            m[6] = dMb * Mg[nk] * (1.f - cf[nk][2] * cf[nk][2]);
            m[7] = 0.f;                               // log_scale_r: the mean does not depend on the scales
        }
        if (q == 0)
            for (int s = 8 * NMIX; s < PITCH; ++s) pp[s] = 0.f;
        __builtin_amdgcn_wave_barrier();
        float4* dst = reinterpret_cast<float4*>(drow + (size_t)p0 * PITCH);
        if constexpr (USED == PITCH) {
            for (int i = lane; i < F4; i += 64) dst[i] = reinterpret_cast<float4*>(st)[i];
        } else {
            constexpr int U4 = USED / 4;
            for (int i = lane; i < 16 * U4; i += 64) {
                const int px = i / U4, c4 = i - px * U4;
                dst[px * (PITCH / 4) + c4] = reinterpret_cast<float4*>(st)[px * (PITCH / 4) + c4];
            }
        }
        if (colsum) {
#pragma unroll
            for (int px = 0; px < 16; ++px) {
                cs0 += st[px * PITCH + lane];
                if (lane + 64 < PITCH) cs1 += st[px * PITCH + lane + 64];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (colsum) {
        csum[wave][0][lane] = cs0;
        csum[wave][1][lane] = cs1;
        __syncthreads();
        for (int i = tid; i < PITCH; i += 256) {
            const int h = i >> 6, l = i & 63;
            colsum[(size_t)row * PITCH + i] = (csum[0][h][l] + csum[1][h][l]) + (csum[2][h][l] + csum[3][h][l]);
        }
    }
}

struct RadamCoef { float step, gsc; bool rect; };
__device__ __forceinline__ RadamCoef radam_coef(const float* __restrict__ state, const float beta1, const float beta2, const float grad_scale) {
    const float t = state[0] + 1.f;
    const float b2t = powf(beta2, t), b1t = powf(beta1, t);
    const float sma_max = 2.f / (1.f - beta2) - 1.f;
    const float sma = sma_max - 2.f * t * b2t / (1.f - b2t);
    RadamCoef c;
    c.rect = sma >= 5.f;
    if (c.rect) c.step = sqrtf((1.f - b2t) * (sma - 4.f) / (sma_max - 4.f) * (sma - 2.f) / sma * sma_max / (sma_max - 2.f)) / (1.f - b1t);
    else c.step = 1.f / (1.f - b1t);
    c.gsc = grad_scale * (state[1] > 0.f ? state[1] : 1.f);     // state[1]: this step's clipping coefficient (gcpx_grad_clip_coef), 0 = unset
    return c;
}
__device__ __forceinline__ void radam_one(float& th, const float gr, float& m, float& v, const RadamCoef& c, const float lr, const float beta1,
                                          const float beta2, const float eps) {
    // (no contraction: the 16-byte and the scalar loop must round alike — the compiler fused different products in the two)
#pragma clang fp contract(off)
    const float g = gr * c.gsc;
    const float mi = beta1 * m + (1.f - beta1) * g;
    const float vi = beta2 * v + (1.f - beta2) * g * g;
    m = mi;
    v = vi;
    th -= c.rect ? lr * c.step * mi / (sqrtf(vi) + eps) : lr * c.step * mi;
}

// VEC: the four vectors are 16-byte aligned — four elements per thread and access, two accesses in flight (a launch held to a few
// workgroups, gcpx_optim_range's max_blocks, still pulls ~10 GB/s per wavefront); the remainder and unaligned slices take the scalar loop.
// The arithmetic per element is the same function either way: a step cut into slices leaves the bits of one call.
template <bool VEC>
__global__ void __launch_bounds__(256) radam_kernel(float* __restrict__ theta, const float* __restrict__ grad,
                                                    float* __restrict__ m, float* __restrict__ v, const float* __restrict__ state,
                                                    const long long n, const float lr, const float beta1, const float beta2,
                                                    const float eps, const float grad_scale) {
    const RadamCoef c = radam_coef(state, beta1, beta2, grad_scale);
    const long long stride = (long long)gridDim.x * 256, first = (long long)blockIdx.x * 256 + threadIdx.x;
    long long done = 0;
    if (VEC) {
        const long long n4 = n >> 2;
        float4* __restrict__ t4 = reinterpret_cast<float4*>(theta);
        const float4* __restrict__ g4 = reinterpret_cast<const float4*>(grad);
        float4* __restrict__ m4 = reinterpret_cast<float4*>(m);
        float4* __restrict__ v4 = reinterpret_cast<float4*>(v);
        for (long long i = first; i < n4; i += 2 * stride) {
            const long long j = i + stride;
            const bool two = j < n4;
            float4 ta = t4[i], ga = g4[i], ma = m4[i], va = v4[i];
            float4 tb = ta, gb = ga, mb = ma, vb = va;
            if (two) { tb = t4[j]; gb = g4[j]; mb = m4[j]; vb = v4[j]; }
            radam_one(ta.x, ga.x, ma.x, va.x, c, lr, beta1, beta2, eps);
            radam_one(ta.y, ga.y, ma.y, va.y, c, lr, beta1, beta2, eps);
            radam_one(ta.z, ga.z, ma.z, va.z, c, lr, beta1, beta2, eps);
            radam_one(ta.w, ga.w, ma.w, va.w, c, lr, beta1, beta2, eps);
            t4[i] = ta; m4[i] = ma; v4[i] = va;
            if (two) {
                radam_one(tb.x, gb.x, mb.x, vb.x, c, lr, beta1, beta2, eps);
                radam_one(tb.y, gb.y, mb.y, vb.y, c, lr, beta1, beta2, eps);
                radam_one(tb.z, gb.z, mb.z, vb.z, c, lr, beta1, beta2, eps);
                radam_one(tb.w, gb.w, mb.w, vb.w, c, lr, beta1, beta2, eps);
                t4[j] = tb; m4[j] = mb; v4[j] = vb;
            }
        }
        done = n4 << 2;
    }
    for (long long i = done + first; i < n; i += stride) radam_one(theta[i], grad[i], m[i], v[i], c, lr, beta1, beta2, eps);
}

__global__ void radam_tick_kernel(float* state) { state[0] += 1.f; }

// The trainer's other optimizers (gcp_builder.py:174-186: 'adam', 'rmsprop', 'sgd' as torch.optim defines them) and the optional
// gradient clipping by global norm.  state[0] = step counter, state[1] = clip coefficient of this step (1 when clipping is off).
//   adam:    m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  theta -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//   rmsprop: v = alpha v + (1 - alpha) g^2;  d = g / (sqrt(v) + eps);  momentum > 0: m = momentum m + d, theta -= lr m;  else theta -= lr d
//   sgd:     momentum > 0: m = momentum m + g (m = g at the first step), theta -= lr m;  else theta -= lr g
__global__ void __launch_bounds__(256) optim_kernel(float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m,
                                                    float* __restrict__ v, const float* __restrict__ state, const long long n,
                                                    const int kind, const float lr, const float p1, const float p2, const float eps,
                                                    const float grad_scale) {
    const float t = state[0] + 1.f;
    const float gs = grad_scale * (state[1] > 0.f ? state[1] : 1.f);
    const float bc1 = 1.f - powf(p1, t), bc2s = sqrtf(1.f - powf(p2, t));
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float g = grad[i] * gs;
        if (kind == 1) {
            const float mi = p1 * m[i] + (1.f - p1) * g;
            const float vi = p2 * v[i] + (1.f - p2) * g * g;
            m[i] = mi; v[i] = vi;
            theta[i] -= lr / bc1 * mi / (sqrtf(vi) / bc2s + eps);
        } else if (kind == 2) {
            const float vi = p2 * v[i] + (1.f - p2) * g * g;
            v[i] = vi;
            const float d = g / (sqrtf(vi) + eps);
            if (p1 > 0.f) { const float mi = p1 * m[i] + d; m[i] = mi; theta[i] -= lr * mi; }
            else theta[i] -= lr * d;
        } else {
            if (p1 > 0.f) { const float mi = t == 1.f ? g : p1 * m[i] + g; m[i] = mi; theta[i] -= lr * mi; }
            else theta[i] -= lr * g;
        }
    }
}

// sum of squares of the gradient in a fixed order: [blocks] partials, then one workgroup
__global__ void __launch_bounds__(256) sqnorm_partial_kernel(const float* __restrict__ g, const long long n, float* __restrict__ part) {
    __shared__ float red[256];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += g[i] * g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// clip_grad_norm_: coefficient = min(1, max_norm / (||grad_scale * g|| + 1e-6)) -> state[1]; the norm itself -> state[2]
__global__ void __launch_bounds__(256) clip_coef_kernel(const float* __restrict__ part, const int nb, const float grad_scale,
                                                        const float max_norm, float* __restrict__ state) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = fabsf(grad_scale) * sqrtf(red[0]);
        state[2] = norm;
        state[1] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
    }
}

int blocks_for(long long items, int cap = 4096) {
    long long b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

#define STREAM() hipStream_t stream = reinterpret_cast<hipStream_t>(stream_)

extern "C" int gcpx_dlm_nll_bwd(const float* params, const float* target, const float* row_weight, float scale, float* dparams,
                                float* colsum, float* nll_out, int32_t rows, int32_t npix, int32_t pitch, int32_t n_mix, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(params && target && dparams && rows > 0, "bad arguments");
    GCPX_CHECK_ARG(n_mix == 10 && pitch == 112 && npix % 64 == 0, "supports 10 mixtures, pitch 112, npix % 64 == 0");
    hipLaunchKernelGGL((dlm_nll_bwd_kernel<10, 112>), dim3(rows), dim3(256), 0, stream, params, target, row_weight, scale, dparams, colsum,
                       nll_out, npix);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_dlm_mean_bwd(const float* params, const float* dimg, float* dparams, float* colsum, int32_t rows, int32_t npix,
                                 int32_t pitch, int32_t n_mix, int32_t used_slots, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(params && dimg && dparams && rows > 0, "bad arguments");
    GCPX_CHECK_ARG(n_mix == 10 && pitch == 112 && npix % 64 == 0, "supports 10 mixtures, pitch 112, npix % 64 == 0");
    GCPX_CHECK_ARG(used_slots == pitch || used_slots == 8 * n_mix, "used_slots: the pitch (every slot written) or 8 * n_mix (the slots the mean reads)");
    if (used_slots == pitch)
        hipLaunchKernelGGL((dlm_mean_bwd_kernel<10, 112, 112>), dim3(rows), dim3(256), 0, stream, params, dimg, dparams, colsum, npix);
    else
        hipLaunchKernelGGL((dlm_mean_bwd_kernel<10, 112, 80>), dim3(rows), dim3(256), 0, stream, params, dimg, dparams, colsum, npix);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_grad_clip_coef(const float* grad, int64_t n, float grad_scale, float max_norm, float* partial, int32_t n_partial,
                                   float* state, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(grad && partial && state && n > 0 && n_partial > 0 && n_partial <= 4096, "bad arguments");
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(n_partial), dim3(256), 0, stream, grad, (long long)n, partial);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, stream, partial, n_partial, grad_scale, max_norm, state);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_optim_step(float* theta, const float* grad, float* m, float* v, float* state, int64_t n, int32_t kind, float lr,
                               float p1, float p2, float eps, float grad_scale, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(theta && grad && m && v && state && n > 0 && kind >= 1 && kind <= 3, "bad arguments");
    hipLaunchKernelGGL(optim_kernel, dim3(blocks_for(n, 16384)), dim3(256), 0, stream, theta, grad, m, v, state, (long long)n, kind, lr,
                       p1, p2, eps, grad_scale);
    hipLaunchKernelGGL(radam_tick_kernel, dim3(1), dim3(1), 0, stream, state);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

namespace {
void launch_radam(float* theta, const float* grad, float* m, float* v, float* state, int64_t n, float lr, float b1, float b2, float eps,
                  float grad_scale, int max_blocks, hipStream_t stream) {
    const bool vec = ((((uintptr_t)theta | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && n >= 4;
    int nb = blocks_for(vec ? (n + 7) / 8 : n, 16384);
    if (max_blocks > 0) nb = std::min(nb, max_blocks);
    if (vec)
        hipLaunchKernelGGL(radam_kernel<true>, dim3(nb), dim3(256), 0, stream, theta, grad, m, v, state, (long long)n, lr, b1, b2, eps, grad_scale);
    else
        hipLaunchKernelGGL(radam_kernel<false>, dim3(nb), dim3(256), 0, stream, theta, grad, m, v, state, (long long)n, lr, b1, b2, eps, grad_scale);
}
}  // namespace

extern "C" int gcpx_optim_range(float* theta, const float* grad, float* m, float* v, float* state, int64_t n, int32_t kind, float lr,
                                float p1, float p2, float eps, float grad_scale, int32_t tick, int32_t max_blocks, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(theta && grad && m && v && state && n > 0 && kind >= 0 && kind <= 3, "bad arguments");
    GCPX_CHECK_ARG(max_blocks >= 0, "max_blocks: 0 (no limit) or a positive number of workgroups");
    if (kind == 0)
        launch_radam(theta, grad, m, v, state, n, lr, p1, p2, eps, grad_scale, max_blocks, stream);
    else
        hipLaunchKernelGGL(optim_kernel, dim3(max_blocks > 0 ? std::min(max_blocks, blocks_for(n, 16384)) : blocks_for(n, 16384)), dim3(256), 0,
                           stream, theta, grad, m, v, state, (long long)n, kind, lr, p1, p2, eps, grad_scale);
    if (tick) hipLaunchKernelGGL(radam_tick_kernel, dim3(1), dim3(1), 0, stream, state);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_radam_step(float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, float* state, int64_t n, float lr,
                               float beta1, float beta2, float eps, float grad_scale, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(theta && grad && exp_avg && exp_avg_sq && state && n > 0, "bad arguments");
    launch_radam(theta, grad, exp_avg, exp_avg_sq, state, n, lr, beta1, beta2, eps, grad_scale, 0, stream);
    hipLaunchKernelGGL(radam_tick_kernel, dim3(1), dim3(1), 0, stream, state);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
