// Adaptive (soft-DTW) frame binding and attentive inference of the long-horizon configuration (c5):
//   image cost matrix (batch_cdist)  -> f32 MFMA "NT" GEMM with split-K partials + norm epilogue
//   soft-DTW forward / backward sweep -> float64, one workgroup per (sequence, direction), row-parallel
//   expected edge frequencies, column normalisation, argmax / entropy bookkeeping, averaging loss
//   masked single-query multi-head attention over the encoded sequence
// Reference: gcp/prediction/models/adaptive_binding/{adaptive,probabilistic_dtw,binding_loss,attentive_inference}.py
#include "common.h"

#include <math.h>

namespace {


__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------------
// attention: one 256-thread workgroup per query row r = (b, j); keys / values of sequence b = r / rpb.
// Scores: one key per thread (the dh-long dot as float4 loads); softmax over the T keys by workgroup reductions; values:
// thread (c4, tg) sums 4 channels over the keys t = tg, tg + G, ... and the G partial sums are combined in a fixed order.
// (The first version gave a row to one wavefront: 200 dependent loads per lane in the value loop, 31 us whatever the level.)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce(float v, float* red, const bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const int64_t* __restrict__ start_ind,
                                                        const int64_t* __restrict__ end_ind,
                                                        const float* __restrict__ temperature, float* __restrict__ out,
                                                        float* __restrict__ att, const int M, const int rpb, const int T,
                                                        const int dk, const int nz, const int heads) {
    extern __shared__ float smem[];
    float* p = smem;                                      // [T] scores -> probabilities
    float* part = smem + ((T + 3) & ~3);                  // [256 * 4] partial value sums (16-byte aligned)
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int r = blockIdx.x;
    const int b = r / rpb;
    const int s = start_ind ? (int)start_ind[b] : 0, e = (int)end_ind[b];
    const float* kb = k + (size_t)b * T * dk;
    const float* vb = v + (size_t)b * T * nz;
    const int dh = dk / heads, vh = nz / heads;
    const float inv = 1.f / sqrtf((float)dh) / temperature[0];
    const int c4n = vh / 4;                               // float4 channel groups of a head
    const int G = 256 / c4n;                              // key groups in the value pass
    for (int h = 0; h < heads; ++h) {
        const float* qr = q + (size_t)r * dk + h * dh;
        float mx = -INFINITY;
        for (int t = tid; t < T; t += 256) {
            const float* kr = kb + (size_t)t * dk + h * dh;
            float d = 0.f;
            for (int i = 0; i < dh; i += 4) {
                const float4 a4 = *reinterpret_cast<const float4*>(qr + i), b4 = *reinterpret_cast<const float4*>(kr + i);
                d = fmaf(a4.x, b4.x, d); d = fmaf(a4.y, b4.y, d); d = fmaf(a4.z, b4.z, d); d = fmaf(a4.w, b4.w, d);
            }
            float sc = d * inv;
            if (t < s || t > e) sc = -INFINITY;
            p[t] = sc;
            mx = fmaxf(mx, sc);
        }
        mx = block_reduce(mx, red, true);
        float sum = 0.f;
        for (int t = tid; t < T; t += 256) {
            const float ex = (p[t] == -INFINITY) ? 0.f : expf(p[t] - mx);
            p[t] = ex;
            sum += ex;
        }
        sum = block_reduce(sum, red, false);
        for (int t = tid; t < T; t += 256) {
            const float a = p[t] / sum;
            p[t] = a;
            if (att) att[(size_t)r * T + t] = (h == 0 ? 0.f : att[(size_t)r * T + t]) + a / (float)heads;
        }
        __syncthreads();
        const int c4 = tid % c4n, tg = tid / c4n;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tg < G) {
            const int t0 = s < 0 ? 0 : s, te = e >= T ? T - 1 : e;
            const float* vc = vb + h * vh + c4 * 4;
#pragma unroll 4
            for (int t = t0 + tg; t <= te; t += G) {
                const float4 x = *reinterpret_cast<const float4*>(vc + (size_t)t * nz);
                const float w = p[t];
                acc.x = fmaf(w, x.x, acc.x); acc.y = fmaf(w, x.y, acc.y); acc.z = fmaf(w, x.z, acc.z); acc.w = fmaf(w, x.w, acc.w);
            }
        }
        reinterpret_cast<float4*>(part)[tid] = acc;
        __syncthreads();
        if (tid < c4n) {
            float4 o = reinterpret_cast<float4*>(part)[tid];
            for (int g = 1; g < G; ++g) {
                const float4 x = reinterpret_cast<float4*>(part)[g * c4n + tid];
                o.x += x.x; o.y += x.y; o.z += x.z; o.w += x.w;
            }
            *reinterpret_cast<float4*>(out + (size_t)r * nz + h * vh + tid * 4) = o;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// batch_cdist: P[s][b][n][t] = sum over K-chunk s of X[b][n][k] * Y[b][t][k]   (f32 MFMA, frames on the i side)
// workgroup = 4 wavefronts, tile 64 nodes x 64 frames, K staged through LDS 32 at a time
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cdist_gemm_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                         float* __restrict__ P, const int B, const int N, const int T,
                                                         const int K, const int kchunk, const int tiles_t) {
    __shared__ float sX[64][36];
    __shared__ float sY[64][36];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = (blockIdx.x / tiles_t) * 64, t0 = (blockIdx.x % tiles_t) * 64;
    const int ks = blockIdx.y, b = blockIdx.z;
    const float* Xb = X + (size_t)b * N * K;
    const float* Yb = Y + (size_t)b * T * K;
    const int nsub = (wave >> 1) * 32, tsub = (wave & 1) * 32;
    const int li = lane & 15, kk = lane >> 4;
    f32x4 acc[2][2];
    for (int a = 0; a < 2; ++a)
        for (int c = 0; c < 2; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int kbeg = ks * kchunk;
    for (int k0 = kbeg; k0 < kbeg + kchunk; k0 += 32) {
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, c4 = idx & 7;
            float4 vx = make_float4(0.f, 0.f, 0.f, 0.f), vy = vx;
            if (n0 + row < N) vx = *reinterpret_cast<const float4*>(Xb + (size_t)(n0 + row) * K + k0 + 4 * c4);
            if (t0 + row < T) vy = *reinterpret_cast<const float4*>(Yb + (size_t)(t0 + row) * K + k0 + 4 * c4);
            *reinterpret_cast<float4*>(&sX[row][4 * c4]) = vx;
            *reinterpret_cast<float4*>(&sY[row][4 * c4]) = vy;
        }
        __syncthreads();
        for (int kg = 0; kg < 2; ++kg) {
            float4 ft[2], fn[2];
            for (int a = 0; a < 2; ++a) {
                ft[a] = *reinterpret_cast<const float4*>(&sY[tsub + 16 * a + li][16 * kg + 4 * kk]);
                fn[a] = *reinterpret_cast<const float4*>(&sX[nsub + 16 * a + li][16 * kg + 4 * kk]);
            }
            for (int a = 0; a < 2; ++a)
                for (int c = 0; c < 2; ++c) {
                    acc[a][c] = mfma16(ft[a].x, fn[c].x, acc[a][c]);
                    acc[a][c] = mfma16(ft[a].y, fn[c].y, acc[a][c]);
                    acc[a][c] = mfma16(ft[a].z, fn[c].z, acc[a][c]);
                    acc[a][c] = mfma16(ft[a].w, fn[c].w, acc[a][c]);
                }
        }
        __syncthreads();
    }
    float* Pb = P + ((size_t)ks * B + b) * N * T;
    for (int a = 0; a < 2; ++a)
        for (int c = 0; c < 2; ++c) {
            const int n = n0 + nsub + 16 * c + li;
            if (n >= N) continue;
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + tsub + 16 * a + 4 * kk + r;
                if (t < T) Pb[(size_t)n * T + t] = acc[a][c][r];
            }
        }
}

__global__ void __launch_bounds__(256) row_sumsq_kernel(const float* __restrict__ x, const long long K4, float* __restrict__ out) {
    __shared__ float red[256];
    const float4* xr = reinterpret_cast<const float4*>(x) + (size_t)blockIdx.x * K4;
    float s = 0.f;
    for (long long i = threadIdx.x; i < K4; i += 256) {
        const float4 v = xr[i];
        s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256) cdist_finish_kernel(const float* __restrict__ P, const int nsplit,
                                                           const float* __restrict__ xn, const float* __restrict__ yn,
                                                           float* __restrict__ out, const int B, const int N, const int T) {
    const size_t total = (size_t)B * N * T;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int t = (int)(i % T);
        const size_t bn = i / T;
        const int b = (int)(bn / N);
        float dot = 0.f;
        for (int s = 0; s < nsplit; ++s) dot += P[(size_t)s * total + i];
        const float d = xn[bn] + yn[(size_t)b * T + t] - 2.f * dot;
        out[i] = d > 0.f ? d : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------
// soft-DTW sweeps (probabilistic_dtw.py:11-73 on the stacked forward / flipped problems, :101-106).
// 'nohor' transitions only come from the previous row, so a row is computed in parallel: thread j keeps D[i-1][j] in a
// register and reads D[i-1][j-1] from LDS.  Same recurrence as the reference's anti-diagonal sweep.
// ---------------------------------------------------------------------------------------------------
// softplus(x) = log(1 + e^x) for x <= 0 in float64, ~1 ulp, about 50 instructions instead of the ~350 of libm's exp + log1p
// (this function is the dependent chain of the soft-DTW sweep: 255 row steps per launch).
//   e^x: x = k ln2 + r, |r| <= ln2 / 2, Taylor to r^13 (next term < 4e-18), scaled by 2^k;
//   log1p(y), 0 < y <= 1: 2 atanh(s), s = y / (2 + y) <= 1/3, odd series to s^33 (next term < 6e-19).
__device__ __forceinline__ double softplus_neg(const double x) {
    if (x < -745.0) return 0.0;
    const double kf = rint(x * 1.4426950408889634);
    const double r = fma(-kf, 1.9082149292705877e-10, fma(-kf, 6.93147180369123816490e-01, x));      // ln2 = hi + lo
    double e = 1.0 / 6227020800.0;
    e = fma(e, r, 1.0 / 479001600.0); e = fma(e, r, 1.0 / 39916800.0); e = fma(e, r, 1.0 / 3628800.0);
    e = fma(e, r, 1.0 / 362880.0); e = fma(e, r, 1.0 / 40320.0); e = fma(e, r, 1.0 / 5040.0);
    e = fma(e, r, 1.0 / 720.0); e = fma(e, r, 1.0 / 120.0); e = fma(e, r, 1.0 / 24.0);
    e = fma(e, r, 1.0 / 6.0); e = fma(e, r, 0.5); e = fma(e, r, 1.0); e = fma(e, r, 1.0);
    const double y = ldexp(e, (int)kf);
    const double s = y / (2.0 + y), s2 = s * s;
    double p = 1.0 / 33.0;
    p = fma(p, s2, 1.0 / 31.0); p = fma(p, s2, 1.0 / 29.0); p = fma(p, s2, 1.0 / 27.0); p = fma(p, s2, 1.0 / 25.0);
    p = fma(p, s2, 1.0 / 23.0); p = fma(p, s2, 1.0 / 21.0); p = fma(p, s2, 1.0 / 19.0); p = fma(p, s2, 1.0 / 17.0);
    p = fma(p, s2, 1.0 / 15.0); p = fma(p, s2, 1.0 / 13.0); p = fma(p, s2, 1.0 / 11.0); p = fma(p, s2, 1.0 / 9.0);
    p = fma(p, s2, 1.0 / 7.0); p = fma(p, s2, 1.0 / 5.0); p = fma(p, s2, 1.0 / 3.0); p = fma(p, s2, 1.0);
    return 2.0 * s * p;
}

// log(e^a + e^b) = max + softplus(min - max)
__device__ __forceinline__ double lse2(const double a, const double b) {
    const double m = fmax(a, b), n = fmin(a, b);
    if (isinf(m)) return m;                                 // both -inf (or one +inf): the sum is m
    return m + softplus_neg(n - m);
}

__device__ __forceinline__ double neg_cost(const float dsum, const float D, const float temp) {
    const float mean = dsum / D;
    return -(double)(mean / temp);
}

__global__ void dtw_sweep_kernel(const float* __restrict__ dsum, const float D, const float* __restrict__ temp_p,
                                 const int64_t* __restrict__ end_ind, double* __restrict__ acc, const int B, const int r,
                                 const int c) {
    extern __shared__ double sh[];
    const int dir = blockIdx.x / B, b = blockIdx.x % B;
    const int j = threadIdx.x;
    const int cp = blockDim.x;
    const int end = (int)end_ind[b];
    const int begin = dir ? c - end - 1 : 0;
    const float temp = temp_p[0];
    const float* Cb = dsum + (size_t)b * r * c;
    double* Ab = acc + (size_t)blockIdx.x * r * c;
    const int sj = dir ? c - 1 - j : j;
    double prev = -INFINITY;
    // the cost of row i + 1 is requested before row i is computed: the row step is a dependent chain (LDS exchange + barrier),
    // the global load must not sit on it
    float cnext = (j < c) ? Cb[(size_t)(dir ? r - 1 : 0) * c + sj] : 0.f;
    for (int i = 0; i < r; ++i) {
        const int si = dir ? r - 1 - i : i;
        double cur = -INFINITY;
        const float craw = cnext;
        if (j < c && i + 1 < r) cnext = Cb[(size_t)(dir ? r - 2 - i : i + 1) * c + sj];
        if (j < c) {
            const double cij = neg_cost(craw, D, temp);
            if (i == 0) {
                cur = (j == begin) ? cij : -INFINITY;
            } else {
                const double left = j > 0 ? sh[((i - 1) & 1) * cp + j - 1] : -INFINITY;
                cur = cij + lse2(prev, left);
            }
            sh[(i & 1) * cp + j] = cur;
            Ab[(size_t)si * c + sj] = cur;
        }
        prev = cur;
        __syncthreads();
    }
}

// expected edge frequencies w = exp(fwd + bwd - C - z) (probabilistic_dtw.py:108-114), then normalised over the node axis
// (adaptive.py:58).  One workgroup per sequence, threads over frames.
__global__ void dtw_combine_kernel(const float* __restrict__ dsum, const float D, const float* __restrict__ temp_p,
                                   const int64_t* __restrict__ end_ind, const double* __restrict__ acc, float* __restrict__ w,
                                   const int B, const int r, const int c) {
    // blockDim = (cp, G): thread (t, g) takes the nodes n = g, g + G, ... of frame t; the G partial column sums are combined in a
    // fixed order
    extern __shared__ float csum[];                     // [G][cp]
    const int b = blockIdx.x, t = threadIdx.x, g = threadIdx.y, cp = blockDim.x, G = blockDim.y;
    const float temp = temp_p[0];
    const double* F = acc + (size_t)b * r * c;
    const double* Bw = acc + (size_t)(B + b) * r * c;
    const float* Cb = dsum + (size_t)b * r * c;
    float* wb = w + (size_t)b * r * c;
    const double z = F[(size_t)(r - 1) * c + (int)end_ind[b]];
    float colsum = 0.f;
    if (t < c) {
        for (int n = g; n < r; n += G) {
            const size_t o = (size_t)n * c + t;
            const double e = F[o] + Bw[o] - neg_cost(Cb[o], D, temp);
            const float wv = (float)exp(e - z);
            wb[o] = wv;
            colsum += wv;
        }
    }
    csum[g * cp + t] = colsum;
    __syncthreads();
    if (t >= c) return;
    float tot = 0.f;
    for (int k = 0; k < G; ++k) tot += csum[k * cp + t];
    const float den = fmaxf(tot, 1e-7f);
    for (int n = g; n < r; n += G) wb[(size_t)n * c + t] /= den;
}

// ---------------------------------------------------------------------------------------------------
// learn_matching_temp (hyperparameters.py:132, adaptive.py:19-21): `soft_dtw(cost.detach() / self.temp, ...)` (adaptive.py:51) keeps
// the division by the temperature in the autograd graph, so the averaging criterion's loss (binding_loss.py:31-35) reaches `temp`
// through the matching weights.  d w / d temp in FORWARD mode along the same lattice: with C = -cost / temp, Cdot = cost / temp^2,
//   Ddot[0][begin] = Cdot;  Ddot[i][j] = Cdot[i][j] + s Ddot[i-1][j] + (1 - s) Ddot[i-1][j-1],  s = e^{D[i-1][j]} / (e^{D[i-1][j]} + e^{D[i-1][j-1]})
// (the derivative of the logsumexp of the sweep above), for the forward and the flipped problem; the values D are the forward pass's
// `acc`, so this sweep's dependent chain is two fmas, an LDS exchange and a barrier per row.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double neg_cost_dtemp(const float dsum, const float D, const float temp) {
    return (double)(dsum / D) / ((double)temp * (double)temp);
}

__global__ void dtw_tangent_sweep_kernel(const float* __restrict__ dsum, const float D, const float* __restrict__ temp_p,
                                         const int64_t* __restrict__ end_ind, const double* __restrict__ acc,
                                         double* __restrict__ tan, const int B, const int r, const int c) {
    extern __shared__ double sh[];                      // [2][cp] values, [2][cp] tangents of the previous rows
    const int dir = blockIdx.x / B, b = blockIdx.x % B;
    const int j = threadIdx.x;
    const int cp = blockDim.x;
    double* shv = sh;
    double* sht = sh + 2 * cp;
    const int end = (int)end_ind[b];
    const int begin = dir ? c - end - 1 : 0;
    const float temp = temp_p[0];
    const float* Cb = dsum + (size_t)b * r * c;
    const double* Ab = acc + (size_t)blockIdx.x * r * c;
    double* Tb = tan + (size_t)blockIdx.x * r * c;
    const int sj = dir ? c - 1 - j : j;
    double vprev = -INFINITY, tprev = 0.0;
    // row i + 1's value and cost are requested before row i is combined (neither is on the dependent chain)
    double vnext = -INFINITY;
    float cnext = 0.f;
    if (j < c) {
        const size_t o = (size_t)(dir ? r - 1 : 0) * c + sj;
        vnext = Ab[o];
        cnext = Cb[o];
    }
    for (int i = 0; i < r; ++i) {
        const int si = dir ? r - 1 - i : i;
        const double vcur = vnext;
        const float craw = cnext;
        if (j < c && i + 1 < r) {
            const size_t o = (size_t)(dir ? r - 2 - i : i + 1) * c + sj;
            vnext = Ab[o];
            cnext = Cb[o];
        }
        double tcur = 0.0;
        if (j < c) {
            const double cdot = neg_cost_dtemp(craw, D, temp);
            if (i == 0) {
                tcur = (j == begin) ? cdot : 0.0;
            } else if (!isinf(vcur)) {
                const double vleft = j > 0 ? shv[((i - 1) & 1) * cp + j - 1] : -INFINITY;
                const double tleft = j > 0 ? sht[((i - 1) & 1) * cp + j - 1] : 0.0;
                // share of the same-column predecessor in e^{D[i-1][j]} + e^{D[i-1][j-1]}
                const double s = isinf(vprev) ? 0.0 : (isinf(vleft) ? 1.0 : 1.0 / (1.0 + exp(vleft - vprev)));
                tcur = cdot + fma(s, tprev, (1.0 - s) * tleft);
            }
            shv[(i & 1) * cp + j] = vcur;
            sht[(i & 1) * cp + j] = tcur;
            Tb[(size_t)si * c + sj] = tcur;
        }
        vprev = vcur;
        tprev = tcur;
        __syncthreads();
    }
}

// d loss / d temp of the averaging criterion through the normalised matching weights: what = w / max(sum_n w, 1e-7) (adaptive.py:58),
// loss = coef * sum_{b,n,t} what * pad * (0.5 d exp(-2 ls) + D (ls + 0.5 log 2 pi)) (binding_loss.py:24-35).  One workgroup per
// sequence, thread (t, g) takes the nodes g, g + G, ... of frame t; all sums in float64 in a fixed order.
__global__ void dtw_dtemp_combine_kernel(const float* __restrict__ dsum, const float D, const float* __restrict__ temp_p,
                                         const int64_t* __restrict__ end_ind, const double* __restrict__ acc,
                                         const double* __restrict__ tan, const float* __restrict__ pad,
                                         const float* __restrict__ log_sigma, double* __restrict__ partial, const int B, const int r,
                                         const int c) {
    extern __shared__ double red[];                     // [4][G][cp]
    const int b = blockIdx.x, t = threadIdx.x, g = threadIdx.y, cp = blockDim.x, G = blockDim.y;
    const float temp = temp_p[0];
    const size_t rc = (size_t)r * c;
    const double* F = acc + (size_t)b * rc;
    const double* Bw = acc + (size_t)(B + b) * rc;
    const double* TF = tan + (size_t)b * rc;
    const double* TB = tan + (size_t)(B + b) * rc;
    const float* Cb = dsum + (size_t)b * rc;
    const size_t zo = (size_t)(r - 1) * c + (int)end_ind[b];
    const double z = F[zo], zdot = TF[zo];
    const float ls = log_sigma[0];
    const double iv2 = exp(-2.0 * (double)ls), c0 = (double)D * ((double)ls + 0.91893853320467274178);
    double s = 0.0, sd = 0.0, a = 0.0, q = 0.0;
    if (t < c) {
        for (int n = g; n < r; n += G) {
            const size_t o = (size_t)n * c + t;
            const float craw = Cb[o];
            const double e = F[o] + Bw[o] - neg_cost(craw, D, temp);
            if (isinf(e) || isnan(e)) continue;         // unreachable cell: w = 0 for every temp
            const double wv = (double)(float)exp(e - z);
            const double wd = wv * (TF[o] + TB[o] - neg_cost_dtemp(craw, D, temp) - zdot);
            const double gl = fma(0.5 * (double)craw, iv2, c0);
            s += wv;
            sd += wd;
            a = fma(gl, wd, a);
            q = fma(gl, wv, q);
        }
    }
    const int slot = g * cp + t, plane = G * cp;
    red[slot] = s; red[plane + slot] = sd; red[2 * plane + slot] = a; red[3 * plane + slot] = q;
    __syncthreads();
    double contrib = 0.0;
    if (g == 0 && t < c) {
        double S = 0.0, SD = 0.0, A = 0.0, Q = 0.0;
        for (int k = 0; k < G; ++k) {
            S += red[k * cp + t]; SD += red[plane + k * cp + t]; A += red[2 * plane + k * cp + t]; Q += red[3 * plane + k * cp + t];
        }
        // below the clamp the denominator is the constant 1e-7 (blox normalize as the forward applies it)
        contrib = (double)pad[(size_t)b * c + t] * (S > 1e-7 ? (A - Q * SD / S) / S : A / 1e-7);
    }
    __syncthreads();
    if (g == 0) red[t] = contrib;
    __syncthreads();
    if (g == 0 && t == 0) {
        double tot = 0.0;
        for (int k = 0; k < c; ++k) tot += red[k];
        partial[b] = tot;
    }
}

__global__ void dtw_dtemp_finish_kernel(const double* __restrict__ partial, const int B, const float coef, float* __restrict__ dst) {
    double tot = 0.0;
    for (int b = 0; b < B; ++b) tot += partial[b];
    dst[0] += (float)((double)coef * tot);
}

// breadth-first index q -> depth-first position
__device__ __forceinline__ int bf2df(const int q, const int L) {
    const int l = 31 - __clz(q + 1);
    const int jj = q + 1 - (1 << l);
    return (2 * jj + 1) * (1 << (L - 1 - l)) - 1;
}

// The bookkeeping kernels below give every output row to ONE WAVEFRONT whose lanes stride over the reduced axis (the first
// versions looped serially per thread: 200 - 255 dependent loads each, 55 - 90 us per launch on an otherwise idle GPU at the tail of
// the adaptive forward).  Reductions are butterflies in a fixed order; arg-max ties resolve to the smaller index like the serial loop.

// per frame: node with the largest matching probability, first maximum in breadth-first order (frame_binding.py:30)
__global__ void __launch_bounds__(256) match_argmax_kernel(const float* __restrict__ w, const int64_t* __restrict__ end_ind,
                                                           int32_t* __restrict__ frame2node, int32_t* __restrict__ matched_idx,
                                                           const int B, const int L, const int T) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B * T) return;
    const int b = i / T, t = i % T;
    const int N = (1 << L) - 1;
    const float* wb = w + (size_t)b * N * T + t;
    float best = -INFINITY;
    int bq = 0x7fffffff;
    for (int q = lane; q < N; q += 64) {                  // increasing q inside a lane: strict > keeps the first maximum
        const float v = wb[(size_t)bf2df(q, L) * T];
        if (v > best) { best = v; bq = q; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oq = __shfl_xor(bq, o);
        if (ov > best || (ov == best && oq < bq)) { best = ov; bq = oq; }
    }
    if (lane == 0) {
        const int bp = (bq == 0x7fffffff) ? 0 : bf2df(bq, L);
        frame2node[i] = bp;
        if (matched_idx) matched_idx[i] = t <= (int)end_ind[b] ? bp : -1;
    }
}

// per node: best frame (first maximum), entropy of its matching distribution, existence probability (tree_module.py:145-147)
__global__ void __launch_bounds__(256) match_node_stats_kernel(const float* __restrict__ w, int32_t* __restrict__ best_t,
                                                               float* __restrict__ entropy, float* __restrict__ p_n,
                                                               const int BN, const int T) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= BN) return;
    const float* wr = w + (size_t)i * T;
    float best = -INFINITY, ent = 0.f, s = 0.f;
    int bt = 0x7fffffff;
    for (int t = lane; t < T; t += 64) {
        const float v = wr[t];
        if (v > best) { best = v; bt = t; }
        if (v > 0.f) ent -= v * logf(v);
        s += v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int ot = __shfl_xor(bt, o);
        if (ov > best || (ov == best && ot < bt)) { best = ov; bt = ot; }
        ent += __shfl_xor(ent, o);
        s += __shfl_xor(s, o);
    }
    if (lane == 0) {
        best_t[i] = (bt == 0x7fffffff) ? 0 : bt;
        entropy[i] = ent;
        p_n[i] = fminf(fmaxf(s, 0.f), 1.f);
    }
}

// learned pruning (adaptive.py:62-77): keep node p unless sigmoid(distance[p-1]) > threshold; compaction of kept positions;
// BCE target 1 where consecutive nodes share their best frame (adaptive.py:118-122).  One wavefront per sequence: 64 nodes per
// trip, compaction by ballot + popcount (order preserved).
__global__ void __launch_bounds__(64) distance_prune_kernel(const float* __restrict__ dist, const float thr, const int32_t* __restrict__ best_t,
                                                            int32_t* __restrict__ leave, int32_t* __restrict__ kept_idx,
                                                            int32_t* __restrict__ count, int32_t* __restrict__ target, const int B,
                                                            const int N) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    int k = 0;
    for (int p0 = 0; p0 < N; p0 += 64) {
        const int p = p0 + lane;
        int keep = 0;
        if (p < N) {
            keep = 1;
            if (p > 0) {
                const float x = dist[(size_t)b * (N - 1) + p - 1];
                keep = !(1.f / (1.f + expf(-x)) > thr);
                if (target) target[(size_t)b * (N - 1) + p - 1] = best_t[(size_t)b * N + p] == best_t[(size_t)b * N + p - 1];
            }
            leave[(size_t)b * N + p] = keep;
        }
        const unsigned long long m = __ballot(keep);
        if (keep) kept_idx[(size_t)b * N + k + __popcll(m & ((1ull << lane) - 1ull))] = p;
        k += __popcll(m);
    }
    if (lane == 0) count[b] = k;
    for (int q = k + lane; q < N; q += 64) kept_idx[(size_t)b * N + q] = -1;
}

// LossAveragingCriterion (binding_loss.py:19-42): nll_bt[b][t] = sum_n w[b][n][t] * (0.5 * d * exp(-ls)^2 + D * (ls + 0.5 log 2pi))
__global__ void __launch_bounds__(256) averaging_nll_kernel(const float* __restrict__ dsum, const float* __restrict__ w,
                                                            const float* __restrict__ log_sigma, const float D,
                                                            float* __restrict__ nll_bt, const int B, const int N, const int T) {
    // workgroup = 64 consecutive frames of one sequence x 4 node groups: coalesced rows, 4 partial sums combined in a fixed order
    __shared__ float part[4][64];
    const int tl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int tb = (T + 63) / 64;
    const int b = blockIdx.x / tb, t = (blockIdx.x % tb) * 64 + tl;
    const float ls = log_sigma[0];
    const float iv = expf(-ls), c0 = D * (ls + 0.91893853320467274178f);
    float s = 0.f;
    if (t < T) {
        const size_t base = (size_t)b * N * T + t;
        for (int n = g; n < N; n += 4) {
            const size_t o = base + (size_t)n * T;
            s += (0.5f * dsum[o] * (iv * iv) + c0) * w[o];
        }
    }
    part[g][tl] = s;
    __syncthreads();
    if (g == 0 && t < T) nll_bt[(size_t)b * T + t] = (part[0][tl] + part[1][tl]) + (part[2][tl] + part[3][tl]);
}

// ---------------------------------------------------------------------------------------------------
// hard DTW of the evaluation harness (dtw_utils.py:77-95 basic_dtw + :201-218 _traceback, as used by
// DTWEvalBinding.get_single_matches, evaluation_matching.py:133-146).  One workgroup per sequence; anti-diagonal
// wavefront in float64 over the float32 cost matrix; traceback and per-frame best node by one thread.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dtw_align_kernel(const float* __restrict__ cost, const int32_t* __restrict__ n_len,
                                                        const int32_t* __restrict__ t_len, double* __restrict__ acc,
                                                        int32_t* __restrict__ inds, int32_t* __restrict__ path,
                                                        int32_t* __restrict__ path_len, double* __restrict__ dist,
                                                        const int N, const int T) {
    const int b = blockIdx.x;
    const int n = n_len ? n_len[b] : N, t = t_len ? t_len[b] : T;
    const float* C = cost + (size_t)b * N * T;
    double* D = acc + (size_t)b * N * T;
    for (int dg = 0; dg < n + t - 1; ++dg) {
        const int ilo = dg - (t - 1) > 0 ? dg - (t - 1) : 0, ihi = dg < n - 1 ? dg : n - 1;
        for (int i = ilo + threadIdx.x; i <= ihi; i += 256) {
            const int j = dg - i;
            double m;
            if (i == 0 && j == 0) m = 0.0;
            else {
                const double a0 = (i > 0 && j > 0) ? D[(size_t)(i - 1) * T + j - 1] : INFINITY;
                const double a1 = i > 0 ? D[(size_t)(i - 1) * T + j] : INFINITY;
                const double a2 = j > 0 ? D[(size_t)i * T + j - 1] : INFINITY;
                m = fmin(a0, fmin(a1, a2));
            }
            D[(size_t)i * T + j] = (double)C[(size_t)i * T + j] + m;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    int32_t* pb = path + (size_t)b * 2 * (N + T);
    int i = n - 1, j = t - 1, len = 0;
    for (int jj = 0; jj < T; ++jj) inds[(size_t)b * T + jj] = -1;
    double best = INFINITY;
    int cur_col = j;
    while (true) {
        pb[len] = i; pb[(N + T) + len] = j; ++len;                 // stored end -> start; the host view reverses it
        if (j != cur_col) { cur_col = j; best = INFINITY; }
        const double v = D[(size_t)i * T + j];
        if (v <= best) { best = v; inds[(size_t)b * T + j] = i; }   // argmin over the column's path cells, first minimum
        if (i == 0 && j == 0) break;
        const double a0 = (i > 0 && j > 0) ? D[(size_t)(i - 1) * T + j - 1] : INFINITY;
        const double a1 = i > 0 ? D[(size_t)(i - 1) * T + j] : INFINITY;
        const double a2 = j > 0 ? D[(size_t)i * T + j - 1] : INFINITY;
        if (a0 <= a1 && a0 <= a2) { --i; --j; }                     // np.argmin: first minimum of (diag, up, left)
        else if (a1 <= a2) --i;
        else --j;
    }
    path_len[b] = len;
    dist[b] = D[(size_t)(n - 1) * T + (t - 1)] / (double)(n + t);
}

// ===================================================================================================
// Backward pass of the adaptive path (training step)
// ===================================================================================================

// d images of LossAveragingCriterion.loss (binding_loss.py:19-42) and generalised weighted sums:
//   acc[b][o][d] = sum_k wk(b, k, o) * y[b][k][d],  wk = w[b][k * ws_k + o * ws_o] * (kmask ? kmask[b][k] : 1)
//   out = sub ? coef * (sub[b][o][d] * (sum_k wk) - acc) : acc
// soft average (forward, visualisation): k = node, o = frame;  averaging-loss gradient: k = frame, o = node, sub = images.
// f32 MFMA: outputs o on the i side (two 16-row tiles per workgroup), the D axis on the j side (a wavefront owns 64 consecutive
// d), k walks the MFMA k index 4 at a time; operands go straight from global memory to registers (A = one weight per lane, B = one
// element of row k per lane, 64-byte segments per 16 lanes).  The first version held 32 accumulators per thread and broadcast the
// weights through LDS with two barriers per k: 1.22 ms at c5 (8 TFLOP/s of VALU) for 10 GFLOP.
__global__ void __launch_bounds__(256) weighted_rows_kernel(const float* __restrict__ w, const long long ws_k, const long long ws_o,
                                                            const long long ws_b, const float* __restrict__ kmask,
                                                            const float* __restrict__ y, const float* __restrict__ sub,
                                                            const float* __restrict__ log_sigma, const float coef,
                                                            float* __restrict__ out, const int Kn, const int On, const long long Dd) {
    const int b = blockIdx.z, o0 = blockIdx.y * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ij = lane & 15, kk = lane >> 4;
    const long long d0 = (long long)blockIdx.x * 256 + wave * 64;
    f32x4 acc[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float wsum[2] = {0.f, 0.f};
    const float* wb = w + (size_t)b * ws_b;
    const float* yb = y + (size_t)b * Kn * Dd;
    bool ov[2], dv[4];
#pragma unroll
    for (int it = 0; it < 2; ++it) ov[it] = o0 + 16 * it + ij < On;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) dv[jt] = d0 + 16 * jt + ij < Dd;
#pragma unroll 4
    for (int k0 = 0; k0 < Kn; k0 += 4) {
        const int k = k0 + kk;
        const bool kv = k < Kn;
        const float km = (kv && kmask) ? kmask[(size_t)b * Kn + k] : 1.f;
        float a[2], bv[4];
#pragma unroll
        for (int it = 0; it < 2; ++it)
            a[it] = (kv && ov[it]) ? wb[(size_t)k * ws_k + (size_t)(o0 + 16 * it + ij) * ws_o] * km : 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) bv[jt] = (kv && dv[jt]) ? yb[(size_t)k * Dd + d0 + 16 * jt + ij] : 0.f;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            wsum[it] += a[it];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) acc[it][jt] = mfma16(a[it], bv[jt], acc[it][jt]);
        }
    }
    const float c = sub ? coef * expf(-2.f * log_sigma[0]) : 1.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        float s = wsum[it];                                  // sum over k of the weights of output o0 + 16 it + ij
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ws = __shfl(s, 4 * kk + r);          // the accumulator rows of this lane are outputs 4 kk + r
            const int o = o0 + 16 * it + 4 * kk + r;
            if (o >= On) continue;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const long long d = d0 + 16 * jt + ij;
                if (d >= Dd) continue;
                const size_t oidx = ((size_t)b * On + o) * Dd + d;
                out[oidx] = sub ? c * (sub[oidx] * ws - acc[it][jt][r]) : acc[it][jt][r];
            }
        }
    }
}

// d loss / d log_sigma of the averaging criterion: sum over (b, n, t) of w * pad * (D - d * exp(-2 ls)) * coef, accumulated into dst
__global__ void __launch_bounds__(1024) averaging_dls_kernel(const float* __restrict__ dsum, const float* __restrict__ w,
                                                             const float* __restrict__ pad, const float* __restrict__ log_sigma,
                                                             const float D, const float coef, float* __restrict__ dst,
                                                             const int B, const int N, const int T) {
    // one workgroup (deterministic sum); 1024 threads x 4 independent elements per trip keep enough loads in flight — the first
    // version (256 threads, one element per trip) spent 1 ms in ~1600 dependent round trips
    __shared__ float red[1024];
    const float s2 = expf(-2.f * log_sigma[0]);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t total = (size_t)B * N * T;
    const size_t NT = (size_t)N * T;
    for (size_t i0 = threadIdx.x; i0 < total; i0 += 4096) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + 1024 * u;
            if (i < total) {
                const int t = (int)(i % T);
                const int b = (int)(i / NT);
                acc[u] += w[i] * pad[(size_t)b * T + t] * (D - dsum[i] * s2);
            }
        }
    }
    red[threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dst[0] += coef * red[0];
}

// attention backward, per query row (one wavefront): dS[r][t] = a_t * (dA_t - sum_t a_t dA_t) with dA_t = dO[r] . V[b][t];
// dq[r] = sum_t dS_t K[b][t] / (sqrt(dk) temp); dtemp_row[r] = -sum_t dS_t score_t / temp   (one head)
__global__ void __launch_bounds__(256) attention_bwd_row_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, const float* __restrict__ att,
                                                                const float* __restrict__ d_out, const int64_t* __restrict__ end_ind,
                                                                const float* __restrict__ temperature, float* __restrict__ dS,
                                                                float* __restrict__ dq, float* __restrict__ dtemp_row, const int M,
                                                                const int rpb, const int T, const int dk, const int nz) {
    // one workgroup per query row (same decomposition as attention_kernel): one key per thread for the two dot products, the
    // query gradient as (channel group, key group) partial sums combined in a fixed order
    extern __shared__ float smem[];
    float* p = smem;
    float* part = smem + ((T + 3) & ~3);
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int r = blockIdx.x;
    const int b = r / rpb;
    const int e = (int)end_ind[b];
    const int te = e >= T ? T - 1 : e;
    const float* kb = k + (size_t)b * T * dk;
    const float* vb = v + (size_t)b * T * nz;
    const float sq = sqrtf((float)dk), temp = temperature[0];
    const float* dor = d_out + (size_t)r * nz;
    const float* qr = q + (size_t)r * dk;
    float dot = 0.f;
    for (int t = tid; t < T; t += 256) {
        float da = 0.f;
        if (t <= te) {
            const float* vr = vb + (size_t)t * nz;
            for (int c = 0; c < nz; c += 4) {
                const float4 a4 = *reinterpret_cast<const float4*>(dor + c), b4 = *reinterpret_cast<const float4*>(vr + c);
                da = fmaf(a4.x, b4.x, da); da = fmaf(a4.y, b4.y, da); da = fmaf(a4.z, b4.z, da); da = fmaf(a4.w, b4.w, da);
            }
        }
        p[t] = da;
        dot += att[(size_t)r * T + t] * da;
    }
    dot = block_reduce(dot, red, false);
    float dtp = 0.f;
    for (int t = tid; t < T; t += 256) {
        const float a = att[(size_t)r * T + t];
        const float ds = a * (p[t] - dot);
        p[t] = ds;
        dS[(size_t)r * T + t] = ds;
        if (t <= te && ds != 0.f) {
            const float* kr = kb + (size_t)t * dk;
            float sc = 0.f;
            for (int i = 0; i < dk; i += 4) {
                const float4 a4 = *reinterpret_cast<const float4*>(qr + i), b4 = *reinterpret_cast<const float4*>(kr + i);
                sc = fmaf(a4.x, b4.x, sc); sc = fmaf(a4.y, b4.y, sc); sc = fmaf(a4.z, b4.z, sc); sc = fmaf(a4.w, b4.w, sc);
            }
            dtp -= ds * (sc / sq / temp) / temp;
        }
    }
    dtp = block_reduce(dtp, red, false);
    if (tid == 0) dtemp_row[r] = dtp;
    __syncthreads();
    const int c4n = dk / 4, G = 256 / c4n;
    const int c4 = tid % c4n, tg = tid / c4n;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tg < G) {
#pragma unroll 4
        for (int t = tg; t <= te; t += G) {
            const float4 x = *reinterpret_cast<const float4*>(kb + (size_t)t * dk + c4 * 4);
            const float w = p[t];
            acc.x = fmaf(w, x.x, acc.x); acc.y = fmaf(w, x.y, acc.y); acc.z = fmaf(w, x.z, acc.z); acc.w = fmaf(w, x.w, acc.w);
        }
    }
    reinterpret_cast<float4*>(part)[tid] = acc;
    __syncthreads();
    if (tid < c4n) {
        float4 o = reinterpret_cast<float4*>(part)[tid];
        for (int g = 1; g < G; ++g) {
            const float4 x = reinterpret_cast<float4*>(part)[g * c4n + tid];
            o.x += x.x; o.y += x.y; o.z += x.z; o.w += x.w;
        }
        const float sc = 1.f / sq / temp;
        *reinterpret_cast<float4*>(dq + (size_t)r * dk + tid * 4) = make_float4(o.x * sc, o.y * sc, o.z * sc, o.w * sc);
    }
}

// attention backward, per (sequence, frame): dK[b][t][:] = sum_j dS[(b, j)][t] q[(b, j)][:] / (sqrt(dk) temp),
// dV[b][t][:] = sum_j a[(b, j)][t] dO[(b, j)][:]; rows of dK / dV have leading dimensions ldk / ldv (level blocks side by side)
__global__ void __launch_bounds__(192) attention_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ att,
                                                               const float* __restrict__ dS, const float* __restrict__ d_out,
                                                               const float* __restrict__ temperature, float* __restrict__ dK,
                                                               const long long ldk, float* __restrict__ dV, const long long ldv,
                                                               const int rpb, const int T, const int dk, const int nz) {
    const int bt = blockIdx.x, b = bt / T, t = bt % T;
    const int c = threadIdx.x;
    if (c >= dk + nz) return;
    const float sc = 1.f / sqrtf((float)dk) / temperature[0];
    float acc = 0.f;
    if (c < dk) {
        for (int j = 0; j < rpb; ++j) {
            const size_t r = (size_t)b * rpb + j;
            acc = fmaf(dS[r * T + t], q[r * dk + c], acc);
        }
        dK[(size_t)bt * ldk + c] = acc * sc;
    } else {
        const int cc = c - dk;
        for (int j = 0; j < rpb; ++j) {
            const size_t r = (size_t)b * rpb + j;
            acc = fmaf(att[r * T + t], d_out[r * nz + cc], acc);
        }
        dV[(size_t)bt * ldv + cc] = acc;
    }
}

}  // namespace

extern "C" int gcpx_attention(const float* q, const float* k, const float* v, const int64_t* start_ind, const int64_t* end_ind,
                              const float* temperature, float* out, float* att, int32_t M, int32_t rpb, int32_t T, int32_t dk,
                              int32_t nz, int32_t heads, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(q && k && v && end_ind && temperature && out, "null pointer");
    GCPX_CHECK_ARG(M > 0 && rpb > 0 && T > 0 && T <= 4096 && heads > 0 && dk % heads == 0 && nz % heads == 0, "bad sizes");
    GCPX_CHECK_ARG((dk / heads) % 4 == 0 && (nz / heads) % 4 == 0 && nz / heads <= 1024 && 256 % (nz / heads / 4) == 0,
                   "head widths must be multiples of 4 and nz / heads / 4 a divisor of 256");
    hipLaunchKernelGGL(attention_kernel, dim3(M), dim3(256), (T + 4 + 1024) * sizeof(float), stream, q, k, v, start_ind, end_ind,
                       temperature, out, att, M, rpb, T, dk, nz, heads);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_cdist_splits(int64_t K) {
    for (int s = 8; s > 1; s >>= 1)
        if (K % (32 * s) == 0) return s;
    return 1;
}

extern "C" int gcpx_cdist(const float* x, const float* y, int32_t B, int32_t N, int32_t T, int64_t K, float* partial, float* xnorm,
                          float* ynorm, float* out, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(x && y && partial && xnorm && ynorm && out, "null pointer");
    GCPX_CHECK_ARG(B > 0 && N > 0 && T > 0 && K > 0 && K % 32 == 0, "K must be a positive multiple of 32");
    const int ns = gcpx_cdist_splits(K);
    const int tiles_n = (N + 63) / 64, tiles_t = (T + 63) / 64;
    hipLaunchKernelGGL(row_sumsq_kernel, dim3(B * N), dim3(256), 0, stream, x, (long long)(K / 4), xnorm);
    hipLaunchKernelGGL(row_sumsq_kernel, dim3(B * T), dim3(256), 0, stream, y, (long long)(K / 4), ynorm);
    hipLaunchKernelGGL(cdist_gemm_kernel, dim3(tiles_n * tiles_t, ns, B), dim3(256), 0, stream, x, y, partial, B, N, T, (int)K,
                       (int)(K / ns), tiles_t);
    const size_t total = (size_t)B * N * T;
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(cdist_finish_kernel, dim3(grid), dim3(256), 0, stream, partial, ns, xnorm, ynorm, out, B, N, T);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_soft_dtw(const float* dsum, float D, const float* temp, const int64_t* end_ind, int32_t B, int32_t N, int32_t T,
                             double* acc, float* w, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(dsum && temp && end_ind && acc && w, "null pointer");
    GCPX_CHECK_ARG(B > 0 && N >= T && T > 0, "needs at least as many nodes as frames (probabilistic_dtw.py:36)");
    if (T > 1024) {
        gcpx_set_error("gcpx_soft_dtw: sequences longer than 1024 frames are not built");
        return GCPX_ERR_UNSUPPORTED;
    }
    const int threads = (T + 63) / 64 * 64;
    hipLaunchKernelGGL(dtw_sweep_kernel, dim3(2 * B), dim3(threads), 2 * threads * sizeof(double), stream, dsum, D, temp, end_ind,
                       acc, B, N, T);
    const int G = threads <= 256 ? 4 : (threads <= 512 ? 2 : 1);
    hipLaunchKernelGGL(dtw_combine_kernel, dim3(B), dim3(threads, G), G * threads * sizeof(float), stream, dsum, D, temp, end_ind, acc, w,
                       B, N, T);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_soft_dtw_dtemp(const float* dsum, float D, const float* temp, const int64_t* end_ind, const double* acc,
                                   const float* pad_mask, const float* log_sigma, float coef, int32_t B, int32_t N, int32_t T,
                                   double* tangent, double* partial, float* dtemp, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(dsum && temp && end_ind && acc && pad_mask && log_sigma && tangent && partial && dtemp, "null pointer");
    GCPX_CHECK_ARG(B > 0 && N >= T && T > 0, "needs at least as many nodes as frames (probabilistic_dtw.py:36)");
    if (T > 1024) {
        gcpx_set_error("gcpx_soft_dtw_dtemp: sequences longer than 1024 frames are not built");
        return GCPX_ERR_UNSUPPORTED;
    }
    const int threads = (T + 63) / 64 * 64;
    hipLaunchKernelGGL(dtw_tangent_sweep_kernel, dim3(2 * B), dim3(threads), 4 * threads * sizeof(double), stream, dsum, D, temp,
                       end_ind, acc, tangent, B, N, T);
    const int G = threads <= 256 ? 4 : (threads <= 512 ? 2 : 1);
    hipLaunchKernelGGL(dtw_dtemp_combine_kernel, dim3(B), dim3(threads, G), 4 * G * threads * sizeof(double), stream, dsum, D, temp,
                       end_ind, acc, tangent, pad_mask, log_sigma, partial, B, N, T);
    hipLaunchKernelGGL(dtw_dtemp_finish_kernel, dim3(1), dim3(1), 0, stream, partial, B, coef, dtemp);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_match_stats(const float* w, const int64_t* end_ind, int32_t B, int32_t L, int32_t T, int32_t* frame2node,
                                int32_t* matched_idx, int32_t* best_t, float* entropy, float* p_n, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(w && end_ind && frame2node && best_t && entropy && p_n, "null pointer");
    GCPX_CHECK_ARG(B > 0 && L > 0 && L < 16 && T > 0, "bad sizes");
    const int N = (1 << L) - 1;
    hipLaunchKernelGGL(match_argmax_kernel, dim3((B * T + 3) / 4), dim3(256), 0, stream, w, end_ind, frame2node, matched_idx, B, L, T);
    hipLaunchKernelGGL(match_node_stats_kernel, dim3((B * N + 3) / 4), dim3(256), 0, stream, w, best_t, entropy, p_n, B * N, T);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_distance_prune(const float* dist, float threshold, const int32_t* best_t, int32_t B, int32_t N, int32_t* leave,
                                   int32_t* kept_idx, int32_t* count, int32_t* target, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(dist && leave && kept_idx && count && B > 0 && N > 1, "null pointer / bad sizes");
    GCPX_CHECK_ARG(!target || best_t, "targets need best_t");
    hipLaunchKernelGGL(distance_prune_kernel, dim3(B), dim3(64), 0, stream, dist, threshold, best_t, leave, kept_idx, count,
                       target, B, N);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_averaging_nll(const float* dsum, const float* w, const float* log_sigma, float D, int32_t B, int32_t N, int32_t T,
                                  float* nll_bt, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(dsum && w && log_sigma && nll_bt && B > 0 && N > 0 && T > 0, "null pointer / bad sizes");
    hipLaunchKernelGGL(averaging_nll_kernel, dim3(B * ((T + 63) / 64)), dim3(256), 0, stream, dsum, w, log_sigma, D, nll_bt, B, N, T);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_soft_average(const float* w, const float* x, float* out, int32_t B, int32_t N, int32_t T, int64_t D, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(w && x && out && B > 0 && N > 0 && T > 0 && D > 0, "null pointer / bad sizes");
    hipLaunchKernelGGL(weighted_rows_kernel, dim3((unsigned)((D + 255) / 256), (T + 31) / 32, B), dim3(256), 0, stream, w,
                       (long long)T, 1LL, (long long)N * T, (const float*)nullptr, x, (const float*)nullptr, (const float*)nullptr, 1.f,
                       out, N, T, (long long)D);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_dtw_align(const float* cost, const int32_t* n_len, const int32_t* t_len, int32_t B, int32_t N, int32_t T,
                              double* acc, int32_t* inds, int32_t* path, int32_t* path_len, double* dist, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(cost && acc && inds && path && path_len && dist && B > 0 && N > 0 && T > 0, "null pointer / bad sizes");
    hipLaunchKernelGGL(dtw_align_kernel, dim3(B), dim3(256), 0, stream, cost, n_len, t_len, acc, inds, path, path_len, dist, N, T);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_averaging_nll_bwd(const float* w, const float* pad_mask, const float* images, const float* traj, const float* dsum,
                                      const float* log_sigma, float coef, int32_t B, int32_t N, int32_t T, int64_t D, float* dimg,
                                      float* dlog_sigma, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(w && pad_mask && images && traj && dsum && log_sigma && dimg && dlog_sigma, "null pointer");
    GCPX_CHECK_ARG(B > 0 && N > 0 && T > 0 && D > 0, "bad sizes");
    hipLaunchKernelGGL(weighted_rows_kernel, dim3((unsigned)((D + 255) / 256), (N + 31) / 32, B), dim3(256), 0, stream, w, 1LL,
                       (long long)T, (long long)N * T, pad_mask, traj, images, log_sigma, coef, dimg, T, N, (long long)D);
    hipLaunchKernelGGL(averaging_dls_kernel, dim3(1), dim3(1024), 0, stream, dsum, w, pad_mask, log_sigma, (float)D, coef, dlog_sigma, B,
                       N, T);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_attention_bwd(const float* q, const float* k, const float* v, const float* att, const float* d_out,
                                  const int64_t* end_ind, const float* temperature, float* dS, float* dq, float* dtemp_row, float* dK,
                                  int64_t ldk, float* dV, int64_t ldv, int32_t M, int32_t rpb, int32_t T, int32_t dk, int32_t nz,
                                  void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(q && k && v && att && d_out && end_ind && temperature && dS && dq && dtemp_row && dK && dV, "null pointer");
    GCPX_CHECK_ARG(M > 0 && rpb > 0 && M % rpb == 0 && T > 0 && T <= 4096 && dk + nz <= 192, "bad sizes (one head, dk + nz <= 192)");
    GCPX_CHECK_ARG(dk % 4 == 0 && nz % 4 == 0 && 256 % (dk / 4) == 0, "dk, nz multiples of 4; dk / 4 a divisor of 256");
    hipLaunchKernelGGL(attention_bwd_row_kernel, dim3(M), dim3(256), (T + 4 + 1024) * sizeof(float), stream, q, k, v, att, d_out,
                       end_ind, temperature, dS, dq, dtemp_row, M, rpb, T, dk, nz);
    hipLaunchKernelGGL(attention_bwd_kv_kernel, dim3((M / rpb) * T), dim3(192), 0, stream, q, att, dS, d_out, temperature, dK,
                       (long long)ldk, dV, (long long)ldv, rpb, T, dk, nz);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
