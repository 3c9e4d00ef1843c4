// Training-time paths of the auxiliary models (inverse model, cost model) and the sampled sequence length.
//   index draws            gcp/prediction/models/auxilliary_models/inverse_mdl.py:84-104, cost_mdl.py:105-107
//   ground-truth cost      cost_mdl.py:101-117 with EuclideanPathLength (gcp/planning/cem/cost_fcn.py:14-21,49-54)
//   sampled length         gcp/prediction/models/base_gcp.py:219-226, auxilliary_models/misc.py:38-51
// Integer / HBM-bound bookkeeping: no MFMA here.
#include "common.h"

namespace {

// u [4][B] uniform draws -> the four index vectors (float64 arithmetic; same formulas as video-gcp_amd/synthetic.py:aux_indices).
// GAUSS: the draws are standard-normal numbers instead (they come out of the same generator launch as the latent noise): their
// normal CDF is uniform on (0, 1)
template <bool GAUSS>
__global__ void aux_sample_indices_kernel(const long long* __restrict__ end_ind, const float* __restrict__ u, const int B,
                                          const int temp_dist, long long* __restrict__ t0, long long* __restrict__ t1,
                                          long long* __restrict__ cs, long long* __restrict__ ce) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const long long e = end_ind[b];
    const double ed = (double)e;
    auto uni = [&](int k) { const double v = (double)u[k * B + b]; return GAUSS ? 0.5 * erfc(-v * 0.70710678118654752440) : v; };
    // the reference asserts end_ind >= temp_dist (inverse_mdl.py:93); a shorter sequence must not turn into a negative frame index
    // (the gather rows and action_seq[b, t0] are addressed with it): t0 is clamped to >= 0, t1 to <= end_ind
    long long a = (long long)floor(uni(0) * (ed - temp_dist + 1));
    a = max(min(a, e - temp_dist), 0ll);
    long long d = (long long)floor(uni(1) * temp_dist);
    d = min(d, (long long)temp_dist - 1);
    long long s = (long long)floor(uni(2) * ed);
    s = min(s, e - 1);
    long long w = (long long)floor(uni(3) * (ed - (double)s));
    w = min(w, e - s - 1);
    t0[b] = a;
    t1[b] = min(a + 1 + d, max(e, 0ll));
    cs[b] = s;
    ce[b] = s + 1 + w;
}

// absolute row numbers for the gather-on-load sources of the two Predictors
__global__ void aux_index_rows_kernel(const long long* __restrict__ t0, const long long* __restrict__ t1,
                                      const long long* __restrict__ cs, const long long* __restrict__ ce, const int B, const int T,
                                      const int Wd, int* __restrict__ rows) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    rows[b] = b * T + (int)t0[b];              // enc_traj_seq[b, t0]
    rows[B + b] = b * Wd + (int)t1[b];         // model_enc_seq[b, t1]
    rows[2 * B + b] = b * Wd + (int)cs[b];     // model_enc_seq[b, start]
    rows[3 * B + b] = b * Wd + (int)ce[b];     // model_enc_seq[b, end]
}

// One wavefront owns one row (the last axis) of one sequence and walks the steps s .. e-1 keeping the previous frame's row in
// registers: every element of the segment is read once.  partial[b][row] = sum_t || x[b,t+1,row,:] - x[b,t,row,:] ||_2.
template <int VEC>
__global__ void __launch_bounds__(256) path_cost_rows_kernel(const float* __restrict__ x, const long long* __restrict__ cs,
                                                             const long long* __restrict__ ce, const int T, const int rows,
                                                             const int row_len, float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (row >= rows) return;
    const int s = (int)cs[b], e = (int)ce[b];
    const size_t frame = (size_t)rows * row_len;
    const float* p = x + ((size_t)b * T + s) * frame + (size_t)row * row_len;
    float acc = 0.f;
    // row_len <= 64 * VEC * NCH handled by looping chunks; the common case (64-pixel image rows) is one element per lane
    for (int t = s; t < e; ++t) {
        float d2 = 0.f;
        for (int c = lane * VEC; c < row_len; c += 64 * VEC) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if (c + v < row_len) {
                    const float d = p[frame + c + v] - p[c + v];
                    d2 = fmaf(d, d, d2);
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) d2 += __shfl_xor(d2, o);
        acc += sqrtf(d2);
        p += frame;
    }
    if (lane == 0) partial[(size_t)b * rows + row] = acc;
}

__global__ void __launch_bounds__(256) path_cost_finish_kernel(const float* __restrict__ partial, const int rows,
                                                               float* __restrict__ out) {
    __shared__ float red[256];
    const int b = blockIdx.x;
    float v = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) v += partial[(size_t)b * rows + i];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[b] = red[0];
}

// inverse CDF of softmax(logits[b]) at u[b] (float64), clamped to >= min_len (base_gcp.py:222): one wavefront per sequence
__global__ void __launch_bounds__(64) sample_length_kernel(const float* __restrict__ logits, const float* __restrict__ u,
                                                           const int T, const int min_len, long long* __restrict__ end_ind) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* l = logits + (size_t)b * T;
    double m = -INFINITY;
    for (int t = lane; t < T; t += 64) m = fmax(m, (double)l[t]);
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    double se = 0.0;
    for (int t = lane; t < T; t += 64) se += exp((double)l[t] - m);
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
    if (lane == 0) {
        // sequential cumulative sum: T <= a few hundred, once per rollout
        const double target = (double)u[b];
        double c = 0.0;
        int idx = 0;
        for (int t = 0; t < T; ++t) {
            c += exp((double)l[t] - m) / se;
            if (c <= target) idx = t + 1;
        }
        idx = min(idx, T - 1);
        end_ind[b] = max(idx, min_len);
    }
}

}  // namespace

#define STREAM() hipStream_t stream = reinterpret_cast<hipStream_t>(stream_)

extern "C" int gcpx_aux_sample_indices(const int64_t* end_ind, const float* u, int32_t B, int32_t temp_dist, int64_t* inv_t0,
                                       int64_t* inv_t1, int64_t* cost_start, int64_t* cost_end, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(end_ind && u && inv_t0 && inv_t1 && cost_start && cost_end && B > 0 && temp_dist >= 1, "bad arguments");
    hipLaunchKernelGGL(aux_sample_indices_kernel<false>, dim3((B + 63) / 64), dim3(64), 0, stream, (const long long*)end_ind, u, B, temp_dist,
                       (long long*)inv_t0, (long long*)inv_t1, (long long*)cost_start, (long long*)cost_end);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_aux_sample_indices_gauss(const int64_t* end_ind, const float* n, int32_t B, int32_t temp_dist, int64_t* inv_t0,
                                             int64_t* inv_t1, int64_t* cost_start, int64_t* cost_end, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(end_ind && n && inv_t0 && inv_t1 && cost_start && cost_end && B > 0 && temp_dist >= 1, "bad arguments");
    hipLaunchKernelGGL(aux_sample_indices_kernel<true>, dim3((B + 63) / 64), dim3(64), 0, stream, (const long long*)end_ind, n, B, temp_dist,
                       (long long*)inv_t0, (long long*)inv_t1, (long long*)cost_start, (long long*)cost_end);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_aux_index_rows(const int64_t* inv_t0, const int64_t* inv_t1, const int64_t* cost_start, const int64_t* cost_end,
                                   int32_t B, int32_t T, int32_t Wd, int32_t* rows, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(inv_t0 && inv_t1 && cost_start && cost_end && rows && B > 0 && T > 0 && Wd > 0, "bad arguments");
    hipLaunchKernelGGL(aux_index_rows_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, (const long long*)inv_t0, (const long long*)inv_t1,
                       (const long long*)cost_start, (const long long*)cost_end, B, T, Wd, rows);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_path_cost(const float* x, const int64_t* start_idx, const int64_t* end_idx, int32_t B, int32_t T, int32_t rows,
                              int32_t row_len, float* partial, float* out, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(x && start_idx && end_idx && partial && out && B > 0 && T > 1 && rows > 0 && row_len > 0, "bad arguments");
    hipLaunchKernelGGL(path_cost_rows_kernel<1>, dim3((rows + 3) / 4, B), dim3(256), 0, stream, x, (const long long*)start_idx,
                       (const long long*)end_idx, T, rows, row_len, partial);
    hipLaunchKernelGGL(path_cost_finish_kernel, dim3(B), dim3(256), 0, stream, partial, rows, out);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_sample_length(const float* logits, const float* u, int32_t B, int32_t T, int32_t min_len, int64_t* end_ind,
                                  void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(logits && u && end_ind && B > 0 && T > 0 && min_len >= 0 && min_len < T, "bad arguments");
    hipLaunchKernelGGL(sample_length_kernel, dim3(B), dim3(64), 0, stream, logits, u, T, min_len, (long long*)end_ind);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
