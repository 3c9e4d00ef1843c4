// Small kernels: BatchNorm statistics finalisation, balanced frame binding (integer, bit-exact), gathers,
// plus the library's error string, hipGraph and event helpers.
#include "common.h"

#include <cstdarg>
#include <cstdio>

// ---------------------------------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void gcpx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gcpx_last_error(void) { return g_err; }
extern "C" int gcpx_version(void) { return GCPX_VERSION; }

namespace {

// ---------------------------------------------------------------------------------------------------
// BatchNorm: partial sums -> scale / shift
// ---------------------------------------------------------------------------------------------------
// (a single 1024-thread workgroup with coalesced column reads was tried: 23-54 us against 11-14 us for one workgroup per channel —
// one CU cannot pull the partial rows fast enough)
__global__ void __launch_bounds__(256) bn_finalize_kernel(const float* __restrict__ partial, const int n_partial,
                                                          const int pitch, const int C, const double count,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float eps,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* running_mean, float* running_var,
                                                          const float momentum, float* __restrict__ mean_out,
                                                          float* __restrict__ rstd_out) {
    const int c = blockIdx.x;
    const int rep = pitch / C;
    double s1 = 0.0, s2 = 0.0;
    // eight rows' loads in flight per thread (the sums stay in row order): rolled, every iteration was a full memory round trip and a
    // table of a few thousand rows made this launch 19 us of the forward's chain
    const int total = n_partial * rep;
    int i = threadIdx.x;
    for (; i + 7 * 256 < total; i += 8 * 256) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ii = i + u * 256, p = ii / rep, k = ii % rep;
            a[u] = partial[((size_t)p * 2 + 0) * pitch + k * C + c];
            b[u] = partial[((size_t)p * 2 + 1) * pitch + k * C + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += (double)a[u]; s2 += (double)b[u]; }
    }
    for (; i < total; i += 256) {
        const int p = i / rep, k = i % rep;
        s1 += (double)partial[((size_t)p * 2 + 0) * pitch + k * C + c];
        s2 += (double)partial[((size_t)p * 2 + 1) * pitch + k * C + c];
    }
    __shared__ double r1[256], r2[256];
    r1[threadIdx.x] = s1;
    r2[threadIdx.x] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            r1[threadIdx.x] += r1[threadIdx.x + s];
            r2[threadIdx.x] += r2[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = r1[0] / count;
        double var = r2[0] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float sc = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
        scale[c] = sc;
        shift[c] = beta[c] - (float)mean * sc;
        if (mean_out) { mean_out[c] = (float)mean; rstd_out[c] = (float)(1.0 / sqrt(var + (double)eps)); }
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Standard-normal draws inside the launch plan (Gaussian.sample() of the reference draws from torch's generator on every forward,
// tree_module.py:79-94; as a torch kernel in front of the captured graph that draw sat at the head of the forward's dependent chain):
// Philox4x32-10 counter-based generator (key = seed, counter = offset + thread index) + Box-Muller, four numbers per thread; the
// offset lives in device memory and is advanced by a second, one-thread launch behind the draw — replaying the graph draws fresh numbers.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}

__global__ void __launch_bounds__(256) randn_kernel(float* __restrict__ out, const long long n, const unsigned long long* __restrict__ state) {
    const unsigned long long seed = state[0], ctr = state[1] + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const uint4 r = philox4x32_10(make_uint4((unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u), make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
    // u in (0, 1] from the top 24 bits; two Box-Muller pairs
    const float u0 = ((r.x >> 8) + 1u) * 5.9604644775390625e-8f, u1 = (r.y >> 8) * 5.9604644775390625e-8f;
    const float u2 = ((r.z >> 8) + 1u) * 5.9604644775390625e-8f, u3 = (r.w >> 8) * 5.9604644775390625e-8f;
    const float ra = sqrtf(-2.f * __logf(u0)), rb = sqrtf(-2.f * __logf(u2));
    float sa, ca, sb, cb;
    __sincosf(6.283185307179586f * u1, &sa, &ca);
    __sincosf(6.283185307179586f * u3, &sb, &cb);
    const float v[4] = {ra * ca, ra * sa, rb * cb, rb * sb};
    if (i + 3 < n) {
        *reinterpret_cast<float4*>(out + i) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int k = 0; k < 4 && i + k < n; ++k) out[i + k] = v[k];
    }
}

__global__ void rng_advance_kernel(unsigned long long* state, const unsigned long long inc) { state[1] += inc; }

__global__ void bn_fold_kernel(const float* __restrict__ rm, const float* __restrict__ rv,
                               const float* __restrict__ gamma, const float* __restrict__ beta, const float eps,
                               const int C, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = gamma[c] / sqrtf(rv[c] + eps);
        scale[c] = sc;
        shift[c] = beta[c] - rm[c] * sc;
    }
}

// ---------------------------------------------------------------------------------------------------
// balanced binding.  Node at depth-first position p: level l = L-1-ctz(p+1), index j = (p+1) >> (L-l).
// Descend from the root with the reference's midpoint rule t = trunc((t_l + t_r) / 2) on int64
// (frame_binding.py:52-54 under torch 1.3).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long trunc_mid(long long tl, long long tr) {
    const long long s = tl + tr;
    return s >= 0 ? s / 2 : -((-s) / 2);
}

__global__ void __launch_bounds__(256) balanced_binding_kernel(const int64_t* __restrict__ end_ind, const int B,
                                                               const int L, const int T, int32_t* __restrict__ node_t,
                                                               int32_t* __restrict__ leave,
                                                               int32_t* __restrict__ frame2node,
                                                               int32_t* __restrict__ etilde_row,
                                                               int32_t* __restrict__ seq_len,
                                                               int32_t* __restrict__ node2row) {
    const int b = blockIdx.x;
    const int N = (1 << L) - 1;
    const long long end = end_ind[b];
    const int root = (N - 1) / 2;
    for (int t = threadIdx.x; t < T; t += blockDim.x) frame2node[(size_t)b * T + t] = root;   // D5: argmax of zeros = bf 0
    if (threadIdx.x == 0) seq_len[b] = (int32_t)(end + 1);
    __syncthreads();
    for (int p = threadIdx.x; p < N; p += blockDim.x) {
        const int tz = __ffs(p + 1) - 1;
        const int l = L - 1 - tz;
        const int jn = (p + 1) >> (tz + 1);
        long long tl = -1, tr = end + 1, t = 0;    // get_init_inds, frame_binding.py:62-65
        for (int k = 0; k <= l; ++k) {
            t = trunc_mid(tl, tr);
            if (k == l) break;
            if ((jn >> (l - 1 - k)) & 1) tl = t; else tr = t;
        }
        const int keep = !(t == tl || t == tr);     // frame_binding.py:47-48
        node_t[(size_t)b * N + p] = (int32_t)t;
        leave[(size_t)b * N + p] = keep;
        if (keep && t >= 0 && t < T) frame2node[(size_t)b * T + t] = p;
        if (node2row) node2row[(size_t)b * N + p] = (keep && t >= 0 && t < T) ? (int32_t)(b * T + t) : -1;
        if (etilde_row) {
            long long tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
            etilde_row[(size_t)B * ((1 << l) - 1) + (size_t)b * (1 << l) + jn] = (int32_t)(b * T + tc);
        }
    }
}

__global__ void compact_index_kernel(const int32_t* __restrict__ leave, const int B, const int N, const int T,
                                     int32_t* __restrict__ dst) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int k = 0;
    for (int p = 0; p < N; ++p)
        if (leave[(size_t)b * N + p]) {
            if (k < T) dst[(size_t)b * T + k] = p;
            ++k;
        }
    for (; k < T; ++k) dst[(size_t)b * T + k] = -1;
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                          float* __restrict__ out, const int T, const int N,
                                                          const int idx_offset, const long long row4, const int32_t* __restrict__ skip) {
    const int bt = blockIdx.x;
    const int b = bt / T;
    if (skip && skip[bt] >= 0) return;                     // (the row was written by its producer already)
    const int i = idx[bt];
    float4* o = reinterpret_cast<float4*>(out) + (size_t)bt * row4;
    if (i < 0) {
        for (long long k = threadIdx.x; k < row4; k += blockDim.x) o[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const float4* s = reinterpret_cast<const float4*>(src) + ((size_t)b * N + i + idx_offset) * row4;
        for (long long k = threadIdx.x; k < row4; k += blockDim.x) o[k] = s[k];
    }
}

// nxt[i][t] = lat[i][t+1] for t+1 < len_i, else the goal latent: second argument rows of the learned pairwise cost
// (LearnedCostEstimate list branch, cost_fcn.py:88-95: pairs of torch.cat((seq, goal)))
__global__ void seq_pairs_kernel(const float* __restrict__ lat, const int32_t* __restrict__ lengths,
                                 const float* __restrict__ goal, float* __restrict__ nxt, const int T, const int nz4) {
    const int it = blockIdx.x;
    const int i = it / T, t = it % T;
    const int len = lengths[i];
    const float4* src;
    if (t + 1 < len) src = reinterpret_cast<const float4*>(lat) + ((size_t)i * T + t + 1) * nz4;
    else if (goal) src = reinterpret_cast<const float4*>(goal) + (size_t)i * nz4;
    else src = reinterpret_cast<const float4*>(lat) + ((size_t)i * T + (len > 0 ? len - 1 : 0)) * nz4;
    float4* dst = reinterpret_cast<float4*>(nxt) + (size_t)it * nz4;
    for (int k = threadIdx.x; k < nz4; k += blockDim.x) dst[k] = src[k];
}

__global__ void __launch_bounds__(256) copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        const int rows, const long long row4, const long long sbr,
                                                        const long long dbr) {
    const int b = blockIdx.x / rows, r = blockIdx.x % rows;
    const float4* s = reinterpret_cast<const float4*>(src) + ((size_t)b * sbr + r) * row4;
    float4* d = reinterpret_cast<float4*>(dst) + ((size_t)b * dbr + r) * row4;
    for (long long k = threadIdx.x; k < row4; k += blockDim.x) d[k] = s[k];
}

// z[r] = mu[r] + exp(log_sigma[r]) * eps[r] for rows r = (b, j) with independent row maps (Gaussian.sample)
__global__ void gauss_sample_kernel(const float* __restrict__ muls, const long long mb, const long long mr,
                                    const float* __restrict__ eps, const long long eb, const long long er,
                                    float* __restrict__ z, const long long zb, const long long zr, const int M,
                                    const int rpb, const int nz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * nz) return;
    const int r = i / nz, d = i % nz;
    const int b = r / rpb, j = r % rpb;
    const float* m = muls + (size_t)b * mb + (size_t)j * mr;
    z[(size_t)b * zb + (size_t)j * zr + d] = m[d] + expf(m[nz + d]) * eps[(size_t)b * eb + (size_t)j * er + d];
}

// idx[b][t] = t for t <= end_ind[b], -1 after (pad_sequence of a prefix), seq_len[b] = end_ind[b] + 1
__global__ void seq_index_kernel(const int64_t* __restrict__ end_ind, const int B, const int T, int32_t* __restrict__ idx,
                                 int32_t* __restrict__ seq_len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i % T;
    const long long e = end_ind[b];
    idx[i] = (t <= e) ? t : -1;
    if (t == 0) seq_len[b] = (int32_t)(e + 1);
}

__global__ void masked_row_sum_kernel(const float* __restrict__ vals, const int32_t* __restrict__ lengths,
                                      float* __restrict__ out, const int n, const int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    const int len = min(lengths[i], T);
    for (int t = 0; t < len; ++t) s += vals[(size_t)i * T + t];
    out[i] = s;
}

// Hand-written planner costs over padded rollouts (gcp/planning/cem/cost_fcn.py:10-77): one workgroup per candidate, a wavefront per
// step computes that step's value over the D columns, then the steps are combined in order (final_step_weight on the last step;
// dense: sum over steps, else the last step's value).
//   kind 0 EuclideanDistance   || x_t - goal ||          1 EuclideanPathLength  || next_t - x_t ||, next = x_{t+1}, goal after the last
//   kind 2 StepPathLength      0, ..., 0, len            3 L2ImageCost          || x_t[:D] - goal || on the image columns only
__global__ void __launch_bounds__(256) rollout_cost_kernel(const float* __restrict__ x, const long long ld, const int32_t* __restrict__ lengths,
                                                           const float* __restrict__ goal, const long long goal_stride, float* __restrict__ out,
                                                           const int T, const int D, const int kind, const int dense, const float final_w) {
    extern __shared__ float steps[];
    const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int len = min(lengths[i], T);
    const float* xi = x + (size_t)i * T * ld;
    const float* g = goal + (size_t)i * goal_stride;
    for (int t = wave; t < len; t += 4) {
        float s = 0.f;
        if (kind == 2) {
            s = t == len - 1 ? (float)len : 0.f;
        } else {
            const float* a = xi + (size_t)t * ld;
            const float* b = (kind == 1 && t + 1 < len) ? a + ld : g;
            for (int d = lane; d < D; d += 64) { const float e = (kind == 1 ? b[d] - a[d] : a[d] - b[d]); s += e * e; }
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            s = sqrtf(s);
        }
        if (lane == 0) steps[t] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = 0.f;
        if (len > 0) {
            if (dense) { for (int t = 0; t + 1 < len; ++t) r += steps[t]; r += steps[len - 1] * final_w; }
            else r = steps[len - 1] * final_w;
        }
        out[i] = r;
    }
}

}  // namespace

extern "C" int gcpx_rollout_cost(const float* x, int64_t ld, const int32_t* lengths, const float* goal, int64_t goal_stride, float* out,
                                 int32_t n, int32_t T, int32_t D, int32_t kind, int32_t dense, float final_step_weight, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(x && lengths && out && n > 0 && T > 0 && D > 0 && ld >= D && kind >= 0 && kind <= 3, "null pointer / bad sizes");
    GCPX_CHECK_ARG(kind == 2 || goal, "goal is NULL");
    GCPX_CHECK_ARG(dense || kind != 1, "the path length needs dense_cost (cost_fcn.py:52)");
    hipLaunchKernelGGL(rollout_cost_kernel, dim3(n), dim3(256), T * sizeof(float), stream, x, (long long)ld, lengths, goal, (long long)goal_stride,
                       out, T, D, kind, dense, final_step_weight);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_seq_pairs(const float* lat, const int32_t* lengths, const float* goal, float* nxt, int32_t n,
                              int32_t T, int32_t nz, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(lat && lengths && nxt && n > 0 && T > 0 && nz > 0 && nz % 4 == 0, "null pointer / bad sizes");
    hipLaunchKernelGGL(seq_pairs_kernel, dim3(n * T), dim3(64), 0, stream, lat, lengths, goal, nxt, T, nz / 4);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_copy_rows(const float* src, float* dst, int32_t B, int32_t rows, int64_t row_floats,
                              int64_t src_batch_rows, int64_t dst_batch_rows, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(src && dst && B > 0 && rows > 0 && row_floats > 0 && row_floats % 4 == 0, "null pointer / bad sizes");
    hipLaunchKernelGGL(copy_rows_kernel, dim3(B * rows), dim3(256), 0, stream, src, dst, rows, (long long)(row_floats / 4),
                       (long long)src_batch_rows, (long long)dst_batch_rows);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_gauss_sample(const float* muls, int64_t mb, int64_t mr, const float* eps, int64_t eb, int64_t er,
                                 float* z, int64_t zb, int64_t zr, int32_t M, int32_t rpb, int32_t nz, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(muls && eps && z && M > 0 && rpb > 0 && nz > 0, "null pointer / bad sizes");
    hipLaunchKernelGGL(gauss_sample_kernel, dim3((M * nz + 255) / 256), dim3(256), 0, stream, muls, (long long)mb,
                       (long long)mr, eps, (long long)eb, (long long)er, z, (long long)zb, (long long)zr, M, rpb, nz);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_seq_index(const int64_t* end_ind, int32_t B, int32_t T, int32_t* idx, int32_t* seq_len, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(end_ind && idx && seq_len && B > 0 && T > 0, "null pointer / bad sizes");
    hipLaunchKernelGGL(seq_index_kernel, dim3((B * T + 255) / 256), dim3(256), 0, stream, end_ind, B, T, idx, seq_len);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_fill_zero(void* ptr, int64_t nbytes, void* stream_) {
    GCPX_CHECK_ARG(ptr && nbytes > 0, "null pointer / bad size");
    hipError_t e = hipMemsetAsync(ptr, 0, (size_t)nbytes, reinterpret_cast<hipStream_t>(stream_));
    if (e != hipSuccess) {
        gcpx_set_error("gcpx_fill_zero: %s", hipGetErrorString(e));
        return GCPX_ERR_HIP;
    }
    return GCPX_OK;
}

extern "C" int gcpx_masked_row_sum(const float* vals, const int32_t* lengths, float* out, int32_t n, int32_t T,
                                   void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(vals && lengths && out && n > 0 && T > 0, "null pointer / bad sizes");
    hipLaunchKernelGGL(masked_row_sum_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, vals, lengths, out, n, T);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_bn_finalize(const float* partial, int32_t n_partial, int32_t pitch, int32_t C, double count,
                                const float* gamma, const float* beta, float eps, float* scale, float* shift,
                                float* running_mean, float* running_var, float momentum, float* mean_out,
                                float* rstd_out, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(partial && gamma && beta && scale && shift, "null pointer");
    GCPX_CHECK_ARG((mean_out == nullptr) == (rstd_out == nullptr), "mean_out and rstd_out go together");
    GCPX_CHECK_ARG(n_partial > 0 && C > 0 && pitch >= C && pitch % C == 0 && count > 0, "bad sizes");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, stream, partial, n_partial, pitch, C, count, gamma,
                       beta, eps, scale, shift, running_mean, running_var, momentum, mean_out, rstd_out);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_randn(float* out, int64_t n, uint64_t* state, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(out && state && n > 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "null pointer / n <= 0 / out not 16-byte aligned");
    const long long threads = (n + 3) / 4;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL(randn_kernel, dim3(blocks), dim3(256), 0, stream, out, (long long)n, reinterpret_cast<const unsigned long long*>(state));
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, stream, reinterpret_cast<unsigned long long*>(state), (unsigned long long)blocks * 256);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_bn_fold(const float* running_mean, const float* running_var, const float* gamma,
                            const float* beta, float eps, int32_t C, float* scale, float* shift, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(running_mean && running_var && gamma && beta && scale && shift && C > 0, "null pointer / C <= 0");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, running_mean, running_var, gamma, beta,
                       eps, C, scale, shift);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_balanced_binding(const int64_t* end_ind, int32_t B, int32_t L, int32_t T, int32_t* node_t,
                                     int32_t* leave, int32_t* frame2node, int32_t* etilde_row, int32_t* seq_len,
                                     int32_t* node2row, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(end_ind && node_t && leave && frame2node && seq_len, "null pointer");
    GCPX_CHECK_ARG(B > 0 && L > 0 && L < 20 && T > 0, "bad sizes");
    hipLaunchKernelGGL(balanced_binding_kernel, dim3(B), dim3(256), 0, stream, end_ind, B, L, T, node_t, leave,
                       frame2node, etilde_row, seq_len, node2row);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_compact_index(const int32_t* leave, int32_t B, int32_t N, int32_t T, int32_t* dst_idx,
                                  void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(leave && dst_idx && B > 0 && N > 0 && T > 0, "null pointer / bad sizes");
    hipLaunchKernelGGL(compact_index_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, leave, B, N, T, dst_idx);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_gather_rows(const float* src, const int32_t* idx, float* out, int32_t B, int32_t T, int32_t N,
                                int32_t idx_offset, int64_t row_floats, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(src && idx && out && B > 0 && T > 0 && N > 0, "null pointer / bad sizes");
    GCPX_CHECK_ARG(row_floats > 0 && row_floats % 4 == 0, "row_floats must be a positive multiple of 4");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(B * T), dim3(256), 0, stream, src, idx, out, T, N,
                       idx_offset, (long long)(row_floats / 4), static_cast<const int32_t*>(nullptr));
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_gather_rows_rest(const float* src, const int32_t* idx, float* out, int32_t B, int32_t T, int32_t N, int32_t idx_offset,
                                     int64_t row_floats, const int32_t* written, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(src && idx && out && written && B > 0 && T > 0 && N > 0, "null pointer / bad sizes");
    GCPX_CHECK_ARG(row_floats > 0 && row_floats % 4 == 0, "row_floats must be a positive multiple of 4");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(B * T), dim3(256), 0, stream, src, idx, out, T, N,
                       idx_offset, (long long)(row_floats / 4), written);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

// ---------------------------------------------------------------------------------------------------
// hipGraph + events
// ---------------------------------------------------------------------------------------------------
#define GCPX_HIP(call)                                                            \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) {                                                   \
            gcpx_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e_)); \
            return GCPX_ERR_HIP;                                                  \
        }                                                                         \
    } while (0)

extern "C" int gcpx_graph_begin(void* stream) {
    GCPX_HIP(hipStreamBeginCapture(reinterpret_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
    return GCPX_OK;
}

extern "C" int gcpx_graph_end(void* stream, void** graph_exec) {
    GCPX_CHECK_ARG(graph_exec != nullptr, "graph_exec is NULL");
    hipGraph_t graph = nullptr;
    GCPX_HIP(hipStreamEndCapture(reinterpret_cast<hipStream_t>(stream), &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        gcpx_set_error("gcpx_graph_end: hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return GCPX_ERR_HIP;
    }
    *graph_exec = exec;
    return GCPX_OK;
}

extern "C" int gcpx_graph_launch(void* graph_exec, void* stream) {
    GCPX_HIP(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), reinterpret_cast<hipStream_t>(stream)));
    return GCPX_OK;
}

extern "C" int gcpx_graph_destroy(void* graph_exec) {
    GCPX_HIP(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec)));
    return GCPX_OK;
}

extern "C" int gcpx_stream_create(void** stream) {
    GCPX_CHECK_ARG(stream != nullptr, "stream is NULL");
    hipStream_t s;
    GCPX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return GCPX_OK;
}

extern "C" int gcpx_stream_create_priority(void** stream, int level) {
    GCPX_CHECK_ARG(stream != nullptr, "stream is NULL");
    int least = 0, greatest = 0;                              // numerically: greatest priority <= least priority
    GCPX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int prio = level < 0 ? greatest : (level > 0 ? least : (least + greatest) / 2);
    hipStream_t s;
    GCPX_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
    *stream = s;
    return GCPX_OK;
}

extern "C" int gcpx_stream_destroy(void* stream) {
    GCPX_HIP(hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)));
    return GCPX_OK;
}

extern "C" int gcpx_stream_wait_event(void* stream, void* ev) {
    GCPX_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(ev), 0));
    return GCPX_OK;
}

extern "C" int gcpx_event_create(void** ev) {
    GCPX_CHECK_ARG(ev != nullptr, "ev is NULL");
    hipEvent_t e;
    GCPX_HIP(hipEventCreate(&e));
    *ev = e;
    return GCPX_OK;
}

// ordering-only event (fork / join of the launch lanes): no timestamps, device-scope release — the default event flushes to
// system scope at every record, which stalls the recording stream between two small kernels
extern "C" int gcpx_event_create_sync(void** ev) {
    GCPX_CHECK_ARG(ev != nullptr, "ev is NULL");
    hipEvent_t e;
    GCPX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToDevice));
    *ev = e;
    return GCPX_OK;
}

extern "C" int gcpx_event_record(void* ev, void* stream) {
    GCPX_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), reinterpret_cast<hipStream_t>(stream)));
    return GCPX_OK;
}

extern "C" int gcpx_event_elapsed_ms(void* start, void* stop, float* ms) {
    GCPX_CHECK_ARG(ms != nullptr, "ms is NULL");
    GCPX_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    GCPX_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return GCPX_OK;
}

extern "C" int gcpx_event_destroy(void* ev) {
    GCPX_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
    return GCPX_OK;
}
