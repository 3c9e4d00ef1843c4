// Element-wise / gather pieces of the explicit backward pass of the gcp_tree training step on gfx950
// (the reference leaves all of this to torch autograd: /root/reference/gcp/prediction/train.py:155-163).
// Everything here is HBM- or latency-bound bookkeeping between the MFMA launches (gcpx_gemm / gcpx_conv3x3 with
// transposed packs for data gradients, gcpx_wgrad for weight gradients).  All reductions are deterministic
// (per-workgroup partial sums combined in a fixed order; no atomics).
#include "common.h"
#include <algorithm>

namespace {

constexpr int ACT_BLOCKS = 1024;   // workgroups of the activation-backward kernels (= rows of their stats partials)
constexpr int GN_ROWS_PER_BLOCK = 8;

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------------------------------------------
// LSTM cell backward
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lstm_bwd_kernel(const gcpx_lstm_bwd_args a) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.M * a.H) return;
    const int r = idx / a.H, u = idx % a.H;
    lstm_bwd_cell(a, r, u, a.dh_dense ? a.dh_dense[(size_t)r * a.dh_stride + u] : 0.f);
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm + LeakyReLU backward: one wavefront per row, lane handles channels lane, lane + 64, ...
// ---------------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) gn_lrelu_bwd_kernel(const float* __restrict__ u, const float* __restrict__ da,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ du, float* __restrict__ partial, const int M,
                                                           const int groups, const float eps, const float slope) {
    constexpr int PER = (C + 63) / 64;           // channels per lane
    __shared__ float red[4][2][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cpg = C / groups;                   // channels per group: a power of two <= 16 here
    float gsum[PER], bsum[PER], gam[PER], bet[PER];
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        const int c = lane + 64 * p;
        gsum[p] = bsum[p] = 0.f;
        gam[p] = c < C ? gamma[c] : 0.f;
        bet[p] = c < C ? beta[c] : 0.f;
    }
    const int row0 = blockIdx.x * GN_ROWS_PER_BLOCK;
    for (int rr = wave; rr < GN_ROWS_PER_BLOCK; rr += 4) {
        const int r = row0 + rr;
        if (r >= M) break;
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int c = lane + 64 * p;
            const bool ok = c < C;
            const float x = ok ? u[(size_t)r * C + c] : 0.f;
            // group statistics: lanes of one group are contiguous (cpg <= 16 divides 64)
            float s = x;
            for (int m = 1; m < cpg; m <<= 1) s += __shfl_xor(s, m);
            const float mean = s / cpg;
            const float d = x - mean;
            float ss = d * d;
            for (int m = 1; m < cpg; m <<= 1) ss += __shfl_xor(ss, m);
            const float rstd = rsqrtf(ss / cpg + eps);
            const float xh = d * rstd;
            const float y = xh * gam[p] + bet[p];
            float dy = ok ? da[(size_t)r * C + c] : 0.f;
            dy *= (y > 0.f ? 1.f : slope);
            gsum[p] += dy * xh;
            bsum[p] += dy;
            const float dxh = dy * gam[p];
            float m1 = dxh, m2 = dxh * xh;
            for (int m = 1; m < cpg; m <<= 1) { m1 += __shfl_xor(m1, m); m2 += __shfl_xor(m2, m); }
            m1 /= cpg; m2 /= cpg;
            if (ok) du[(size_t)r * C + c] = rstd * (dxh - m1 - xh * m2);
        }
    }
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        const int c = lane + 64 * p;
        if (c < C) { red[wave][0][c] = gsum[p]; red[wave][1][c] = bsum[p]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i % C;
        partial[((size_t)blockIdx.x * 2 + which) * C + c] =
            (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

__global__ void __launch_bounds__(256) lrelu_bwd_kernel(const float* __restrict__ a, const float* __restrict__ dy,
                                                        float* __restrict__ dx, const long long n, const float slope) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dx[i] = dy[i] * (a[i] > 0.f ? 1.f : slope);
}

// dst[i] (+)= sum_p partial[p * stride + i]: a workgroup owns 16 outputs, its 16 thread groups take every 16th partial and
// are combined through LDS in a fixed order (deterministic)
__global__ void __launch_bounds__(256) reduce_partials_kernel(const float* __restrict__ partial, const int n, const long long stride,
                                                              const int len, float* __restrict__ dst, const int accumulate) {
    __shared__ float red[16][17];
    const int oi = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + oi;
    float s = 0.f;
    if (i < len)
        for (int p = zl; p < n; p += 16) s += partial[(size_t)p * stride + i];
    red[zl][oi] = s;
    __syncthreads();
    if (zl != 0 || i >= len) return;
    s = 0.f;
#pragma unroll
    for (int z = 0; z < 16; ++z) s += red[z][oi];
    dst[i] = accumulate ? dst[i] + s : s;
}

// ---------------------------------------------------------------------------------------------------
// latent variables
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kl_bwd_kernel(const float* __restrict__ qz, const float* __restrict__ pz,
                                                     float* __restrict__ dqz, float* __restrict__ dpz, const int N, const int nz,
                                                     const long long batch_stride, const long long node_stride,
                                                     const float free_nats, float coef, const int total,
                                                     const float* __restrict__ node_weight, const long long weight_bstride,
                                                     const float* __restrict__ coef_dev) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    if (coef_dev) coef *= *coef_dev;                         // (the scheduled part of the KL weight lives in device memory)
    const int d = idx % nz, n = (idx / nz) % N, b = idx / (nz * N);
    const size_t o = (size_t)b * batch_stride + (size_t)n * node_stride;
    const float mq = qz[o + d], lq = qz[o + nz + d], mp = pz[o + d], lp = pz[o + nz + d];
    const float diff = mq - mp;
    const float e2q = expf(2.f * lq), ie2p = expf(-2.f * lp);
    const float kl = lp - lq + (e2q + diff * diff) * 0.5f * ie2p - 0.5f;
    const float wgt = node_weight ? node_weight[(size_t)b * weight_bstride + n] : 1.f;
    const float g = kl > free_nats ? coef * wgt : 0.f;  // clamp(min=free_nats): no gradient below the floor
    dqz[o + d] = g * diff * ie2p;
    dpz[o + d] = -g * diff * ie2p;
    dqz[o + nz + d] = g * (e2q * ie2p - 1.f);
    dpz[o + nz + d] = g * (1.f - (e2q + diff * diff) * ie2p);
}

__global__ void __launch_bounds__(256) latent_bwd_kernel(const float* __restrict__ dqz_pos, const float* __restrict__ dpz_pos,
                                                         const float* __restrict__ qz_pos, const long long pb, const long long prow,
                                                         const float* __restrict__ eps, const long long eb, const long long erow,
                                                         const float* __restrict__ dz0, const long long ldz0,
                                                         const float* __restrict__ dz1, const long long ldz1,
                                                         float* __restrict__ dq_out, float* __restrict__ dp_out, const int M,
                                                         const int rpb, const int nz) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * nz) return;
    const int r = idx / nz, d = idx % nz;
    const int b = r / rpb, j = r % rpb;
    const size_t o = (size_t)b * pb + (size_t)j * prow;
    float dz = dz0[(size_t)r * ldz0 + d];
    if (dz1) dz += dz1[(size_t)r * ldz1 + d];
    const float ls = qz_pos[o + nz + d];
    const float e = eps[(size_t)b * eb + (size_t)j * erow + d];
    dq_out[(size_t)r * 2 * nz + d] = dqz_pos[o + d] + dz;
    dq_out[(size_t)r * 2 * nz + nz + d] = dqz_pos[o + nz + d] + dz * expf(ls) * e;
    dp_out[(size_t)r * 2 * nz + d] = dpz_pos[o + d];
    dp_out[(size_t)r * 2 * nz + nz + d] = dpz_pos[o + nz + d];
}

// grid: (ceil(width/64), n + 1 parent slots, B)
__global__ void __launch_bounds__(64) tree_accum_kernel(const gcpx_tree_accum_args a) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= a.width) return;
    const int k = blockIdx.y, b = blockIdx.z, n = a.n;
    for (int s = 0; s < a.nsrc; ++s) {
        const gcpx_tree_accum_src src = a.src[s];
        float acc = 0.f;
        if (k < n && src.off_left >= 0) acc += src.ptr[(size_t)(b * n + k) * src.ld + src.off_left + c];
        if (k >= 1 && src.off_right >= 0) acc += src.ptr[(size_t)(b * n + k - 1) * src.ld + src.off_right + c];
        if (k == 0 && src.off_ctx0 >= 0)
            for (int i = 0; i < n; ++i) acc += src.ptr[(size_t)(b * n + i) * src.ld + src.off_ctx0 + c];
        if (k == n && src.off_ctxg >= 0)
            for (int i = 0; i < n; ++i) acc += src.ptr[(size_t)(b * n + i) * src.ld + src.off_ctxg + c];
        float* d = a.dst + (size_t)b * a.dst_sb + (size_t)k * a.slot_stride + src.dst_col + c;
        *d += acc;
    }
}

__global__ void __launch_bounds__(128) timestep_scatter_kernel(const float* __restrict__ det, const long long db, const long long dp,
                                                               const int* __restrict__ node_t, float* __restrict__ out, const int N,
                                                               const int T, const int nz) {
    const int t = blockIdx.x, b = blockIdx.y;
    for (int c = threadIdx.x; c < nz; c += 128) {
        float s = 0.f;
        for (int p = 0; p < N; ++p)
            if (node_t[b * N + p] == t) s += det[(size_t)b * db + (size_t)p * dp + c];
        out[((size_t)b * T + t) * nz + c] = s;
    }
}

__global__ void __launch_bounds__(256) add_rows_kernel(float* __restrict__ dst, const long long dst_sb, const long long dst_sr,
                                                       const float* __restrict__ src1, const float* __restrict__ src2,
                                                       const int rpb, const int width, const int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % width, r = idx / width;
    const int b = r / rpb, j = r % rpb;
    float v = src1[idx];
    if (src2) v += src2[idx];
    dst[(size_t)b * dst_sb + (size_t)j * dst_sr + c] += v;
}

__global__ void __launch_bounds__(256) zero_unmapped_rows_kernel(float4* __restrict__ p, const long long row_f4, const int* __restrict__ row2frame) {
    const int r = blockIdx.y;
    if (row2frame[r] >= 0) return;
    float4* row = p + (size_t)r * row_f4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < row_f4; i += (long long)gridDim.x * 256) row[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// rows (b, j) of `width` floats between two strided layouts.  mode 0: dst = src; 1: dst += src; 2: dst[b] += sum over j of src[b][j]
// (dst_sr unused; the sum runs over j in order: deterministic)
__global__ void __launch_bounds__(256) rows_strided_kernel(float* __restrict__ dst, const long long dst_sb, const long long dst_sr,
                                                           const float* __restrict__ src, const long long src_sb, const long long src_sr,
                                                           const int rpb, const int width, const int mode, const int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % width, r = idx / width;
    if (mode == 2) {
        const float* sp = src + (size_t)r * src_sb + c;
        float v = 0.f;
        for (int j = 0; j < rpb; ++j) v += sp[(size_t)j * src_sr];
        dst[(size_t)r * dst_sb + c] += v;
        return;
    }
    const int b = r / rpb, j = r % rpb;
    const float v = src[(size_t)b * src_sb + (size_t)j * src_sr + c];
    float* d = dst + (size_t)b * dst_sb + (size_t)j * dst_sr + c;
    *d = mode == 1 ? *d + v : v;
}

// dx[r][c] = dy[r][c] * (1 - y[r][c]^2): backward of y = tanh(u) (the non-LSTM subgoal predictor's output, tree_module.py:109-110);
// dy / y rows (b, j) at base + b*sb + j*sr, dx dense [B * rpb][width]
__global__ void __launch_bounds__(256) tanh_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                                            const long long sb, const long long sr, const int rpb, const int width,
                                                            const long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % width);
    const long long r = i / width;
    const long long o = (r / rpb) * sb + (r % rpb) * sr + c;
    const float t = y[o];
    dx[i] = dy[o] * (1.f - t * t);
}

__global__ void __launch_bounds__(256) index_offset_kernel(const int* __restrict__ idx, int* __restrict__ out, const int T,
                                                           const int stride, const int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = idx[i] + (i / T) * stride;
}

__global__ void __launch_bounds__(256) index_fill_kernel(int* __restrict__ out, const int n, const int v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v;
}

__global__ void __launch_bounds__(256) index_inverse_kernel(const int* __restrict__ fwd, const int n, int* __restrict__ inv, const int n_inv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int r = fwd[i];
        if (r >= 0 && r < n_inv) inv[r] = i;
    }
}

// ---------------------------------------------------------------------------------------------------
// conv stacks
// ---------------------------------------------------------------------------------------------------
// items = float4 groups of the output [F][H][W][C]; a thread keeps a fixed channel group (1024 % C == 0)
__global__ void __launch_bounds__(256) act_bwd_kernel(const gcpx_actbwd_args a) {
    const int C = a.C, C4 = C / 4;
    const long long total = (long long)a.F * a.H * a.W * C4;
    const int c = (threadIdx.x * 4) % C;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 mu = sh, rs = sc;
    if (a.scale) { sc = *reinterpret_cast<const float4*>(a.scale + c); sh = *reinterpret_cast<const float4*>(a.shift + c); }
    if (a.mean) { mu = *reinterpret_cast<const float4*>(a.mean + c); rs = *reinterpret_cast<const float4*>(a.rstd + c); }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int Hd = a.up ? 2 * a.H : a.H, Wd = a.up ? 2 * a.W : a.W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p = i / C4;
        const int x = (int)(p % a.W); p /= a.W;
        const int y = (int)(p % a.H);
        const int f = (int)(p / a.H);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int fs = 0; fs < a.fsum; ++fs) {
            const float* base = a.da + ((size_t)(f * a.fsum + fs) * Hd * Wd) * a.ldc + a.c_off + c;
            if (a.up) {
#pragma unroll
                for (int ty = 0; ty < 4; ++ty) {
                    const int yy = min(max(2 * y - 1 + ty, 0), Hd - 1);
                    const float wy = (ty == 0 || ty == 3) ? 0.25f : 0.75f;
#pragma unroll
                    for (int tx = 0; tx < 4; ++tx) {
                        const int xx = min(max(2 * x - 1 + tx, 0), Wd - 1);
                        const float w = wy * ((tx == 0 || tx == 3) ? 0.25f : 0.75f);
                        const float4 v = *reinterpret_cast<const float4*>(base + ((size_t)yy * Wd + xx) * a.ldc);
                        g.x += w * v.x; g.y += w * v.y; g.z += w * v.z; g.w += w * v.w;
                    }
                }
            } else {
                const float4 v = *reinterpret_cast<const float4*>(base + ((size_t)y * Wd + x) * a.ldc);
                g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
            }
        }
        const size_t o = (size_t)i * 4;
        if (a.add) {
            const float4 v = *reinterpret_cast<const float4*>(a.add + o);
            g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
        }
        if (a.r) {
            const float4 rv = *reinterpret_cast<const float4*>(a.r + o);
            const float yv[4] = {fmaf(rv.x, sc.x, sh.x), fmaf(rv.y, sc.y, sh.y), fmaf(rv.z, sc.z, sh.z), fmaf(rv.w, sc.w, sh.w)};
            float gv[4] = {g.x, g.y, g.z, g.w};
            const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
            const float mm[4] = {mu.x, mu.y, mu.z, mu.w}, ss[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (a.act == GCPX_ACT_LRELU) gv[k] *= (yv[k] > 0.f ? 1.f : 0.2f);
                s1[k] += gv[k];
                s2[k] += gv[k] * (rr[k] - mm[k]) * ss[k];
            }
            g = make_float4(gv[0], gv[1], gv[2], gv[3]);
        }
        *reinterpret_cast<float4*>(a.dy + o) = g;
    }
    if (a.stats_partial) {
        __shared__ float red[256][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[threadIdx.x][k] = s1[k]; red[threadIdx.x][4 + k] = s2[k]; }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += 256) {
            const int which = i / C, ch = i % C;
            float s = 0.f;
            for (int t = ch / 4; t < 256; t += C4) s += red[t][which * 4 + (ch & 3)];
            a.stats_partial[((size_t)blockIdx.x * 2 + which) * C + ch] = s;
        }
    }
}

// The two passes over the data gradient of an upsampling decoder block's concatenated input [F][2H][2W][ldc] in ONE: the transpose of
// the bilinear x2 (a 4 x 4 window per low-resolution pixel) for
//   * the channels that came from the previous block (a.c_off .. + a.C): times the activation's derivative, BatchNorm-backward sums,
//     stored per frame (what gcpx_act_bwd with up = 1 does), and
//   * the channels that came from the skip connection of I_0 (c_off_s .. + Cs): summed over the rpb node frames of a sequence, in frame
//     order (what gcpx_act_bwd with fsum = rpb does) — the skip activations are one per sequence (base_gcp.py:190).
// Both halves of a pixel sit in the same 128-byte lines, so each pass alone moved the whole tensor (1.07 GB at c2 for the last block).
// A thread owns one float4 channel group of one low-resolution pixel of one SEQUENCE and walks that sequence's frames.
__global__ void __launch_bounds__(256) act_skip_bwd_kernel(const gcpx_actbwd_args a, float* __restrict__ ds, const int c_off_s, const int Cs,
                                                           const int rpb) {
    const int Ca4 = a.C / 4, G = Ca4 + Cs / 4;
    const int B = a.F / rpb;
    const long long total = (long long)B * a.H * a.W * G;
    const int cg = threadIdx.x % G;                                  // (256 % G == 0: fixed per thread over the grid-stride loop)
    const bool skip = cg >= Ca4;
    const int c = skip ? (cg - Ca4) * 4 : cg * 4;
    const int csrc = skip ? c_off_s + c : a.c_off + c;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 mu = sh, rs = sc;
    if (!skip) {
        if (a.scale) { sc = *reinterpret_cast<const float4*>(a.scale + c); sh = *reinterpret_cast<const float4*>(a.shift + c); }
        if (a.mean) { mu = *reinterpret_cast<const float4*>(a.mean + c); rs = *reinterpret_cast<const float4*>(a.rstd + c); }
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int Hd = 2 * a.H, Wd = 2 * a.W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long p = i / G;
        const int x = (int)(p % a.W); p /= a.W;
        const int y = (int)(p % a.H);
        const int b = (int)(p / a.H);
        int yy[4], xx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            yy[t] = min(max(2 * y - 1 + t, 0), Hd - 1);
            xx[t] = min(max(2 * x - 1 + t, 0), Wd - 1);
        }
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);                    // skip half: ONE running sum over frames and taps (as gcpx_act_bwd with fsum)
        // the 16 window loads of the NEXT frame are in flight while this frame's are summed (a thread walks rpb frames: without the
        // second register set every frame would cost a full memory round trip)
        float4 v[16], vn[16];
        auto load_window = [&](const int fs, float4 (&w)[16]) __attribute__((always_inline)) {
            const float* base = a.da + ((size_t)(b * rpb + min(fs, rpb - 1)) * Hd * Wd) * a.ldc + csrc;
#pragma unroll
            for (int ty = 0; ty < 4; ++ty)
#pragma unroll
                for (int tx = 0; tx < 4; ++tx) w[ty * 4 + tx] = *reinterpret_cast<const float4*>(base + ((size_t)yy[ty] * Wd + xx[tx]) * a.ldc);
        };
        load_window(0, vn);
        for (int fs = 0; fs < rpb; ++fs) {
            const int f = b * rpb + fs;
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = vn[t];
            load_window(fs + 1, vn);
            if (!skip) g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ty = 0; ty < 4; ++ty) {
                const float wy = (ty == 0 || ty == 3) ? 0.25f : 0.75f;
#pragma unroll
                for (int tx = 0; tx < 4; ++tx) {
                    const float w = wy * ((tx == 0 || tx == 3) ? 0.25f : 0.75f);
                    const float4 u = v[ty * 4 + tx];
                    g.x += w * u.x; g.y += w * u.y; g.z += w * u.z; g.w += w * u.w;
                }
            }
            if (skip) continue;
            const size_t o = ((((size_t)f * a.H + y) * a.W + x) * a.C) + c;
            if (a.add) {
                const float4 av = *reinterpret_cast<const float4*>(a.add + o);
                g.x += av.x; g.y += av.y; g.z += av.z; g.w += av.w;
            }
            if (a.r) {
                const float4 rv = *reinterpret_cast<const float4*>(a.r + o);
                const float yv[4] = {fmaf(rv.x, sc.x, sh.x), fmaf(rv.y, sc.y, sh.y), fmaf(rv.z, sc.z, sh.z), fmaf(rv.w, sc.w, sh.w)};
                float gv[4] = {g.x, g.y, g.z, g.w};
                const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
                const float mm[4] = {mu.x, mu.y, mu.z, mu.w}, ss[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (a.act == GCPX_ACT_LRELU) gv[k] *= (yv[k] > 0.f ? 1.f : 0.2f);
                    s1[k] += gv[k];
                    s2[k] += gv[k] * (rr[k] - mm[k]) * ss[k];
                }
                g = make_float4(gv[0], gv[1], gv[2], gv[3]);
            }
            *reinterpret_cast<float4*>(a.dy + o) = g;
        }
        if (skip) *reinterpret_cast<float4*>(ds + ((((size_t)b * a.H + y) * a.W + x) * Cs) + c) = g;
    }
    if (a.stats_partial) {
        __shared__ float red[256][8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[threadIdx.x][k] = s1[k]; red[threadIdx.x][4 + k] = s2[k]; }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * a.C; i += 256) {
            const int which = i / a.C, ch = i % a.C;
            float s = 0.f;
            for (int t = ch / 4; t < 256; t += G) s += red[t][which * 4 + (ch & 3)];      // threads of channel group ch / 4 (previous-block half)
            a.stats_partial[((size_t)blockIdx.x * 2 + which) * a.C + ch] = s;
        }
    }
}

__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const float* __restrict__ partial, const int n_partial, const int C,
                                                              const double count, const float* __restrict__ gamma,
                                                              const float* __restrict__ rstd, float* __restrict__ coef,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              const int accumulate) {
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    int p = threadIdx.x;
    for (; p + 3 * 256 < n_partial; p += 4 * 256) {                // four rows' loads in flight per thread, summed in row order
        float a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = partial[((size_t)(p + u * 256) * 2 + 0) * C + c];
            b[u] = partial[((size_t)(p + u * 256) * 2 + 1) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s1 += a[u]; s2 += b[u]; }
    }
    for (; p < n_partial; p += 256) {
        s1 += partial[((size_t)p * 2 + 0) * C + c];
        s2 += partial[((size_t)p * 2 + 1) * C + c];
    }
    r1[threadIdx.x] = s1;
    r2[threadIdx.x] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { r1[threadIdx.x] += r1[threadIdx.x + s]; r2[threadIdx.x] += r2[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        s1 = r1[0]; s2 = r2[0];
        coef[c] = gamma[c] * rstd[c];
        coef[C + c] = (float)(s1 / count);
        coef[2 * C + c] = (float)(s2 / count);
        if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
        if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
    }
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(float* __restrict__ dy, const float* __restrict__ r,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ coef, const long long n4, const int C) {
    const int c = (threadIdx.x * 4) % C;
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), rs = *reinterpret_cast<const float4*>(rstd + c);
    const float4 k1 = *reinterpret_cast<const float4*>(coef + c), m1 = *reinterpret_cast<const float4*>(coef + C + c);
    const float4 m2 = *reinterpret_cast<const float4*>(coef + 2 * C + c);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 g = reinterpret_cast<float4*>(dy)[i];
        const float4 rv = reinterpret_cast<const float4*>(r)[i];
        g.x = k1.x * (g.x - m1.x - (rv.x - mu.x) * rs.x * m2.x);
        g.y = k1.y * (g.y - m1.y - (rv.y - mu.y) * rs.y * m2.y);
        g.z = k1.z * (g.z - m1.z - (rv.z - mu.z) * rs.z * m2.z);
        g.w = k1.w * (g.w - m1.w - (rv.w - mu.w) * rs.w * m2.w);
        reinterpret_cast<float4*>(dy)[i] = g;
    }
}

// out[f][Y][X][c] = (upsampled) concat of the normalised + activated sources
__global__ void __launch_bounds__(256) conv_stage_kernel(const gcpx_conv_args a) {
    // a thread keeps one 4-channel group for the whole launch (the grid stride is a multiple of Cin / 4, checked by the launcher):
    // its source, affine parameters and channel offset are loop invariants; pixel indices are 32-bit
    const int C4 = a.Cin / 4;
    const unsigned npix = (unsigned)a.F * a.Hout * a.Wout;
    const int c0 = a.src[0].C;
    const unsigned gtid = blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gtid % C4) * 4;
    const bool first = cg < c0;
    const gcpx_conv_src s = first ? a.src[0] : a.src[1];
    const int cl = first ? cg : cg - c0;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s.scale) { sc = *reinterpret_cast<const float4*>(s.scale + cl); sh = *reinterpret_cast<const float4*>(s.shift + cl); }
    const bool aff = s.scale != nullptr, lre = s.act == GCPX_ACT_LRELU;
    auto xf = [&](float4 t) {                              // affine_act4 with the parameters held in registers
        if (aff) { t.x = fmaf(t.x, sc.x, sh.x); t.y = fmaf(t.y, sc.y, sh.y); t.z = fmaf(t.z, sc.z, sh.z); t.w = fmaf(t.w, sc.w, sh.w); }
        if (lre) { t.x = lrelu(t.x, 0.2f); t.y = lrelu(t.y, 0.2f); t.z = lrelu(t.z, 0.2f); t.w = lrelu(t.w, 0.2f); }
        return t;
    };
    const unsigned pstride = gridDim.x * 256 / C4;
    const size_t plane = (size_t)a.Hin * a.Win * s.C;
    if (a.upsample) {
        // one low-resolution pixel per item: its 3 x 3 neighbourhood (clamped) makes the 2 x 2 output pixels it owns — 9 loads for 4 outputs
        // instead of 4 for 1 (the loads, not the stores, bounded the one-output form: 2.5 TB/s of output).  Same products and sums, in the same
        // order, as the one-output form below.
        const unsigned nlow = (unsigned)a.F * a.Hin * a.Win;
        for (unsigned p = gtid / C4; p < nlow; p += pstride) {
            const unsigned xl = p % a.Win, t_ = p / a.Win;
            const unsigned yl = t_ % a.Hin;
            int f = (int)(t_ / a.Hin);
            const int fo = f;
            float4 o[2][2];
#pragma unroll
            for (int ay = 0; ay < 2; ++ay)
#pragma unroll
                for (int ax = 0; ax < 2; ++ax) o[ay][ax] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.src_row_map) f = a.src_row_map[f];
            if (f >= 0) {
                const float* base = s.ptr + (size_t)(f / s.frame_div) * plane + cl;
                const int ry[3] = {max((int)yl - 1, 0), (int)yl, min((int)yl + 1, a.Hin - 1)};
                const int rx[3] = {max((int)xl - 1, 0), (int)xl, min((int)xl + 1, a.Win - 1)};
                float4 v[3][3];
#pragma unroll
                for (int iy = 0; iy < 3; ++iy)
#pragma unroll
                    for (int ix = 0; ix < 3; ++ix)
                        v[iy][ix] = xf(*reinterpret_cast<const float4*>(base + ((unsigned)ry[iy] * a.Win + rx[ix]) * s.C));
#pragma unroll
                for (int ay = 0; ay < 2; ++ay) {
                    const float wy1 = ay ? 0.25f : 0.75f, wy0 = 1.f - wy1;        // rows (ay, ay + 1) of the neighbourhood
#pragma unroll
                    for (int ax = 0; ax < 2; ++ax) {
                        const float wx1 = ax ? 0.25f : 0.75f, wx0 = 1.f - wx1;
                        const float4 v00 = v[ay][ax], v01 = v[ay][ax + 1], v10 = v[ay + 1][ax], v11 = v[ay + 1][ax + 1];
                        float4 r_;
                        r_.x = wy0 * (wx0 * v00.x + wx1 * v01.x) + wy1 * (wx0 * v10.x + wx1 * v11.x);
                        r_.y = wy0 * (wx0 * v00.y + wx1 * v01.y) + wy1 * (wx0 * v10.y + wx1 * v11.y);
                        r_.z = wy0 * (wx0 * v00.z + wx1 * v01.z) + wy1 * (wx0 * v10.z + wx1 * v11.z);
                        r_.w = wy0 * (wx0 * v00.w + wx1 * v01.w) + wy1 * (wx0 * v10.w + wx1 * v11.w);
                        o[ay][ax] = r_;
                    }
                }
            }
            float4* op = reinterpret_cast<float4*>(a.out) + (((size_t)fo * a.Hout + 2 * yl) * a.Wout + 2 * xl) * C4 + cg / 4;
            op[0] = o[0][0];
            op[C4] = o[0][1];
            op[(size_t)a.Wout * C4] = o[1][0];
            op[(size_t)a.Wout * C4 + C4] = o[1][1];
        }
        return;
    }
    for (unsigned p = gtid / C4; p < npix; p += pstride) {
        const unsigned X = p % a.Wout, t_ = p / a.Wout;
        const unsigned Y = t_ % a.Hout;
        int f = (int)(t_ / a.Hout);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.src_row_map) f = a.src_row_map[f];
        if (f >= 0) {
            const float* base = s.ptr + (size_t)(f / s.frame_div) * plane + cl;
            auto ld = [&](int y, int x) { return xf(*reinterpret_cast<const float4*>(base + ((unsigned)y * a.Win + x) * s.C)); };
            if (a.upsample) {
                // align_corners=False x2: Y even -> rows (Y/2 - 1, Y/2) weights (.25, .75); odd -> (Y/2, Y/2 + 1) weights (.75, .25)
                const int y0 = (Y & 1) ? Y / 2 : (int)(Y / 2) - 1, x0 = (X & 1) ? X / 2 : (int)(X / 2) - 1;
                const float wy1 = (Y & 1) ? 0.25f : 0.75f, wx1 = (X & 1) ? 0.25f : 0.75f;
                const int ya = max(y0, 0), yb = min(y0 + 1, a.Hin - 1), xa = max(x0, 0), xb = min(x0 + 1, a.Win - 1);
                const float4 v00 = ld(ya, xa), v01 = ld(ya, xb), v10 = ld(yb, xa), v11 = ld(yb, xb);
                const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
                v.x = wy0 * (wx0 * v00.x + wx1 * v01.x) + wy1 * (wx0 * v10.x + wx1 * v11.x);
                v.y = wy0 * (wx0 * v00.y + wx1 * v01.y) + wy1 * (wx0 * v10.y + wx1 * v11.y);
                v.z = wy0 * (wx0 * v00.z + wx1 * v01.z) + wy1 * (wx0 * v10.z + wx1 * v11.z);
                v.w = wy0 * (wx0 * v00.w + wx1 * v01.w) + wy1 * (wx0 * v10.w + wx1 * v11.w);
            } else {
                v = ld(Y, X);
            }
        }
        reinterpret_cast<float4*>(a.out)[(size_t)p * C4 + cg / 4] = v;
    }
}

__global__ void __launch_bounds__(256) col2im4x4s2_kernel(const float* __restrict__ dcol, float* __restrict__ dx, const int F,
                                                          const int H, const int W, const int Cin) {
    const int C4 = Cin / 4;
    const long long total = (long long)F * H * W * C4;
    const int Ho = H / 2, Wo = W / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        long long p = i / C4;
        const int ix = (int)(p % W); p /= W;
        const int iy = (int)(p % H);
        const int f = (int)(p / H);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ky = ((iy + 1) & 1) + 2 * a;
            const int oy = (iy + 1 - ky) / 2;
            if (iy + 1 - ky < 0 || oy >= Ho) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int kx = ((ix + 1) & 1) + 2 * b;
                const int ox = (ix + 1 - kx) / 2;
                if (ix + 1 - kx < 0 || ox >= Wo) continue;
                const float4 v = *reinterpret_cast<const float4*>(dcol + (((size_t)f * Ho + oy) * Wo + ox) * 16 * Cin +
                                                                  (ky * 4 + kx) * Cin + c);
                g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
            }
        }
        reinterpret_cast<float4*>(dx)[i] = g;
    }
}

__global__ void __launch_bounds__(256) im2col_image_kernel(const float* __restrict__ x, float* __restrict__ col, const int F,
                                                           const int H, const int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)F * Ho * Wo * 48;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int k = (int)(i % 48);
        long long p = i / 48;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho);
        const int f = (int)(p / Ho);
        const int ci = k / 16, ky = (k / 4) & 3, kx = k & 3;
        const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)f * 3 + ci) * H + iy) * W + ix];
        col[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// loss gradients
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) loss_heads_bwd_kernel(const gcpx_loss_args a, float* __restrict__ dlen,
                                                             float* __restrict__ dexist, float* __restrict__ dstate) {
    const int B = a.B, T = a.T, N = a.N;
    const float inv_div = 1.f / a.total_div;
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    if (dlen && a.len_logits) {
        const int ldl = (T + 15) & ~15;              // rows padded to the GEMM K granularity, pad columns zero
        for (int i = tid; i < B * ldl; i += nth) {
            const int b = i / ldl, t = i % ldl;
            float v = 0.f;
            if (t < T) {
                const float* lg = a.len_logits + (size_t)b * T;
                float mx = lg[0];
                for (int k = 1; k < T; ++k) mx = fmaxf(mx, lg[k]);
                float se = 0.f;
                for (int k = 0; k < T; ++k) se += expf(lg[k] - mx);
                const float p = expf(lg[t] - mx) / se;
                v = (p - (t == (int)a.end_ind[b] ? 1.f : 0.f)) * a.w_len * inv_div / B;
            }
            dlen[i] = v;
        }
    }
    if (dexist && a.existence) {
        for (int i = tid; i < B * N * 16; i += nth) {
            const int r = i / 16, c = i % 16;
            float v = 0.f;
            if (c == 0) v = (sigmoid_acc(a.existence[r]) - (a.leave[r] ? 1.f : 0.f)) * a.w_exist * inv_div / (float)(B * N);
            dexist[i] = v;
        }
    }
    if (dstate && a.regressed_state && a.state_target) {
        int rl = 0;
        for (int b = 0; b < B; ++b) rl = max(rl, a.seq_len[b]);
        const float k = 2.f * a.w_state * inv_div / (float)(B * rl * a.state_dim);
        for (int i = tid; i < B * T * 16; i += nth) {
            const int r = i / 16, c = i % 16;
            const int t = r % T;
            float v = 0.f;
            if (c < a.state_dim && t < rl) {
                const float pm = (a.state_mask ? a.state_mask : a.pad_mask)[r];
                v = k * pm * (a.regressed_state[(size_t)r * a.state_dim + c] - a.state_target[(size_t)r * a.state_dim + c]);
            }
            dstate[i] = v;
        }
    }
}

// d total / d outputs of the inverse-model and cost-model heads (L2 means, inverse_mdl.py:181-191, cost_mdl.py:59-62)
__global__ void __launch_bounds__(256) loss_aux_heads_bwd_kernel(const gcpx_loss_args a, float* __restrict__ daction,
                                                                 float* __restrict__ dcost) {
    const int B = a.B, T = a.T;
    const float inv_div = 1.f / a.total_div;
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    if (daction && a.action_pred) {
        const int na = a.n_actions;
        const float k = 2.f * a.w_action * inv_div / (float)(B * na);
        for (int i = tid; i < B * 16; i += nth) {
            const int b = i / 16, c = i % 16;
            float v = 0.f;
            if (c < na) v = k * (a.action_pred[b * na + c] - a.action_seq[((size_t)b * (T - 1) + (int)a.inv_t0[b]) * na + c]);
            daction[i] = v;
        }
    }
    if (dcost && a.cost_pred) {
        const float k = 2.f * a.w_cost * inv_div / (float)B;
        for (int i = tid; i < B * 16; i += nth) {
            const int b = i / 16, c = i % 16;
            dcost[i] = c == 0 ? k * (a.cost_pred[b] - a.cost_target[b]) : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// parameters
// ---------------------------------------------------------------------------------------------------
// four consecutive destination elements per thread (arena leaves are padded to multiples of 4): 16-byte index loads and stores
__global__ void __launch_bounds__(256) repack_kernel(const float* __restrict__ theta, const int* __restrict__ idx0,
                                                     const int* __restrict__ idx1, float* __restrict__ dst, const long long n) {
    const long long n4 = n >> 2, stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (!idx1) {
        // (two index -> value chains in flight: a launch held to a few workgroups — gcpx_repack_blocks — is bound by their latency)
        for (; i + stride < n4; i += 2 * stride) {
            const int4 a = reinterpret_cast<const int4*>(idx0)[i], b = reinterpret_cast<const int4*>(idx0)[i + stride];
            const float4 va = make_float4(a.x >= 0 ? theta[a.x] : 0.f, a.y >= 0 ? theta[a.y] : 0.f, a.z >= 0 ? theta[a.z] : 0.f,
                                          a.w >= 0 ? theta[a.w] : 0.f);
            const float4 vb = make_float4(b.x >= 0 ? theta[b.x] : 0.f, b.y >= 0 ? theta[b.y] : 0.f, b.z >= 0 ? theta[b.z] : 0.f,
                                          b.w >= 0 ? theta[b.w] : 0.f);
            reinterpret_cast<float4*>(dst)[i] = va;
            reinterpret_cast<float4*>(dst)[i + stride] = vb;
        }
    }
    for (; i < n4; i += stride) {
        const int4 a = reinterpret_cast<const int4*>(idx0)[i];
        float4 v = make_float4(a.x >= 0 ? theta[a.x] : 0.f, a.y >= 0 ? theta[a.y] : 0.f, a.z >= 0 ? theta[a.z] : 0.f,
                               a.w >= 0 ? theta[a.w] : 0.f);
        if (idx1) {
            const int4 b = reinterpret_cast<const int4*>(idx1)[i];
            if (b.x >= 0) v.x += theta[b.x];
            if (b.y >= 0) v.y += theta[b.y];
            if (b.z >= 0) v.z += theta[b.z];
            if (b.w >= 0) v.w += theta[b.w];
        }
        reinterpret_cast<float4*>(dst)[i] = v;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int a = idx0[i];
        float v = a >= 0 ? theta[a] : 0.f;
        if (idx1 && idx1[i] >= 0) v += theta[idx1[i]];
        dst[i] = v;
    }
}

int blocks_for(long long items, int cap = 4096) {
    long long b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

#define STREAM() hipStream_t stream = reinterpret_cast<hipStream_t>(stream_)

extern "C" int gcpx_lstm_bwd(const gcpx_lstm_bwd_args* a, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(a && a->gates && a->c_prev && a->c_new && a->dgates && a->dc_prev, "missing pointer");
    GCPX_CHECK_ARG(a->M > 0 && a->H > 0 && a->rpb > 0, "bad sizes");
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3((a->M * a->H + 255) / 256), dim3(256), 0, stream, *a);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_gn_bwd_blocks(int32_t M) { return (M + GN_ROWS_PER_BLOCK - 1) / GN_ROWS_PER_BLOCK; }

extern "C" int gcpx_gn_lrelu_bwd(const float* u, const float* da, const float* gamma, const float* beta, float* du, float* partial,
                                 int32_t M, int32_t C, int32_t groups, float eps, float slope, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(u && da && gamma && beta && du && partial && M > 0, "missing pointer");
    GCPX_CHECK_ARG(groups > 0 && C % groups == 0 && (C / groups) <= 16 && ((C / groups) & (C / groups - 1)) == 0,
                   "channels per group must be a power of two <= 16");
    const int nb = gcpx_gn_bwd_blocks(M);
    if (C == 128) hipLaunchKernelGGL(gn_lrelu_bwd_kernel<128>, dim3(nb), dim3(256), 0, stream, u, da, gamma, beta, du, partial, M, groups, eps, slope);
    else if (C == 32) hipLaunchKernelGGL(gn_lrelu_bwd_kernel<32>, dim3(nb), dim3(256), 0, stream, u, da, gamma, beta, du, partial, M, groups, eps, slope);
    else { gcpx_set_error("gcpx_gn_lrelu_bwd: unsupported C=%d (128 or 32)", C); return GCPX_ERR_UNSUPPORTED; }
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_lrelu_bwd(const float* a, const float* dy, float* dx, int64_t n, float slope, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(a && dy && dx && n > 0, "bad arguments");
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, dy, dx, (long long)n, slope);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_reduce_partials(const float* partial, int32_t n, int64_t stride, int32_t len, float* dst, int32_t accumulate,
                                    void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(partial && dst && n > 0 && len > 0, "bad arguments");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((len + 15) / 16), dim3(256), 0, stream, partial, n, (long long)stride, len, dst,
                       accumulate);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_kl_bwd(const float* qz, const float* pz, float* dqz, float* dpz, int32_t B, int32_t N, int32_t nz,
                           int64_t batch_stride, int64_t node_stride, float free_nats, float coef, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(qz && pz && dqz && dpz && B > 0 && N > 0 && nz > 0, "bad arguments");
    const int total = B * N * nz;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, qz, pz, dqz, dpz, N, nz,
                       (long long)batch_stride, (long long)node_stride, free_nats, coef, total, (const float*)nullptr, 0ll, (const float*)nullptr);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_kl_bwd_weighted(const float* qz, const float* pz, float* dqz, float* dpz, int32_t B, int32_t N, int32_t nz,
                                    int64_t batch_stride, int64_t node_stride, float free_nats, float coef, const float* node_weight,
                                    int64_t weight_bstride, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(qz && pz && dqz && dpz && node_weight && B > 0 && N > 0 && nz > 0, "bad arguments");
    const int total = B * N * nz;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, qz, pz, dqz, dpz, N, nz,
                       (long long)batch_stride, (long long)node_stride, free_nats, coef, total, node_weight, (long long)weight_bstride,
                       (const float*)nullptr);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_kl_bwd_scheduled(const float* qz, const float* pz, float* dqz, float* dpz, int32_t B, int32_t N, int32_t nz,
                                     int64_t batch_stride, int64_t node_stride, float free_nats, float coef, const float* node_weight,
                                     int64_t weight_bstride, const float* coef_dev, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(qz && pz && dqz && dpz && coef_dev && B > 0 && N > 0 && nz > 0, "bad arguments");
    const int total = B * N * nz;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, qz, pz, dqz, dpz, N, nz,
                       (long long)batch_stride, (long long)node_stride, free_nats, coef, total, node_weight, (long long)weight_bstride, coef_dev);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_zero_unmapped_rows(float* ptr, int64_t row_floats, const int32_t* row2frame, int32_t rows, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(ptr && row2frame && rows > 0 && row_floats > 0 && row_floats % 4 == 0, "bad arguments");
    hipLaunchKernelGGL(zero_unmapped_rows_kernel, dim3(64, rows), dim3(256), 0, stream, reinterpret_cast<float4*>(ptr), (long long)(row_floats / 4),
                       row2frame);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_rows_strided(float* dst, int64_t dst_sb, int64_t dst_sr, const float* src, int64_t src_sb, int64_t src_sr, int32_t B,
                                 int32_t rpb, int32_t width, int32_t mode, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(dst && src && B > 0 && rpb > 0 && width > 0 && mode >= 0 && mode <= 2, "bad arguments");
    const int total = (mode == 2 ? B : B * rpb) * width;
    hipLaunchKernelGGL(rows_strided_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, dst, (long long)dst_sb, (long long)dst_sr, src,
                       (long long)src_sb, (long long)src_sr, rpb, width, mode, total);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_latent_bwd(const float* dqz_pos, const float* dpz_pos, const float* qz_pos, int64_t pb, int64_t prow,
                               const float* eps, int64_t eb, int64_t erow, const float* dz0, int64_t ldz0, const float* dz1,
                               int64_t ldz1, float* dq_out, float* dp_out, int32_t M, int32_t rpb, int32_t nz, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(dqz_pos && dpz_pos && qz_pos && eps && dz0 && dq_out && dp_out && M > 0 && rpb > 0 && nz > 0, "bad arguments");
    hipLaunchKernelGGL(latent_bwd_kernel, dim3((M * nz + 255) / 256), dim3(256), 0, stream, dqz_pos, dpz_pos, qz_pos, (long long)pb,
                       (long long)prow, eps, (long long)eb, (long long)erow, dz0, (long long)ldz0, dz1, (long long)ldz1, dq_out,
                       dp_out, M, rpb, nz);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_tree_accum(const gcpx_tree_accum_args* a, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(a && a->dst && a->nsrc >= 1 && a->nsrc <= 6 && a->B > 0 && a->n > 0 && a->width > 0, "bad arguments");
    hipLaunchKernelGGL(tree_accum_kernel, dim3((a->width + 63) / 64, a->n + 1, a->B), dim3(64), 0, stream, *a);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_timestep_scatter(const float* det, int64_t db, int64_t dp, const int32_t* node_t, float* out, int32_t B,
                                     int32_t N, int32_t T, int32_t nz, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(det && node_t && out && B > 0 && N > 0 && T > 0 && nz > 0, "bad arguments");
    hipLaunchKernelGGL(timestep_scatter_kernel, dim3(T, B), dim3(128), 0, stream, det, (long long)db, (long long)dp, node_t, out, N, T, nz);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_add_rows(float* dst, int64_t dst_sb, int64_t dst_sr, const float* src1, const float* src2, int32_t B,
                             int32_t rpb, int32_t width, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(dst && src1 && B > 0 && rpb > 0 && width > 0, "bad arguments");
    const int total = B * rpb * width;
    hipLaunchKernelGGL(add_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, dst, (long long)dst_sb, (long long)dst_sr,
                       src1, src2, rpb, width, total);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_index_offset(const int32_t* idx, int32_t* out, int32_t B, int32_t T, int32_t stride, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(idx && out && B > 0 && T > 0, "bad arguments");
    hipLaunchKernelGGL(index_offset_kernel, dim3((B * T + 255) / 256), dim3(256), 0, stream, idx, out, T, stride, B * T);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_index_inverse(const int32_t* fwd, int32_t n, int32_t* inv, int32_t n_inv, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(fwd && inv && n > 0 && n_inv > 0, "bad arguments");
    hipLaunchKernelGGL(index_fill_kernel, dim3((n_inv + 255) / 256), dim3(256), 0, stream, inv, n_inv, -1);
    hipLaunchKernelGGL(index_inverse_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fwd, n, inv, n_inv);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_act_bwd_blocks(void) { return ACT_BLOCKS; }

extern "C" int gcpx_act_bwd(const gcpx_actbwd_args* a, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(a && a->da && a->dy, "missing pointer");
    GCPX_CHECK_ARG(a->F > 0 && a->H > 0 && a->W > 0 && a->C >= 4 && (a->C & (a->C - 1)) == 0 && a->C <= 1024, "C must be a power of two in [4, 1024]");
    GCPX_CHECK_ARG(a->fsum >= 1 && a->ldc % 4 == 0 && a->c_off % 4 == 0, "bad fsum / ldc / c_off");
    GCPX_CHECK_ARG(!a->stats_partial || (a->mean && a->rstd && a->r), "stats need r, mean, rstd");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ACT_BLOCKS), dim3(256), 0, stream, *a);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_act_skip_bwd(const gcpx_actbwd_args* a, float* ds, int32_t c_off_s, int32_t Cs, int32_t rpb, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(a && a->da && a->dy && ds, "missing pointer");
    GCPX_CHECK_ARG(a->up == 1 && a->fsum == 1 && rpb >= 1 && a->F % rpb == 0, "the fused form is the upsampled one, one output frame per input frame, whole sequences");
    GCPX_CHECK_ARG(a->C >= 4 && Cs >= 4 && a->C % 4 == 0 && Cs % 4 == 0 && 256 % ((a->C + Cs) / 4) == 0, "channel groups must divide 256");
    GCPX_CHECK_ARG(a->ldc % 4 == 0 && a->c_off % 4 == 0 && c_off_s % 4 == 0, "bad ldc / c_off");
    GCPX_CHECK_ARG(!a->stats_partial || (a->mean && a->rstd && a->r), "stats need r, mean, rstd");
    hipLaunchKernelGGL(act_skip_bwd_kernel, dim3(ACT_BLOCKS), dim3(256), 0, stream, *a, ds, c_off_s, Cs, rpb);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_tanh_bwd_rows(const float* dy, const float* y, float* dx, int64_t sb, int64_t sr, int32_t B, int32_t rpb, int32_t width,
                                  void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(dy && y && dx && B > 0 && rpb > 0 && width > 0, "bad arguments");
    const long long total = (long long)B * rpb * width;
    hipLaunchKernelGGL(tanh_bwd_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, dy, y, dx, (long long)sb, (long long)sr,
                       rpb, width, total);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_bn_bwd_finalize(const float* partial, int32_t n_partial, int32_t C, double count, const float* gamma,
                                    const float* rstd, float* coef, float* dgamma, float* dbeta, int32_t accumulate, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(partial && gamma && rstd && coef && n_partial > 0 && C > 0 && count > 0, "bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, stream, partial, n_partial, C, count, gamma, rstd,
                       coef, dgamma, dbeta, accumulate);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_bn_bwd_apply(float* dy, const float* r, const float* mean, const float* rstd, const float* coef, int64_t n,
                                 int32_t C, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(dy && r && mean && rstd && coef && n > 0 && n % 4 == 0, "bad arguments");
    GCPX_CHECK_ARG(C >= 4 && (C & (C - 1)) == 0 && C <= 1024, "C must be a power of two in [4, 1024]");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks_for(n / 4)), dim3(256), 0, stream, dy, r, mean, rstd, coef, (long long)(n / 4), C);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_conv_stage(const gcpx_conv_args* a, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(a && a->out && a->nsrc >= 1 && a->nsrc <= 2 && a->F > 0, "bad arguments");
    GCPX_CHECK_ARG(a->Cin % 4 == 0 && a->src[0].C % 4 == 0, "channels % 4");
    GCPX_CHECK_ARG(a->Cin == a->src[0].C + (a->nsrc == 2 ? a->src[1].C : 0), "Cin != sum of sources");
    if (a->upsample) GCPX_CHECK_ARG(a->Hout == 2 * a->Hin && a->Wout == 2 * a->Win, "upsample: Hout != 2*Hin");
    else GCPX_CHECK_ARG(a->Hout == a->Hin && a->Wout == a->Win, "no upsample: Hout != Hin");
    const long long items = (long long)a->F * a->Hout * a->Wout * (a->Cin / 4);
    GCPX_CHECK_ARG(256 % (a->Cin / 4) == 0 && (long long)a->F * a->Hout * a->Wout < (1LL << 31), "Cin / 4 must divide 256; < 2^31 pixels");
    hipLaunchKernelGGL(conv_stage_kernel, dim3(blocks_for(items, 16384)), dim3(256), 0, stream, *a);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_col2im4x4s2(const float* dcol, float* dx, int32_t F, int32_t H, int32_t W, int32_t Cin, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(dcol && dx && F > 0 && H % 2 == 0 && W % 2 == 0 && Cin % 4 == 0, "bad arguments");
    hipLaunchKernelGGL(col2im4x4s2_kernel, dim3(blocks_for((long long)F * H * W * (Cin / 4), 16384)), dim3(256), 0, stream, dcol, dx, F, H, W, Cin);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_im2col_image(const float* x, float* col, int32_t F, int32_t H, int32_t W, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(x && col && F > 0 && H % 2 == 0 && W % 2 == 0, "bad arguments");
    hipLaunchKernelGGL(im2col_image_kernel, dim3(blocks_for((long long)F * (H / 2) * (W / 2) * 48, 16384)), dim3(256), 0, stream, x, col, F, H, W);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_loss_heads_bwd(const gcpx_loss_args* a, float* dlen, float* dexist, float* dstate, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(a && a->seq_len && a->end_ind && a->pad_mask, "missing pointer");
    hipLaunchKernelGGL(loss_heads_bwd_kernel, dim3(64), dim3(256), 0, stream, *a, dlen, dexist, dstate);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_loss_aux_heads_bwd(const gcpx_loss_args* a, float* daction, float* dcost, void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(a && a->B > 0 && a->total_div > 0, "bad arguments");
    GCPX_CHECK_ARG(!(daction && a->action_pred) || (a->action_seq && a->inv_t0 && a->n_actions > 0 && a->n_actions <= 16),
                   "inverse-model head needs action_seq, inv_t0 and 1..16 actions");
    GCPX_CHECK_ARG(!(dcost && a->cost_pred) || a->cost_target, "cost-model head needs cost_target");
    hipLaunchKernelGGL(loss_aux_heads_bwd_kernel, dim3(4), dim3(256), 0, stream, *a, daction, dcost);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_repack_blocks(const float* theta, const int32_t* idx0, const int32_t* idx1, float* dst, int64_t n, int32_t max_blocks,
                                  void* stream_) {
    STREAM();
    GCPX_CHECK_ARG(theta && idx0 && dst && n > 0 && max_blocks >= 0, "bad arguments");
    GCPX_CHECK_ARG((((uintptr_t)idx0 | (uintptr_t)idx1 | (uintptr_t)dst) & 15) == 0, "idx0 / idx1 / dst must be 16-byte aligned");
    int nb = blocks_for((n + 3) / 4, 16384);
    if (max_blocks > 0) nb = std::min(nb, max_blocks);
    hipLaunchKernelGGL(repack_kernel, dim3(nb), dim3(256), 0, stream, theta, idx0, idx1, dst, (long long)n);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_repack(const float* theta, const int32_t* idx0, const int32_t* idx1, float* dst, int64_t n, void* stream_) {
    return gcpx_repack_blocks(theta, idx0, idx1, dst, n, 0, stream_);
}

