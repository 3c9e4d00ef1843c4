// Weight gradients on gfx950 f32 MFMA:   dW[n][k] = sum_r dY[r][n] * X[r][k]      ("TN" GEMM, reduction over rows)
//
// Backward of every Linear / LSTMCell / Conv1d / Conv2d of the gcp_tree training step
// (/root/reference/gcp/prediction/train.py:155-163 `losses.total.value.backward()`; the reference relies on torch
// autograd, this build has no autograd: each wgrad is an explicit launch of this kernel).
//
// Both operands are row-major with the reduction index r as the slow dimension, so the MFMA k index (4 rows per
// step) is the strided one and the i / j sides are contiguous in memory.  A lane loads 16 B = 4 consecutive columns
// of its row for each operand and treats them as 4 different MFMA tiles (tile t holds columns 4*i + t): one pair of
// float4 loads feeds 16 MFMAs and a wavefront owns a 64 x 64 block of dW.  X can be addressed as plain rows (with
// gather / shift, like gcpx_row_src) or as the implicit im2col of a conv1d(3) / conv3x3(pad 1) / conv4x4(stride 2,
// pad 1) input in NHWC, optionally with the producer's BatchNorm affine + LeakyReLU applied on load.
// The row range can be split over blockIdx.z; partial results are combined by gcpx_wgrad_reduce in a fixed order
// (deterministic), which also maps (n, k) to the canonical torch parameter layout.
#include "common.h"

#include <cstdlib>

// split-f16 form of the direct-mode row problems with whole 128 x 128 blocks (wgrad_rows_split.hip)
bool gcpx_wgrad_rows_split_applies(const gcpx_wgrad_args* a);
int gcpx_wgrad_rows_split_blocks(const gcpx_wgrad_args* a);
int gcpx_launch_wgrad_rows_split(const gcpx_wgrad_args* a, hipStream_t stream);
int gcpx_launch_wgrad_rows_split_group(const gcpx_wgrad_args* tab, const int32_t* block_start, int nprob, int total_blocks, hipStream_t stream);

namespace {

__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }

// RS (row split inside the workgroup): small dW (few 64 x 64 tiles) with many rows would leave the chip idle and every
// wavefront in a long latency-bound loop; there the 4 wavefronts of a workgroup share ONE tile, take every 4th row chunk
// and are combined through LDS in a fixed order (deterministic), instead of owning 4 different K tiles.
// (bx, by, bz, gz) = the workgroup's position in the problem's own grid: blockIdx / gridDim.z of a single-problem launch, or derived
// from the block table of a grouped launch (gcpx_wgrad_group)
template <int TA, bool RS>
__device__ __forceinline__ void wgrad_body(const gcpx_wgrad_args& a, const int bx, const int by, const int bz, const int gz) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ij = lane & 15, kk = lane >> 4;
    const int k0 = RS ? bx * 64 : (bx * 4 + wave) * 64;
    if (!RS && k0 >= a.K) return;                // no barriers below (non-RS): whole wavefronts may leave
    const int n0 = by * (TA == 4 ? 64 : 16);
    // bz = row split (partial mode) or batch index (nbatch > 1: independent problems with strided pointers)
    const int zb = a.nbatch > 1 ? bz : 0;
    const int nsplit = a.nbatch > 1 ? 1 : gz;
    const int zs = a.nbatch > 1 ? 0 : bz;
    const int rows_per = (((a.R + nsplit - 1) / nsplit) + 3) & ~3;
    const int r_begin = zs * rows_per;
    const int r_end = min(a.R, r_begin + rows_per);
    const float* __restrict__ dyp = a.dy + (size_t)zb * a.z_dy_off;
    const float* __restrict__ xpb = a.x + (size_t)zb * a.z_x_off;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);          // bias gradient: column sums of dy (wavefronts with k0 == 0)

    // ---- per-lane constants of the X (B operand) column group ----
    const int kcol = k0 + 4 * ij;
    const bool kvalid = kcol < a.K;
    int tap = 0, ci = kcol;
    if (a.mode != GCPX_WG_ROWS) {
        tap = kvalid ? kcol / a.Cin : 0;
        ci = kcol - tap * a.Cin;
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool xf = a.scale != nullptr;
    if (xf && kvalid) {
        const int c = (a.mode == GCPX_WG_ROWS) ? (kcol & (a.cmod - 1)) : ci;
        sc = *reinterpret_cast<const float4*>(a.scale + c);
        sh = *reinterpret_cast<const float4*>(a.shiftv + c);
    }
    const int act = a.act;
    const int lw = ilog2(a.W > 0 ? a.W : 1), lh = ilog2(a.H > 0 ? a.H : 1);
    int dyo = 0, dxo = 0;
    if (a.mode == GCPX_WG_CONV3X3) { dyo = tap / 3 - 1; dxo = tap % 3 - 1; }
    if (a.mode == GCPX_WG_CONV4X4S2) { dyo = tap / 4 - 1; dxo = tap % 4 - 1; }
    const int shift = (a.mode == GCPX_WG_CONV1D) ? tap - 1 : a.shift;
    const bool dense = (a.mode == GCPX_WG_ROWS) && a.rowidx == nullptr && a.shift == 0 && a.sb == a.sr * a.rpb;

    // ---- A operand column(s) ----
    const int ncol = n0 + (TA == 4 ? 4 * ij : ij);
    const bool nvalid = ncol < a.N;

    f32x4 acc[TA][4];
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = f32x4{0, 0, 0, 0};

    constexpr int UNR = 2;
    auto load = [&](const int r0, float4 (&av)[UNR], float4 (&bv)[UNR], bool (&bok)[UNR]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int row = r0 + 4 * u + kk;
            const bool rvalid = row < r_end;
            av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            bv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            bok[u] = false;
            if (rvalid && nvalid) {
                const float* ap;
                if (a.dy_sb) {
                    const int b = row / a.dy_rpb;
                    ap = dyp + (size_t)b * a.dy_sb + (size_t)(row - b * a.dy_rpb) * a.ldy + ncol;
                } else {
                    ap = dyp + (size_t)row * a.ldy + ncol;
                }
                if (TA == 4) av[u] = *reinterpret_cast<const float4*>(ap);
                else av[u].x = *ap;
            }
            if (rvalid && kvalid) {
                const float* xp = nullptr;
                if (a.mode == GCPX_WG_ROWS || a.mode == GCPX_WG_CONV1D) {
                    if (dense) {
                        xp = xpb + (size_t)row * a.sr + ci;
                    } else if (a.rowidx) {
                        xp = xpb + (size_t)a.rowidx[row] * a.sr + ci;
                    } else {
                        const int b = row / a.rpb, j = row - b * a.rpb + shift;
                        if (j >= 0 && j < a.rpb) xp = xpb + (size_t)b * a.sb + (size_t)j * a.sr + ci;
                    }
                } else if (a.mode == GCPX_WG_CONV3X3) {
                    const int x = row & (a.W - 1), y = (row >> lw) & (a.H - 1);
                    int f = row >> (lw + lh);
                    if (a.frame_map) f = a.frame_map[f];
                    const int iy = y + dyo, ix = x + dxo;
                    if (f >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                        xp = a.x + (((size_t)f * a.H + iy) * a.W + ix) * a.Cin + ci;
                } else {   // 4x4 stride 2 pad 1: rows are output pixels (H/2 x W/2), X is the H x W input
                    const int ox = row & ((a.W >> 1) - 1), oy = (row >> (lw - 1)) & ((a.H >> 1) - 1);
                    int f = row >> (lw + lh - 2);
                    if (a.frame_map) f = a.frame_map[f];
                    const int iy = 2 * oy + dyo, ix = 2 * ox + dxo;
                    if (f >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                        xp = a.x + (((size_t)f * a.H + iy) * a.W + ix) * a.Cin + ci;
                }
                if (xp) { bv[u] = *reinterpret_cast<const float4*>(xp); bok[u] = true; }
            }
        }
    };
    // software pipeline with two register sets in ping-pong: the loads of the next row chunk are in flight while the current
    // chunk's 32 MFMAs run (no register copies — a copy lets the scheduler pull the wait for the next chunk in front of them)
    const int rstep = (RS ? 4 : 1) * 4 * UNR;
    auto mm = [&](const float4 (&av)[UNR], const float4 (&bv)[UNR], const bool (&bok)[UNR]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float4 b = bv[u];
            if (bok[u] && (xf || act)) {
                b.x = fmaf(b.x, sc.x, sh.x); b.y = fmaf(b.y, sc.y, sh.y); b.z = fmaf(b.z, sc.z, sh.z); b.w = fmaf(b.w, sc.w, sh.w);
                if (act == GCPX_ACT_LRELU) { b.x = lrelu(b.x, 0.2f); b.y = lrelu(b.y, 0.2f); b.z = lrelu(b.z, 0.2f); b.w = lrelu(b.w, 0.2f); }
            }
            bsum.x += av[u].x; bsum.y += av[u].y; bsum.z += av[u].z; bsum.w += av[u].w;
            const float aa[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
            const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int ta = 0; ta < TA; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = mfma16(aa[ta], bb[tb], acc[ta][tb]);
        }
    };
    float4 avA[UNR], bvA[UNR], avB[UNR], bvB[UNR];
    bool bokA[UNR], bokB[UNR];
    int r0 = r_begin + (RS ? wave * 4 * UNR : 0);
    if (r0 < r_end) load(r0, avA, bvA, bokA);
    for (; r0 < r_end; r0 += 2 * rstep) {
        const bool more = r0 + rstep < r_end;
        if (more) load(r0 + rstep, avB, bvB, bokB);
        mm(avA, bvA, bokA);
        if (r0 + 2 * rstep < r_end) load(r0 + 2 * rstep, avA, bvA, bokA);
        if (more) mm(avB, bvB, bokB);
    }

    if constexpr (RS) {
        // combine the 4 row classes in wavefront order (fixed -> deterministic): waves 1..3 park their accumulators in LDS
        __shared__ float4 racc[3][TA * 4 + 1][64];
        if (wave > 0) {
#pragma unroll
            for (int ta = 0; ta < TA; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
                    racc[wave - 1][ta * 4 + tb][lane] = make_float4(acc[ta][tb][0], acc[ta][tb][1], acc[ta][tb][2], acc[ta][tb][3]);
            racc[wave - 1][TA * 4][lane] = bsum;
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w) {
#pragma unroll
            for (int ta = 0; ta < TA; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb) {
                    const float4 v = racc[w][ta * 4 + tb][lane];
                    acc[ta][tb][0] += v.x; acc[ta][tb][1] += v.y; acc[ta][tb][2] += v.z; acc[ta][tb][3] += v.w;
                }
            const float4 v = racc[w][TA * 4][lane];
            bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
        }
    }

    // ---- bias gradient (direct mode): sum the 4 row classes kk, lane (ij, kk == 0) owns columns ncol .. ----
    if (a.dbias && k0 == 0 && !a.partial) {
        bsum.x += __shfl_xor(bsum.x, 16); bsum.y += __shfl_xor(bsum.y, 16); bsum.z += __shfl_xor(bsum.z, 16); bsum.w += __shfl_xor(bsum.w, 16);
        bsum.x += __shfl_xor(bsum.x, 32); bsum.y += __shfl_xor(bsum.y, 32); bsum.z += __shfl_xor(bsum.z, 32); bsum.w += __shfl_xor(bsum.w, 32);
        if (kk == 0) {
            const float bs[4] = {bsum.x, bsum.y, bsum.z, bsum.w};
#pragma unroll
            for (int t = 0; t < (TA == 4 ? 4 : 1); ++t) {
                const int n = ncol + t;
                if (n < a.n_valid) {
                    float* d1 = a.dbias + (size_t)zb * a.z_bias_off + n;
                    *d1 = a.accumulate ? *d1 + bs[t] : bs[t];
                    if (a.dbias2) { float* d2 = a.dbias2 + n; *d2 = a.accumulate ? *d2 + bs[t] : bs[t]; }
                }
            }
        }
    }
    // ---- store: lane holds dW[n][k .. k+3] for n = n0 + {TA==4: 16*kk + 4*reg + ta | 4*kk + reg}, k = k0 + 4*ij ----
    if (!kvalid) return;
#pragma unroll
    for (int ta = 0; ta < TA; ++ta) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int n = n0 + (TA == 4 ? 16 * kk + 4 * reg + ta : 4 * kk + reg);
            if (n >= a.n_valid) continue;
            float4 v = make_float4(acc[ta][0][reg], acc[ta][1][reg], acc[ta][2][reg], acc[ta][3][reg]);
            float* op;
            if (a.partial) op = a.out + ((size_t)bz * a.n_valid + n) * a.K + kcol;
            else op = a.out + (size_t)zb * a.z_out_off + (size_t)n * a.ldw + a.k_off + kcol;
            if (!a.partial && a.accumulate) {
                const float4 o = *reinterpret_cast<const float4*>(op);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *reinterpret_cast<float4*>(op) = v;
        }
    }
}

template <int TA, bool RS>
__global__ void __launch_bounds__(256) wgrad_kernel(const gcpx_wgrad_args a) {
    wgrad_body<TA, RS>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z);
}

__host__ __device__ inline void wgrad_grid(const gcpx_wgrad_args& a, bool rs, int& gx, int& gy, int& gz) {
    const int kch = (a.K + 63) / 64;
    gx = rs ? kch : (kch + 3) / 4;
    gy = a.N > 16 ? (a.N + 63) / 64 : 1;
    gz = a.nbatch > 1 ? a.nbatch : a.nsplit;
}

// Grouped launch: the ~40 small weight-gradient GEMMs of one tree level (LSTM ih / hh, projections, embed slices, Predictor
// layers) are independent problems of 1 - 64 workgroups each; launched one by one they cost 15 - 35 us apiece on a side lane
// (launch + tail latency, not work).  One launch walks a device table of problem descriptors: block_start[p] is the first
// workgroup of problem p.
template <int TA, bool RS>
__global__ void __launch_bounds__(256) wgrad_group_kernel(const gcpx_wgrad_args* __restrict__ tab, const int* __restrict__ block_start,
                                                          const int nprob) {
    int p = 0;
    while (p + 1 < nprob && (int)blockIdx.x >= block_start[p + 1]) ++p;       // wave-uniform scan (nprob <= 64)
    const gcpx_wgrad_args a = tab[p];
    int gx, gy, gz;
    wgrad_grid(a, RS, gx, gy, gz);
    const int lb = blockIdx.x - block_start[p];
    wgrad_body<TA, RS>(a, lb % gx, (lb / gx) % gy, lb / (gx * gy), gz);
}

// partial [nsplit][N][K] -> dst in the canonical parameter layout.  A workgroup owns 16 consecutive outputs; 16 thread
// groups split the partial index and are combined through LDS in a fixed order.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, const int nsplit, const int N,
                                                           const int K, float* __restrict__ dst, const int map_mode,
                                                           const int Cin, const int ntap, const int Cout,
                                                           const int* __restrict__ n_map, const long long ldw,
                                                           const int k_off, const int accumulate) {
    __shared__ float red[16][17];
    const int oi = threadIdx.x & 15, zl = threadIdx.x >> 4;
    const long long total = (long long)N * K;
    const long long idx = (long long)blockIdx.x * 16 + oi;
    float s = 0.f;
    if (idx < total)
        for (int z = zl; z < nsplit; z += 16) s += partial[(size_t)z * total + idx];
    red[zl][oi] = s;
    __syncthreads();
    if (zl != 0 || idx >= total) return;
    s = 0.f;
#pragma unroll
    for (int z = 0; z < 16; ++z) s += red[z][oi];
    const int n = (int)(idx / K), k = (int)(idx % K);
    long long o;
    if (map_mode == GCPX_WMAP_LINEAR) {
        o = (long long)n * ldw + k_off + k;
    } else if (map_mode == GCPX_WMAP_CONV) {          // k = (tap, ci) -> w[n][ci][tap]
        const int nn = n_map ? n_map[n] : n;
        if (nn < 0) return;
        const int tap = k / Cin, ci = k % Cin;
        o = ((long long)nn * Cin + ci) * ntap + tap;
    } else {                                           // ConvTranspose 1x1 -> 4x4: n = (tap, co), k = ci -> w[ci][co][tap]
        const int tap = n / Cout, co = n % Cout;
        o = ((long long)k * Cout + co) * ntap + tap;
    }
    dst[o] = accumulate ? dst[o] + s : s;
}

// column sums: db[n] (+)= sum_r dY[r][n].  A workgroup owns a chunk of rows and up to 256 float4 column groups: thread
// (rr, c4) walks rows rr, rr + RPI, ... of the chunk with 4 independent float4 loads in flight, the RPI row classes are
// combined through LDS in a fixed order.  gridDim.y = row chunks: > 1 writes partial [chunks][N] (deterministic second pass),
// 1 writes dst directly.
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ dy, const long long ldy, const int R, const int N,
                                                     const int dy_rpb, const long long dy_sb, float* __restrict__ partial,
                                                     float* __restrict__ dst, float* __restrict__ dst2, const int accumulate,
                                                     const int tpr_log2) {
    __shared__ float4 red[256];
    const int tpr = 1 << tpr_log2, rpi = 256 >> tpr_log2;         // threads per row, rows per iteration
    const int c4 = blockIdx.x * tpr + (threadIdx.x & (tpr - 1)), rr = threadIdx.x >> tpr_log2;
    const int nchunk = gridDim.y;
    const int rows_per = (R + nchunk - 1) / nchunk;
    const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    const bool cvalid = 4 * c4 < N;
    auto row_ptr = [&](int r) -> const float* {
        if (dy_sb) { const int b = r / dy_rpb; return dy + (size_t)b * dy_sb + (size_t)(r - b * dy_rpb) * ldy + 4 * c4; }
        return dy + (size_t)r * ldy + 4 * c4;
    };
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cvalid) {
        int r = r0 + rr;
        for (; r + 3 * rpi < r1; r += 4 * rpi) {
            const float4 v0 = *reinterpret_cast<const float4*>(row_ptr(r));
            const float4 v1 = *reinterpret_cast<const float4*>(row_ptr(r + rpi));
            const float4 v2 = *reinterpret_cast<const float4*>(row_ptr(r + 2 * rpi));
            const float4 v3 = *reinterpret_cast<const float4*>(row_ptr(r + 3 * rpi));
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; r < r1; r += rpi) {
            const float4 v = *reinterpret_cast<const float4*>(row_ptr(r));
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rr != 0 || !cvalid) return;
    for (int k = 1; k < rpi; ++k) {
        const float4 v = red[(k << tpr_log2) + (threadIdx.x & (tpr - 1))];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float o[4] = {s.x, s.y, s.z, s.w};
    for (int t = 0; t < 4; ++t) {
        const int n = 4 * c4 + t;
        if (n >= N) break;
        if (nchunk > 1) partial[(size_t)blockIdx.y * N + n] = o[t];
        else {
            dst[n] = accumulate ? dst[n] + o[t] : o[t];
            if (dst2) dst2[n] = accumulate ? dst2[n] + o[t] : o[t];
        }
    }
}

}  // namespace

static int wgrad_check(const gcpx_wgrad_args* a) {
    GCPX_CHECK_ARG(a != nullptr, "null args");
    GCPX_CHECK_ARG(a->dy && a->x && a->out, "dy / x / out is NULL");
    GCPX_CHECK_ARG(a->R > 0 && a->N > 0 && a->K > 0 && a->K % 4 == 0, "bad R/N/K (K % 4)");
    GCPX_CHECK_ARG(a->n_valid > 0 && a->n_valid <= a->N, "n_valid out of range");
    GCPX_CHECK_ARG(a->nsplit >= 1 && (a->partial || a->nsplit == 1), "row splits need partial output");
    GCPX_CHECK_ARG(a->nbatch <= 1 || (!a->partial && a->nsplit == 1), "batched launches write directly");
    GCPX_CHECK_ARG(!a->dbias || !a->partial, "fused bias gradient needs direct output");
    GCPX_CHECK_ARG(a->mode >= GCPX_WG_ROWS && a->mode <= GCPX_WG_CONV4X4S2, "bad mode");
    GCPX_CHECK_ARG(a->dy_sb == 0 || a->dy_rpb > 0, "dy_rpb <= 0");
    if (a->mode == GCPX_WG_ROWS || a->mode == GCPX_WG_CONV1D) GCPX_CHECK_ARG(a->rpb > 0, "rpb <= 0");
    if (a->mode != GCPX_WG_ROWS) GCPX_CHECK_ARG(a->Cin > 0 && a->Cin % 4 == 0 && a->K % a->Cin == 0, "conv modes: K = ntap * Cin");
    if (a->mode == GCPX_WG_CONV3X3 || a->mode == GCPX_WG_CONV4X4S2)
        GCPX_CHECK_ARG(a->H > 1 && a->W > 1 && (a->H & (a->H - 1)) == 0 && (a->W & (a->W - 1)) == 0, "conv modes: H, W powers of two");
    GCPX_CHECK_ARG(!a->scale || a->mode != GCPX_WG_ROWS || (a->cmod > 0 && (a->cmod & (a->cmod - 1)) == 0), "cmod must be a power of two");
    const bool wide = a->N > 16;
    GCPX_CHECK_ARG(!wide || (a->N % 4 == 0 && a->ldy % 4 == 0), "N > 16 needs N % 4 == 0 and ldy % 4 == 0");
    GCPX_CHECK_ARG(a->partial || (a->ldw % 4 == 0 && a->k_off % 4 == 0), "direct output needs ldw, k_off % 4 == 0");
    return GCPX_OK;
}

// kernel variant of a problem: bit 0 = wide (64-column A tiles), bit 1 = row split inside the workgroup
static int wgrad_variant(const gcpx_wgrad_args* a) {
    const int kch = (a->K + 63) / 64;
    const bool wide = a->N > 16;
    const int nch = wide ? (a->N + 63) / 64 : 1;
    const int nz = a->nbatch > 1 ? a->nbatch : a->nsplit;
    // few tiles x many rows: share each tile between the 4 wavefronts of a workgroup (row split inside the workgroup)
    const long long rows_per = (a->R + a->nsplit - 1) / a->nsplit;
    const bool rs = (long long)((kch + 3) / 4) * nch * nz < 512 && rows_per >= 64;
    return (wide ? 1 : 0) | (rs ? 2 : 0);
}

extern "C" int gcpx_wgrad(const gcpx_wgrad_args* a, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int st = wgrad_check(a);
    if (st != GCPX_OK) return st;
    if (a->split_f16 && gcpx_wgrad_rows_split_applies(a)) {
        const int s2 = gcpx_launch_wgrad_rows_split(a, stream);
        if (s2 != GCPX_OK) return s2;
        GCPX_CHECK_LAUNCH();
        return GCPX_OK;
    }
    const int v = wgrad_variant(a);
    int gx, gy, gz;
    wgrad_grid(*a, (v & 2) != 0, gx, gy, gz);
    const dim3 grid(gx, gy, gz);
    if (v == 3) hipLaunchKernelGGL((wgrad_kernel<4, true>), grid, dim3(256), 0, stream, *a);
    else if (v == 2) hipLaunchKernelGGL((wgrad_kernel<1, true>), grid, dim3(256), 0, stream, *a);
    else if (v == 1) hipLaunchKernelGGL((wgrad_kernel<4, false>), grid, dim3(256), 0, stream, *a);
    else hipLaunchKernelGGL((wgrad_kernel<1, false>), grid, dim3(256), 0, stream, *a);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_wgrad_classify(const gcpx_wgrad_args* a, int32_t row_split, int32_t* variant, int32_t* nblocks) {
    const int st = wgrad_check(a);
    if (st != GCPX_OK) return st;
    GCPX_CHECK_ARG(variant && nblocks, "null output");
    if (a->split_f16 && gcpx_wgrad_rows_split_applies(a)) {          // variant 4: the split-f16 kernel, one workgroup per 128 x 128 block
        *variant = 4;
        *nblocks = gcpx_wgrad_rows_split_blocks(a);
        return GCPX_OK;
    }
    int v = wgrad_variant(a);
    if (row_split == 0) v &= 1;          // the caller fills the chip with the group: one wavefront per 64 x 64 tile
    else if (row_split == 1) v |= 2;
    int gx, gy, gz;
    wgrad_grid(*a, (v & 2) != 0, gx, gy, gz);
    *variant = v;
    *nblocks = gx * gy * gz;
    return GCPX_OK;
}

extern "C" int gcpx_wgrad_group(const gcpx_wgrad_args* tab, const int32_t* block_start, int32_t nprob, int32_t total_blocks,
                                int32_t variant, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(tab && block_start && nprob > 0 && nprob <= 64 && total_blocks > 0 && variant >= 0 && variant <= 4, "bad arguments");
    if (variant == 4) {
        const int s2 = gcpx_launch_wgrad_rows_split_group(tab, block_start, nprob, total_blocks, stream);
        if (s2 != GCPX_OK) return s2;
        GCPX_CHECK_LAUNCH();
        return GCPX_OK;
    }
    const dim3 grid(total_blocks);
    if (variant == 3) hipLaunchKernelGGL((wgrad_group_kernel<4, true>), grid, dim3(256), 0, stream, tab, block_start, nprob);
    else if (variant == 2) hipLaunchKernelGGL((wgrad_group_kernel<1, true>), grid, dim3(256), 0, stream, tab, block_start, nprob);
    else if (variant == 1) hipLaunchKernelGGL((wgrad_group_kernel<4, false>), grid, dim3(256), 0, stream, tab, block_start, nprob);
    else hipLaunchKernelGGL((wgrad_group_kernel<1, false>), grid, dim3(256), 0, stream, tab, block_start, nprob);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_wgrad_reduce(const float* partial, int32_t nsplit, int32_t N, int32_t K, float* dst, int32_t map_mode,
                                 int32_t Cin, int32_t ntap, int32_t Cout, const int32_t* n_map, int64_t ldw, int32_t k_off,
                                 int32_t accumulate, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(partial && dst && nsplit >= 1 && N > 0 && K > 0, "bad arguments");
    GCPX_CHECK_ARG(map_mode >= GCPX_WMAP_LINEAR && map_mode <= GCPX_WMAP_CONVT, "bad map_mode");
    const long long total = (long long)N * K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, partial, nsplit, N, K,
                       dst, map_mode, Cin, ntap, Cout, n_map, (long long)ldw, k_off, accumulate);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_colsum(const float* dy, int64_t ldy, int32_t R, int32_t N, int32_t dy_rpb, int64_t dy_sb, int32_t nsplit,
                           float* partial, float* dst, float* dst2, int32_t accumulate, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(dy && R > 0 && N > 0 && nsplit >= 1, "bad arguments");
    GCPX_CHECK_ARG(nsplit > 1 ? partial != nullptr : dst != nullptr, "missing output");
    GCPX_CHECK_ARG(dy_sb == 0 || dy_rpb > 0, "dy_rpb <= 0");
    GCPX_CHECK_ARG(ldy % 4 == 0 && dy_sb % 4 == 0 && ldy >= ((N + 3) & ~3), "column sums read float4: ldy, dy_sb % 4 == 0, ldy >= N rounded up");
    const int c4 = (N + 3) / 4;
    int tl = 0;
    while ((1 << tl) < c4 && tl < 8) ++tl;                 // threads per row = next power of two >= N / 4, at most 256
    hipLaunchKernelGGL(colsum_kernel, dim3((c4 + (1 << tl) - 1) >> tl, nsplit), dim3(256), 0, stream, dy, (long long)ldy, R, N,
                       dy_rpb, (long long)dy_sb, partial, dst, dst2, accumulate, tl);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
