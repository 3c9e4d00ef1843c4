// Weight gradients of the tree's Linear / LSTMCell layers on the f16 matrix pipes of gfx950 with f32-equivalent arithmetic ("split-f16"):
//     dW[n][k] (+)= sum over rows r of dY[r][n] * X[r][k]                                   ("TN" GEMM, reduction over rows)
// (backward of HiddenStatePredictorModel / the split_linear merge / the Predictors called through
//  /root/reference/gcp/prediction/models/tree/tree_module.py:67-114 and tree_lstm.py:43-49; same descriptor, outputs and fused bias
//  gradient as the direct mode of gcpx_wgrad in wgrad.hip, whose wavefronts each stream their own 64 + 64 columns of both operands
//  through registers: bound by the bytes a CU pulls — the 40 weight gradients of tree level 6 took 0.6 - 1.3 ms at c2.)
//
// One 256-thread workgroup owns a 128 x 128 block of dW and walks ALL rows in passes of 64: both operand tiles ([64 rows][128 columns]
// of dY and of X) are staged row-major into f16 planes in LDS — each value as two pieces under a running power-of-two scale per
// workgroup and operand, exactly as in wgrad_conv_split.hip — and come back through the transposing LDS read with the MFMA k index on
// the rows.  Wavefront w accumulates k-column tiles {w, w + 4} x all eight n-column tiles: 16 accumulator tiles, 96 MFMAs per pass.
// Three MFMAs per product (x2 y1 + x1 y2 + x1 y1, small terms first), f32 accumulate: the error of a sum is a few f32 roundings of its
// largest terms (tests/test_gpu_kernels.py measures it against float64 next to the exact kernel).
#include "common.h"
#include "split_tr.h"

#include <cstdlib>

namespace {

template <int NTA, int NTB>
struct RowsSplitCfg {
    static constexpr int PL = 64 * 16 + 16;                     // halfs per (piece, 16-column tile) plane: [64 rows][16 columns] + 32 B
    static constexpr int A_HALFS = 2 * NTA * PL, B_HALFS = 2 * NTB * PL;
    static constexpr int LDS_BYTES = (A_HALFS + B_HALFS) * 2 + 64;
    static constexpr int GPW = NTB / 4;                         // k-column tiles per wavefront
    static constexpr int NACC = GPW * NTA;
};

// (bx, by, bz) = k-column block, n-column block, batch index of the problem
template <int NTA, int NTB>
__device__ __forceinline__ void wgrad_rows_split_body(const gcpx_wgrad_args& a, const int bx, const int by, const int bz, float4* smem4) {
    using Cfg = RowsSplitCfg<NTA, NTB>;
    constexpr int PL = Cfg::PL, GPW = Cfg::GPW, NACC = Cfg::NACC, NS = NTA + NTB;
    _Float16* sA = reinterpret_cast<_Float16*>(smem4);                    // [2][NTA][PL]
    _Float16* sB = sA + Cfg::A_HALFS;                                     // [2][NTB][PL]
    float* red = reinterpret_cast<float*>(sB + Cfg::B_HALFS);             // [4][2] tile maxima

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ij = lane & 15, kq = lane >> 4;
    const int n0 = by * NTA * 16, k0 = bx * NTB * 16;
    const int R = a.R;
    const int npass = (R + 63) / 64;
    const float* __restrict__ dyp = a.dy + (size_t)bz * a.z_dy_off + n0;
    const float* __restrict__ xpb = a.x + (size_t)bz * a.z_x_off + k0;
    const bool do_bias = a.dbias != nullptr && bx == 0;

    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float4 bsum[NTA];
#pragma unroll
    for (int s = 0; s < NTA; ++s) bsum[s] = make_float4(0.f, 0.f, 0.f, 0.f);

    // staging: thread (sp, sc4) moves float4 sc4 of row sp of every 16-column tile (slot s = tile s): 64 contiguous bytes per row and tile
    // in global memory, 8-byte stores of 4 columns into the row-major planes
    const int sp = tid >> 2, sc4 = tid & 3;
    const int dloff0 = sp * 16 + 4 * sc4;
    float4 pre[NS];
    const float *dyrow = nullptr, *xrow = nullptr;
    bool rok = false, xok = false;                                         // validity of the rows the registers hold
    bool nrok = false, nxok = false;
    auto row_base = [&](const int pass) __attribute__((always_inline)) {
        const int r = pass * 64 + sp;
        nrok = r < R;
        const int rc = nrok ? r : R - 1;                                   // (loads stay unconditional; rows past the end are zeroed before use)
        if (a.dy_sb) {
            const int b = rc / a.dy_rpb;
            dyrow = dyp + (size_t)b * a.dy_sb + (size_t)(rc - b * a.dy_rpb) * a.ldy;
        } else {
            dyrow = dyp + (size_t)rc * a.ldy;
        }
        const int b = rc / a.rpb, j = rc - b * a.rpb;
        nxok = nrok;
        xrow = xpb + (size_t)b * a.sb + (size_t)j * a.sr;
    };
    auto load_a = [&](const int s) __attribute__((always_inline)) { pre[s] = *reinterpret_cast<const float4*>(dyrow + 16 * s + 4 * sc4); };
    auto load_b = [&](const int s) __attribute__((always_inline)) { pre[NTA + s] = *reinterpret_cast<const float4*>(xrow + 16 * s + 4 * sc4); };

    int ea = 0, eb = 0;                 // running scales of dY / X: staged values are multiplied by 2^ea / 2^eb
    bool have_a = false, have_b = false;
    // operand addresses of the transposing reads (see wgrad_conv_split.hip): lane (s = lane & 15, group kq) points at row 4 kq + (s >> 2)
    // (first read; + 16: second) of a 32-row k-step, columns 4 (s & 3) .. + 3
    const int rpix = 4 * kq + (ij >> 2), rch = 4 * (ij & 3);

    row_base(0);
#pragma unroll
    for (int s = 0; s < NTA; ++s) load_a(s);
#pragma unroll
    for (int s = 0; s < NTB; ++s) load_b(s);
    rok = nrok; xok = nxok;

    for (int pass = 0; pass < npass; ++pass) {
        // ---- largest |value| of the tile, per operand ----
        float ma = 0.f, mb = 0.f;
#pragma unroll
        for (int s = 0; s < NTA; ++s) {
            if (!rok) pre[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            ma = fmaxf(ma, fmaxf(fmaxf(fabsf(pre[s].x), fabsf(pre[s].y)), fmaxf(fabsf(pre[s].z), fabsf(pre[s].w))));
        }
#pragma unroll
        for (int s = 0; s < NTB; ++s) {
            if (!xok) pre[NTA + s] = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 v = pre[NTA + s];
            mb = fmaxf(mb, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        ma = wave_max_nonneg(ma);
        mb = wave_max_nonneg(mb);
        if (lane == 0) { red[2 * wave] = ma; red[2 * wave + 1] = mb; }
        __syncthreads();                                      // maxima visible; the previous pass's operand reads are done
        ma = fmaxf(fmaxf(red[0], red[2]), fmaxf(red[4], red[6]));
        mb = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
        ma = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ma)));
        mb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, mb)));
        float resc = 1.f;
        if (ma > 0.f) {
            const int e = max(-100, min(100, 14 + 127 - (int)((__float_as_uint(ma) >> 23) & 0xff)));      // ma 2^e in [2^14, 2^15)
            if (!have_a) { ea = e; have_a = true; }
            else if (e < ea) { resc *= __uint_as_float((unsigned)(127 + max(e - ea, -126)) << 23); ea = e; }
        }
        if (mb > 0.f) {
            const int e = max(-100, min(100, 14 + 127 - (int)((__float_as_uint(mb) >> 23) & 0xff)));
            if (!have_b) { eb = e; have_b = true; }
            else if (e < eb) { resc *= __uint_as_float((unsigned)(127 + max(e - eb, -126)) << 23); eb = e; }
        }
        if (resc != 1.f) {                                    // larger values than before: lower the scale, rescale the sums (exact)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] *= resc;
        }
        const float sa = __uint_as_float((unsigned)(127 + ea) << 23), sb = __uint_as_float((unsigned)(127 + eb) << 23);

        // ---- registers -> LDS: split, 8-byte stores; each slot's register takes the next pass's load at once (the last pass loads
        //      itself again: the loads stay unconditional) ----
        row_base(min(pass + 1, npass - 1));
#pragma unroll
        for (int s = 0; s < NTA; ++s) {
            h4 p1, p2;
            split4(pre[s], sa, p1, p2);
            if (do_bias) { bsum[s].x += pre[s].x; bsum[s].y += pre[s].y; bsum[s].z += pre[s].z; bsum[s].w += pre[s].w; }
            load_a(s);
            *reinterpret_cast<h4*>(sA + s * PL + dloff0) = p1;
            *reinterpret_cast<h4*>(sA + (NTA + s) * PL + dloff0) = p2;
        }
#pragma unroll
        for (int s = 0; s < NTB; ++s) {
            h4 p1, p2;
            split4(pre[NTA + s], sb, p1, p2);
            load_b(s);
            *reinterpret_cast<h4*>(sB + s * PL + dloff0) = p1;
            *reinterpret_cast<h4*>(sB + (NTB + s) * PL + dloff0) = p2;
        }
        rok = nrok; xok = nxok;
        __syncthreads();

        // ---- MFMA phase: two k-steps of 32 rows ----
        __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
        for (int step = 0; step < 2; ++step) {
            const _Float16* ap = sA + (32 * step + rpix) * 16 + rch;
            const _Float16* bp = sB + (32 * step + rpix) * 16 + rch;
            h8 b1[GPW], b2[GPW];
#pragma unroll
            for (int g = 0; g < GPW; ++g) {
                const int kt = wave + 4 * g;
                b1[g] = ds_tr8(bp + kt * PL, 16 * 16);
                b2[g] = ds_tr8(bp + (NTB + kt) * PL, 16 * 16);
            }
#pragma unroll
            for (int nt = 0; nt < NTA; ++nt) {
                const h8 a1 = ds_tr8(ap + nt * PL, 16 * 16), a2 = ds_tr8(ap + (NTA + nt) * PL, 16 * 16);
#pragma unroll
                for (int g = 0; g < GPW; ++g) {
                    f32x4& c = acc[g * NTA + nt];
                    c = mfma32h(a2, b1[g], c);        // small terms first
                    c = mfma32h(a1, b2[g], c);
                    c = mfma32h(a1, b1[g], c);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }

    // ---- dW: lane holds n = n0 + nt*16 + 4*kq + reg, k = k0 + kt*16 + ij ----
    const float ia = __uint_as_float((unsigned)(127 - ea) << 23), ib = __uint_as_float((unsigned)(127 - eb) << 23);
    float* outp = a.out + (size_t)bz * a.z_out_off + a.k_off + k0;
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const int k = (wave + 4 * g) * 16 + ij;
#pragma unroll
        for (int nt = 0; nt < NTA; ++nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + nt * 16 + 4 * kq + r;
                if (n < a.n_valid) {
                    float* op = outp + (size_t)n * a.ldw + k;
                    const float v = acc[g * NTA + nt][r] * ia * ib;
                    *op = a.accumulate ? *op + v : v;
                }
            }
        }
    }

    // ---- bias gradient: column sums of dY (workgroups of the first k-column block): rows over the 16 row lanes of a wavefront, then
    //      over the four wavefronts through LDS, in a fixed order ----
    if (do_bias) {
        __syncthreads();                                      // every wavefront is done with the operand planes
        float4* bred = reinterpret_cast<float4*>(smem4);      // [4 wavefronts][NTA][4 column groups]
#pragma unroll
        for (int s = 0; s < NTA; ++s) {
            float4 v = bsum[s];
#pragma unroll
            for (int m = 4; m < 64; m <<= 1) {
                v.x += __shfl_xor(v.x, m); v.y += __shfl_xor(v.y, m); v.z += __shfl_xor(v.z, m); v.w += __shfl_xor(v.w, m);
            }
            if (lane < 4) bred[(wave * NTA + s) * 4 + lane] = v;
        }
        __syncthreads();
        if (tid < NTA * 4) {
            const float4 v0 = bred[tid], v1 = bred[NTA * 4 + tid], v2 = bred[2 * NTA * 4 + tid], v3 = bred[3 * NTA * 4 + tid];
            const float o[4] = {(v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y), (v0.z + v1.z) + (v2.z + v3.z),
                                (v0.w + v1.w) + (v2.w + v3.w)};
            const int nb = n0 + (tid >> 2) * 16 + (tid & 3) * 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int n = nb + t;
                if (n < a.n_valid) {
                    float* d1 = a.dbias + (size_t)bz * a.z_bias_off + n;
                    *d1 = a.accumulate ? *d1 + o[t] : o[t];
                    if (a.dbias2) { float* d2 = a.dbias2 + n; *d2 = a.accumulate ? *d2 + o[t] : o[t]; }
                }
            }
        }
    }
}

template <int NTA, int NTB>
__global__ void __launch_bounds__(256, 2) wgrad_rows_split_kernel(const gcpx_wgrad_args a) {
    extern __shared__ float4 smem4[];
    wgrad_rows_split_body<NTA, NTB>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem4);
}

// grouped launch (see wgrad_group_kernel in wgrad.hip): block_start[p] = first workgroup of problem p
template <int NTA, int NTB>
__global__ void __launch_bounds__(256, 2) wgrad_rows_split_group_kernel(const gcpx_wgrad_args* __restrict__ tab, const int* __restrict__ block_start,
                                                                        const int nprob) {
    extern __shared__ float4 smem4[];
    int p = 0;
    while (p + 1 < nprob && (int)blockIdx.x >= block_start[p + 1]) ++p;       // wave-uniform scan (nprob <= 64)
    const gcpx_wgrad_args a = tab[p];
    const int gx = a.K / (NTB * 16), gy = a.N / (NTA * 16);
    const int lb = blockIdx.x - block_start[p];
    wgrad_rows_split_body<NTA, NTB>(a, lb % gx, (lb / gx) % gy, lb / (gx * gy), smem4);
}

template <class K>
int set_lds(K kern, int bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        gcpx_set_error("wgrad rows split: hipFuncSetAttribute(%d B LDS): %s", bytes, hipGetErrorString(e));
        return GCPX_ERR_HIP;
    }
    return GCPX_OK;
}

}  // namespace

// Does this direct-mode problem have a split-f16 form, and is it worth it?  Plain rows (no gather, shift, affine or activation on load),
// whole 128 x 128 blocks of dW, direct output, and enough rows for the staged passes to pay (below, the in-workgroup row split of
// wgrad.hip is latency-bound anyway).
bool gcpx_wgrad_rows_split_applies(const gcpx_wgrad_args* a) {
    static const int min_rows = [] { const char* e = getenv("GCPX_WGRAD_SPLIT_MIN_ROWS"); return e ? atoi(e) : 256; }();
    return a->mode == GCPX_WG_ROWS && !a->rowidx && !a->scale && !a->act && a->shift == 0 && !a->partial && a->nsplit == 1 &&
           a->R >= min_rows && a->N % 128 == 0 && a->K % 128 == 0 && a->n_valid == a->N && a->ldy % 4 == 0 && a->sr % 4 == 0 &&
           a->sb % 4 == 0 && a->dy_sb % 4 == 0 && a->rpb > 0 && (a->dy_sb == 0 || a->dy_rpb > 0);
}

int gcpx_wgrad_rows_split_blocks(const gcpx_wgrad_args* a) {
    return (a->K / 128) * (a->N / 128) * (a->nbatch > 1 ? a->nbatch : 1);
}

int gcpx_launch_wgrad_rows_split(const gcpx_wgrad_args* a, hipStream_t stream) {
    using Cfg = RowsSplitCfg<8, 8>;
    static int lds_set = -1;
    if (lds_set < 0) lds_set = set_lds(wgrad_rows_split_kernel<8, 8>, Cfg::LDS_BYTES);
    if (lds_set != GCPX_OK) return lds_set;
    const dim3 grid(a->K / 128, a->N / 128, a->nbatch > 1 ? a->nbatch : 1);
    hipLaunchKernelGGL((wgrad_rows_split_kernel<8, 8>), grid, dim3(256), Cfg::LDS_BYTES, stream, *a);
    return GCPX_OK;
}

int gcpx_launch_wgrad_rows_split_group(const gcpx_wgrad_args* tab, const int32_t* block_start, int nprob, int total_blocks, hipStream_t stream) {
    using Cfg = RowsSplitCfg<8, 8>;
    static int lds_set = -1;
    if (lds_set < 0) lds_set = set_lds(wgrad_rows_split_group_kernel<8, 8>, Cfg::LDS_BYTES);
    if (lds_set != GCPX_OK) return lds_set;
    hipLaunchKernelGGL((wgrad_rows_split_group_kernel<8, 8>), dim3(total_blocks), dim3(256), Cfg::LDS_BYTES, stream, tab, block_start, nprob);
    return GCPX_OK;
}
