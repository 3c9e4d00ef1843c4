// Weight gradient of the decoder's 3x3 convs (pad 1) on gfx950 f32 MFMA, LDS-tiled:
//     dW[n][tap][ci] = sum over pixels p of dY[p][n] * U[p + tap][ci]
// (backward of blox ConvDecoder blocks / gen_head as called through /root/reference/gcp/prediction/models/tree/
//  tree_dense_rec.py:42; the reference gets this from cuDNN via torch autograd).
//
// The generic gcpx_wgrad kernel reads both operands from global memory for every 64x64 block of dW (16 FLOP/B: L2
// bound).  Here a persistent workgroup stages a TH x TW pixel tile of dY and the haloed (TH+2) x (TW+2) region of U in
// LDS once and computes ALL taps / channels of dW for it:
//   * MFMA i side = output channel n (A operand: lane (i, kk) reads dY[pixel p0+kk][n]), j side = input channel ci
//     (B operand: lane (kk, j) reads U[pixel p0+kk shifted by the tap][ci]); the MFMA k index walks 4 pixels of a row;
//   * wavefront w owns the (tap, ci-tile) groups g = w, w + NW, ... for all NT n-tiles; accumulators stay in registers
//     across the whole launch and are written once as a partial [workgroup][n][tap*Cin + ci]; gcpx_wgrad_reduce sums the
//     partials in a fixed order (deterministic) and maps them to the torch weight layout;
//   * the next tile's global loads are issued before the MFMA phase of the current one.
#include "common.h"

namespace {

template <int NT, int CIT, int NW>
struct WCfg {
    static constexpr int G = 9 * CIT;                       // (tap, ci-tile) groups
    static constexpr int GPW = (G + NW - 1) / NW;            // groups per wavefront
    static constexpr int N = NT * 16, CC = CIT * 16;
    static constexpr int NP = (N % 32 == 16) ? N : N + 16;   // LDS pitches = 16 mod 32: conflict-free operand reads
    static constexpr int CP = (CC % 32 == 16) ? CC : CC + 16;
};

// BAL (output head: NT = 7 n-tiles, 9 taps, NW = 4): the 63 (tap, n-tile) accumulator tiles are dealt 16 / 16 / 16 / 15 over the four
// wavefronts instead of 3 taps x 7 tiles over three.  The three-wave version needed 240 registers (two wavefronts per SIMD), so a CU held
// two workgroups = 6 wavefronts = 2 / 2 / 1 / 1 per SIMD and the 768-workgroup grid ran as 1.5 rounds; 2 x 4 balanced wavefronts per CU fill
// every SIMD's MFMA pipe in one round.
//   wavefront w < 3: taps 3w .. 3w+2 x n-tiles 0 .. 4, plus (tap 3w, n-tile 5);   wavefront 3: n-tile 6 x 9 taps + n-tile 5 x taps {1,2,4,5,7,8}
template <int NT, int CIT, int NW, int TW, bool BAL = false>
__global__ void __launch_bounds__(NW * 64) wgrad_conv3x3_kernel(const float* __restrict__ dy, const float* __restrict__ u,
                                                                float* __restrict__ partial, const int F, const int H,
                                                                const int W, const int Cin, const int ldy, const int TH) {
    using Cfg = WCfg<NT, CIT, NW>;
    constexpr int GPW = Cfg::GPW, N = Cfg::N, CC = Cfg::CC, NP = Cfg::NP, CP = Cfg::CP, G = Cfg::G;
    constexpr int NTH = NW * 64;
    extern __shared__ float4 smem4[];
    float* sdy = reinterpret_cast<float*>(smem4);            // [PX][NP]
    constexpr int RW = TW + 2;
    const int PX = TH * TW, RH = TH + 2;
    float* su = sdy + PX * NP;                               // [RH*RW][CP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ij = lane & 15, kk = lane >> 4;
    const int ci0 = blockIdx.y * CC;                         // this workgroup's input-channel chunk
    const int ntx = W / TW, nty = H / TH;
    const int ntiles = F * nty * ntx;

    static_assert(!BAL || (NT == 7 && CIT == 1 && NW == 4), "balanced unit split is written for the 112-column output head");
    constexpr int NACC_G = BAL ? 1 : GPW, NACC_N = BAL ? 16 : NT;
    f32x4 acc[NACC_G][NACC_N];
#pragma unroll
    for (int g = 0; g < NACC_G; ++g)
#pragma unroll
        for (int nt = 0; nt < NACC_N; ++nt) acc[g][nt] = f32x4{0, 0, 0, 0};

    // per-group LDS offsets of the B operand (tap shift + channel tile), in floats
    int boff[GPW];
    bool gok[GPW];
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const int gi = wave + g * NW;
        gok[g] = gi < G;
        const int tap = gok[g] ? gi / CIT : 0, cit = gok[g] ? gi % CIT : 0;
        boff[g] = ((tap / 3) * RW + (tap % 3)) * CP + cit * 16 + ij;
    }

    // staging slots: dY tile = PX * N/4 float4, U region = RH*RW * CC/4 float4
    constexpr int N4 = N / 4, C4 = CC / 4;
    constexpr int MAXS = (NT == 7) ? (BAL ? 10 : 13) : 8;     // float4 slots per thread kept in flight
    const int ndy = PX * N4, nu = RH * RW * C4;
    const int nslot = ndy + nu;
    float4 pre[MAXS];

    // staging slots are the same for every tile: global offsets relative to the tile origin, LDS offsets and (for the haloed U
    // region) the position inside the region are computed once, not per tile (the divisions by N / 4 = 28 etc. dominated the VALU
    // count of the kernel: 2.3 - 13 VALU instructions per MFMA in the PMC counters)
    int goff[MAXS], loff[MAXS];
    short sry[MAXS], srx[MAXS];                              // < 0: dY slot; else region row / column of a U slot
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        const int idx = tid + s * NTH;
        goff[s] = 0; loff[s] = -1; sry[s] = -1; srx[s] = -1;
        if (idx < ndy) {
            const int p = idx / N4, c4 = idx % N4;
            goff[s] = ((p / TW) * W + (p % TW)) * ldy + c4 * 4;
            loff[s] = p * NP + c4 * 4;
        } else if (idx < nslot) {
            const int k = idx - ndy;
            const int r = k / C4, c4 = k % C4;
            sry[s] = (short)(r / RW); srx[s] = (short)(r % RW);
            goff[s] = ((r / RW - 1) * W + (r % RW - 1)) * Cin + c4 * 4;
            loff[s] = PX * NP + r * CP + c4 * 4;
        }
    }
    auto issue = [&](int tile) {
        const int tx = tile % ntx;
        const int t2 = tile / ntx;
        const int f = t2 / nty, y0 = (t2 % nty) * TH, x0 = tx * TW;
        const float* dyb = dy + (((size_t)f * H + y0) * W + x0) * ldy;
        const float* ub = u + (((size_t)f * H + y0) * W + x0) * Cin + ci0;
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            pre[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (loff[s] < 0) continue;
            if (sry[s] < 0) {
                pre[s] = *reinterpret_cast<const float4*>(dyb + goff[s]);
            } else {
                const int y = y0 - 1 + sry[s], x = x0 - 1 + srx[s];
                if (y >= 0 && y < H && x >= 0 && x < W) pre[s] = *reinterpret_cast<const float4*>(ub + goff[s]);
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (loff[s] >= 0) *reinterpret_cast<float4*>(sdy + loff[s]) = pre[s];
    };

    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                      // previous tile's operand reads are done
        commit();
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (BAL) {
            auto tapoff = [&](int tap) { return ((tap / 3) * RW + (tap % 3)) * CP + ij; };
            auto opnd = [&](const int p0, const float*& ap, const float*& bp) {
                const int p = p0 + kk;
                ap = sdy + p * NP + ij;
                bp = su + ((p / TW) * RW + (p % TW)) * CP;
            };
            if (wave < 3) {
                const int bo0 = tapoff(3 * wave), bo1 = tapoff(3 * wave + 1), bo2 = tapoff(3 * wave + 2);
                auto rd = [&](const int p0, float (&a)[6], float (&b)[3]) {
                    const float *ap, *bp;
                    opnd(p0, ap, bp);
#pragma unroll
                    for (int nt = 0; nt < 6; ++nt) a[nt] = ap[nt * 16];
                    b[0] = bp[bo0]; b[1] = bp[bo1]; b[2] = bp[bo2];
                };
                auto mm = [&](const float (&a)[6], const float (&b)[3]) {
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int nt = 0; nt < 5; ++nt) acc[0][t * 5 + nt] = mfma16(a[nt], b[t], acc[0][t * 5 + nt]);
                    acc[0][15] = mfma16(a[5], b[0], acc[0][15]);
                };
                float aA[6], bA[3], aB[6], bB[3];
                rd(0, aA, bA);
                for (int p0 = 0; p0 < PX; p0 += 8) {
                    rd(p0 + 4, aB, bB);
                    mm(aA, bA);
                    if (p0 + 8 < PX) rd(p0 + 8, aA, bA);
                    mm(aB, bB);
                }
            } else {
                auto rd = [&](const int p0, float (&a)[2], float (&b)[9]) {
                    const float *ap, *bp;
                    opnd(p0, ap, bp);
                    a[0] = ap[5 * 16]; a[1] = ap[6 * 16];
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) b[tap] = bp[((tap / 3) * RW + (tap % 3)) * CP + ij];
                };
                auto mm = [&](const float (&a)[2], const float (&b)[9]) {
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        acc[0][tap] = mfma16(a[1], b[tap], acc[0][tap]);
                        if (tap % 3 != 0) {
                            const int k5 = 9 + (tap / 3) * 2 + (tap % 3) - 1;
                            acc[0][k5] = mfma16(a[0], b[tap], acc[0][k5]);
                        }
                    }
                };
                float aA[2], bA[9], aB[2], bB[9];
                rd(0, aA, bA);
                for (int p0 = 0; p0 < PX; p0 += 8) {
                    rd(p0 + 4, aB, bB);
                    mm(aA, bA);
                    if (p0 + 8 < PX) rd(p0 + 8, aA, bA);
                    mm(aB, bB);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            continue;
        }
        // operands of the next 4-pixel step are read while the current step's MFMAs run (two register sets, no copies)
        auto rd = [&](const int p0, float (&a)[NT], float (&b)[GPW]) {
            const int p = p0 + kk;
            const int py = p / TW, px = p % TW;
            const float* ap = sdy + p * NP + ij;
            const float* bp = su + (py * RW + px) * CP;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a[nt] = ap[nt * 16];
#pragma unroll
            for (int g = 0; g < GPW; ++g) b[g] = bp[boff[g]];
        };
        auto mm = [&](const float (&a)[NT], const float (&b)[GPW]) {
#pragma unroll
            for (int g = 0; g < GPW; ++g) {
                if (gok[g]) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[g][nt] = mfma16(a[nt], b[g], acc[g][nt]);
                }
            }
        };
        float aA[NT], bA[GPW], aB[NT], bB[GPW];
        rd(0, aA, bA);
        for (int p0 = 0; p0 < PX; p0 += 8) {               // PX % 8 == 0 (TH * TW = 64 or H * W >= 64)
            rd(p0 + 4, aB, bB);
            mm(aA, bA);
            if (p0 + 8 < PX) rd(p0 + 8, aA, bA);
            mm(aB, bB);
        }
        __builtin_amdgcn_s_setprio(0);
    }

    // partial [blockIdx.x][n][K = 9*Cin]: lane holds n = nt*16 + 4*kk + reg, k = tap*Cin + ci0 + cit*16 + ij
    const int K = 9 * Cin;
    float* out = partial + (size_t)blockIdx.x * N * K;
    if constexpr (BAL) {
        auto store = [&](const f32x4& v, int nt, int tap) {
            const int k = tap * Cin + ci0 + ij;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(nt * 16 + 4 * kk + r) * K + k] = v[r];
        };
        if (wave < 3) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) store(acc[0][t * 5 + nt], nt, 3 * wave + t);
            store(acc[0][15], 5, 3 * wave);
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                store(acc[0][tap], 6, tap);
                if (tap % 3 != 0) store(acc[0][9 + (tap / 3) * 2 + (tap % 3) - 1], 5, tap);
            }
        }
        return;
    }
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const int gi = wave + g * NW;
        if (gi >= G) continue;
        const int tap = gi / CIT, cit = gi % CIT;
        const int k = tap * Cin + ci0 + cit * 16 + ij;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(nt * 16 + 4 * kk + r) * K + k] = acc[g][nt][r];
    }
}

template <int NT, int CIT, int NW, int TW, bool BAL = false>
int launch_wc2(const float* dy, const float* u, float* partial, int F, int H, int W, int Cin, int ldy, int grid, hipStream_t stream) {
    using Cfg = WCfg<NT, CIT, NW>;
    int TH = 64 / TW;
    if (TH > H) TH = H;
    const int PX = TH * TW;
    const int nslot = PX * (Cfg::N / 4) + (TH + 2) * (TW + 2) * (Cfg::CC / 4);
    constexpr int MAXS = (NT == 7) ? (BAL ? 10 : 13) : 8;
    if (nslot > MAXS * NW * 64 || PX % 8 || W % TW || H % TH) {
        gcpx_set_error("gcpx_wgrad_conv3x3: unsupported tile (H=%d W=%d N=%d)", H, W, Cfg::N);
        return GCPX_ERR_UNSUPPORTED;
    }
    const size_t lds = ((size_t)PX * Cfg::NP + (size_t)(TH + 2) * (TW + 2) * Cfg::CP) * 4;
    auto kern = wgrad_conv3x3_kernel<NT, CIT, NW, TW, BAL>;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) {
            gcpx_set_error("%s: hipFuncSetAttribute(64 KiB LDS): %s", __func__, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid, Cin / Cfg::CC), dim3(NW * 64), lds, stream, dy, u, partial, F, H, W, Cin, ldy, TH);
    return GCPX_OK;
}

template <int NT, int CIT, int NW, bool BAL = false>
int launch_wc(const float* dy, const float* u, float* partial, int F, int H, int W, int Cin, int ldy, int grid, hipStream_t stream) {
    if (W >= 32 && W % 32 == 0) return launch_wc2<NT, CIT, NW, 32, BAL>(dy, u, partial, F, H, W, Cin, ldy, grid, stream);
    if (W == 16) return launch_wc2<NT, CIT, NW, 16, BAL>(dy, u, partial, F, H, W, Cin, ldy, grid, stream);
    if (W == 8) return launch_wc2<NT, CIT, NW, 8, BAL>(dy, u, partial, F, H, W, Cin, ldy, grid, stream);
    gcpx_set_error("gcpx_wgrad_conv3x3: unsupported width %d", W);
    return GCPX_ERR_UNSUPPORTED;
}

}  // namespace

// partial: [grid][N][9*Cin] (N = Cout rounded up to 16); every workgroup row is written (zeros if it had no tile)
extern "C" int gcpx_wgrad_conv3x3(const float* dy, int32_t ldy, const float* u, int32_t F, int32_t H, int32_t W, int32_t Cin,
                                  int32_t Cout, float* partial, int32_t grid, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(dy && u && partial && F > 0 && grid > 0, "bad arguments");
    GCPX_CHECK_ARG(ldy % 4 == 0 && Cin % 16 == 0, "ldy % 4, Cin % 16");
    const int NT = (Cout + 15) / 16;
    GCPX_CHECK_ARG(ldy >= NT * 16, "dy rows must hold Cout rounded up to 16 columns");
    int st = GCPX_ERR_UNSUPPORTED;
    if (NT == 7 && Cin == 16) st = launch_wc<7, 1, 4, true>(dy, u, partial, F, H, W, Cin, ldy, grid, stream);
    else if (NT == 1 && Cin % 32 == 0) st = launch_wc<1, 2, 4>(dy, u, partial, F, H, W, Cin, ldy, grid, stream);
    else if (NT == 1) st = launch_wc<1, 1, 3>(dy, u, partial, F, H, W, Cin, ldy, grid, stream);
    else if (NT == 2 && Cin % 32 == 0) st = launch_wc<2, 2, 4>(dy, u, partial, F, H, W, Cin, ldy, grid, stream);
    else if (NT == 4 && Cin % 32 == 0) st = launch_wc<4, 2, 4>(dy, u, partial, F, H, W, Cin, ldy, grid, stream);
    else gcpx_set_error("gcpx_wgrad_conv3x3: unsupported Cout=%d Cin=%d", Cout, Cin);
    if (st != GCPX_OK) return st;
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
