// Weight gradient of the decoder's 3x3 convs on the f16 matrix pipes of gfx950 with f32-equivalent arithmetic ("split-f16"):
//     dW[n][tap][ci] = sum over pixels p of dY[p][n] * U[p + tap][ci]
// (backward of the blox ConvDecoder blocks / gen_head called through /root/reference/gcp/prediction/models/tree/
//  tree_dense_rec.py:42; same contract, grid and partial layout as gcpx_wgrad_conv3x3 in wgrad_conv.hip, which is bound by
//  the f32 MFMA rate: 0.54 - 0.62 of 157 TFLOP/s).
//
// Both operands are data here, so both are split in the kernel: x = (x1 + x2) / 2^E, x1 = rn16(x 2^E), x2 = rn16(x 2^E - x1), and
// a product costs three v_mfma_f32_16x16x32_f16 (x2 y1 + x1 y2 + x1 y1, small terms first, f32 accumulate).  The MFMA k index
// walks PIXELS, so a lane's operand is 8 pixels of one channel, while the tensors are pixel-major (channels fastest).  The staged
// tiles stay pixel-major in LDS ([piece][16-channel tile][pixel][16] f16 planes, 8-byte writes of 4 channels) and the operand
// reads transpose: ds_read_b64_tr_b16 hands lane i channel i of four pixels whose addresses the 16-lane group supplies, so a tap
// is an address offset of whole pixels (no alignment cases, no shuffles).  A first version transposed 4 x 4 blocks in registers
// (DPP quad permutes) on the way into LDS: 2.3x the VALU instructions, 59 % of the LDS cycles lost to bank conflicts.
//
// Scales: the sum runs over every pixel of every frame, so what matters is the error relative to the largest terms.  Each
// workgroup keeps a running power-of-two scale per operand (from the largest value staged so far) and rescales its accumulators
// exactly when a larger value turns up; pieces of smaller tiles lose nothing that the f32 sum would keep.
#include "common.h"
#include "split_tr.h"

#include <type_traits>

extern "C" int gcpx_wgrad_conv3x3(const float* dy, int32_t ldy, const float* u, int32_t F, int32_t H, int32_t W, int32_t Cin,
                                  int32_t Cout, float* partial, int32_t grid, void* stream_);

namespace {

template <int NT, int CIT, int TW>
struct WSCfg {
    static constexpr int NW = 4, NTH = 256;
    static constexpr int N = NT * 16, CC = CIT * 16;
    static constexpr int TH = 64 / TW, RH = TH + 2, RW = TW + 2, RPX = RH * RW;
    static constexpr int PA = 64 * 16 + 16;                    // halfs per (piece, n-tile) plane of dY: [64 pixels][16 channels] + 32 B (the planes start 8 banks apart)
    static constexpr int PB = RPX * 16 + 16;                   // halfs per (piece, ci-tile) plane of U: [RH x RW pixels][16 channels] + 32 B
    static constexpr int A_HALFS = 2 * NT * PA, B_HALFS = 2 * CIT * PB;
    static constexpr int G = 9 * CIT, GPW = (G + NW - 1) / NW;
    static constexpr int NDS = NT;                             // dY staging slots per thread: slot s = n-tile s, 64 pixels x 4 float4
    static constexpr int UPS = (RPX + 63) / 64;                // U staging slots per ci-tile: 64 region pixels x 4 float4 each
    static constexpr int NUS = UPS * CIT;
    static constexpr int NS = NDS + NUS;
    static constexpr int LDS_BYTES = (A_HALFS + B_HALFS) * 2 + 64 + 2 * CC * 4;      // planes, tile maxima, optional operand affine
};

template <int NT, int CIT, int TW>
__global__ void __launch_bounds__(256, 2) wgrad_conv3x3_split_kernel(const float* __restrict__ dy, const float* __restrict__ u,
                                                                     float* __restrict__ partial, const int F, const int H, const int W,
                                                                     const int Cin, const int ldy, const int* __restrict__ fmap,
                                                                     const float* __restrict__ usc, const float* __restrict__ ush,
                                                                     float* __restrict__ bias_partial) {
    using Cfg = WSCfg<NT, CIT, TW>;
    constexpr int NW = Cfg::NW, N = Cfg::N, CC = Cfg::CC, TH = Cfg::TH, RH = Cfg::RH, RW = Cfg::RW, RPX = Cfg::RPX;
    constexpr int PA = Cfg::PA, PB = Cfg::PB, G = Cfg::G, GPW = Cfg::GPW, NDS = Cfg::NDS, UPS = Cfg::UPS, NUS = Cfg::NUS;
    extern __shared__ float4 smem4[];
    _Float16* sA = reinterpret_cast<_Float16*>(smem4);                     // [2][NT][PA]
    _Float16* sB = sA + Cfg::A_HALFS;                                      // [2][CIT][PB]
    float* red = reinterpret_cast<float*>(sB + Cfg::B_HALFS);              // [NW][2] tile maxima
    // usc / ush (optional): the operand is LeakyReLU(usc * u + ush) of the tensor at `u`, frame f of the operand = frame fmap[f] of that
    // tensor (the output head's input: the last decoder block's raw output at the matched nodes) — what gcpx_conv_stage would materialise.
    // The two vectors sit in LDS behind the maxima (in registers they would be the values that spill).
    float* uaff = red + 8;                                                 // [2][CC]
    if (usc) {
        if (threadIdx.x < 2 * CC) uaff[threadIdx.x] = (threadIdx.x < CC ? usc : ush)[blockIdx.y * CC + (threadIdx.x % CC)];
        __syncthreads();
    }

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ij = lane & 15, kq = lane >> 4;
    // the (tap, ci-tile) groups are dealt round-robin over the wavefronts; the deal is rotated per workgroup so that the wavefronts with one
    // group more land on different SIMDs of a CU that holds two workgroups
    const int wrot = __builtin_amdgcn_readfirstlane((wave + (int)blockIdx.x) & 3);
    const int ci0 = blockIdx.y * CC;
    const int ntx = W / TW, nty = H / TH;
    const int ntiles = F * nty * ntx;

    // BAL (output head: 7 n-tiles, 9 taps): the 63 (tap, n-tile) accumulator tiles are dealt 16 / 16 / 16 / 15 as in wgrad_conv.hip —
    //   rotated wavefront w < 3: taps 3w .. 3w+2 x n-tiles 0 .. 4, plus (tap 3w, n-tile 5);   3: n-tile 6 x 9 taps + n-tile 5 x taps {1,2,4,5,7,8}
    // (64 accumulator registers instead of 84: the generic deal spilled, and a scratch reload waits for every load in flight)
    constexpr bool BAL = (NT == 7 && CIT == 1);
    constexpr int NACC = BAL ? 16 : GPW * NT;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    // BAL + bias_partial: the bias gradient (column sums of dY over every pixel) comes out of the same staged tiles as two more
    // accumulator tiles per wavefront — dY against a fragment of ones — instead of a pass of its own over dY (2.35 GB at c2).
    // Rotated wavefront 0 / 1 / 2 / 3 sums n-tiles {0, 1} / {2, 3} / {4} / {5, 6}.
    f32x4 bacc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    const bool do_bias = BAL && bias_partial != nullptr;

    // ---- staging slots (the same for every tile; per-slot data is one register: anything spilled is reloaded with a wait for every
    //      load in flight).  A thread moves float4 (tid & 3) of pixel (tid >> 2) of a 16-channel tile: 64 contiguous bytes per pixel in
    //      global memory, 512 contiguous bytes per wavefront and piece in LDS ----
    const int sp = tid >> 2, sc4 = tid & 3;
    const unsigned dgoff0 = (unsigned)((((sp / TW) * W + (sp % TW)) * ldy + 4 * sc4) * 4);       // dY slot s = n-tile s: + 64 s bytes
    const int dloff0 = sp * 16 + 4 * sc4;                                                          //                       + s PA halfs
    // U slot (cit, k): region pixel r = sp + 64 k of ci-tile cit; row / column of r packed (r >= RPX: no such pixel)
    unsigned umeta[Cfg::UPS];
#pragma unroll
    for (int k = 0; k < UPS; ++k) {
        const int r = sp + 64 * k;
        umeta[k] = r < RPX ? (1u << 16) | ((unsigned)(r / RW) << 8) | (unsigned)(r % RW) : 0u;
    }
    float4 pre[Cfg::NS];
    unsigned okbits = 0;
    // loads of one tile: uniform tile data first, then one slot at a time (a slot's register is reloaded for the next tile as soon as its
    // values have been split, before the LDS writes, the barrier and the MFMA phase: the phase alone is shorter than a trip to HBM)
    const char *dyb = nullptr, *ub = nullptr;
    int ty0 = 0, tx0 = 0;
    auto tile_base = [&](const int tile) __attribute__((always_inline)) {
        const int tx = tile % ntx;
        const int t2 = tile / ntx;
        const int f = t2 / nty;
        ty0 = (t2 % nty) * TH; tx0 = tx * TW;
        dyb = reinterpret_cast<const char*>(dy + (((size_t)f * H + ty0) * W + tx0) * ldy);
        const int fu = fmap ? max(fmap[f], 0) : f;                       // (rows without a frame carry a zero gradient: any frame will do)
        ub = reinterpret_cast<const char*>(u + (size_t)fu * H * W * Cin + ci0);
    };
    auto load_d = [&](const int s) __attribute__((always_inline)) { pre[s] = gload4(dyb, dgoff0 + 64u * s); };
    auto load_u = [&](const int s) __attribute__((always_inline)) {
        // out-of-image positions (the conv's zero padding) are loaded from the clamped position and zeroed before use: the loads stay
        // unconditional
        const int cit = s / UPS, k = s % UPS;
        const int row = (umeta[k] >> 8) & 255, col = umeta[k] & 255;
        const int y = ty0 - 1 + row, x = tx0 - 1 + col;
        const bool ok = (umeta[k] >> 16) != 0 && y >= 0 && y < H && x >= 0 && x < W;
        const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
        pre[NDS + s] = gload4(ub, (unsigned)(((yc * W + xc) * Cin + 16 * cit + 4 * sc4) * 4));
        okbits = (okbits & ~(1u << s)) | (ok ? (1u << s) : 0u);
    };

    int ea = 0, eb = 0;                 // running scales of dY / U: staged values are multiplied by 2^ea / 2^eb
    bool have_a = false, have_b = false;
    const int my_first = wrot;          // first group of this wavefront

    // operand addresses of the transposing reads: lane (s = lane & 15, group kq) points at pixel 4 kq + (s >> 2) (first read; + 16: second)
    // of a 32-pixel k-step, channels 4 (s & 3) .. + 3.  Groups 0 / 1 (2 / 3) read 256 contiguous bytes: every bank once.
    const int rpix = 4 * kq + (ij >> 2), rch = 4 * (ij & 3);

    int tile = blockIdx.x;
    if (tile < ntiles) {
        tile_base(tile);
#pragma unroll
        for (int s = 0; s < NDS; ++s) load_d(s);
#pragma unroll
        for (int s = 0; s < NUS; ++s) load_u(s);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        // ---- largest |value| of the tile, per operand ----
        float ma = 0.f, mb = 0.f;
#pragma unroll
        for (int s = 0; s < NDS; ++s)
            ma = fmaxf(ma, fmaxf(fmaxf(fabsf(pre[s].x), fabsf(pre[s].y)), fmaxf(fabsf(pre[s].z), fabsf(pre[s].w))));
#pragma unroll
        for (int s = 0; s < NUS; ++s) {
            if (usc) {
                const float4 a_sc = *reinterpret_cast<const float4*>(uaff + 16 * (s / UPS) + 4 * sc4);
                const float4 a_sh = *reinterpret_cast<const float4*>(uaff + CC + 16 * (s / UPS) + 4 * sc4);
                float4 t = pre[NDS + s];
                t.x = lrelu(fmaf(t.x, a_sc.x, a_sh.x), 0.2f); t.y = lrelu(fmaf(t.y, a_sc.y, a_sh.y), 0.2f);
                t.z = lrelu(fmaf(t.z, a_sc.z, a_sh.z), 0.2f); t.w = lrelu(fmaf(t.w, a_sc.w, a_sh.w), 0.2f);
                pre[NDS + s] = t;
            }
            if (!((okbits >> s) & 1u)) pre[NDS + s] = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 v = pre[NDS + s];
            mb = fmaxf(mb, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        ma = wave_max_nonneg(ma);
        mb = wave_max_nonneg(mb);
        if (lane == 0) { red[2 * wave] = ma; red[2 * wave + 1] = mb; }
        __syncthreads();                                      // maxima visible; the previous tile's operand reads are done
        ma = fmaxf(fmaxf(red[0], red[2]), fmaxf(red[4], red[6]));
        mb = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
        ma = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ma)));
        mb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, mb)));
        float resc = 1.f;
        if (ma > 0.f) {
            const int e = max(-100, min(100, 14 + 127 - (int)((__float_as_uint(ma) >> 23) & 0xff)));      // ma 2^e in [2^14, 2^15)
            if (!have_a) { ea = e; have_a = true; }
            else if (e < ea) {
                resc *= __uint_as_float((unsigned)(127 + max(e - ea, -126)) << 23); ea = e;
                if constexpr (BAL) { bacc[0] *= resc; bacc[1] *= resc; }                                  // (the bias sums carry the dY scale only)
            }
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

        // ---- registers -> LDS: split, 8-byte stores into the pixel-major planes; the slot's register takes the next tile's load (the last
        //      tile loads itself again: the loads stay unconditional) ----
        tile_base(min(tile + (int)gridDim.x, ntiles - 1));
#pragma unroll
        for (int s = 0; s < NDS; ++s) {
            h4 p1, p2;
            split4(pre[s], sa, p1, p2);
            load_d(s);
            *reinterpret_cast<h4*>(sA + s * PA + dloff0) = p1;
            *reinterpret_cast<h4*>(sA + (NT + s) * PA + dloff0) = p2;
        }
#pragma unroll
        for (int s = 0; s < NUS; ++s) {
            const int cit = s / UPS, k = s % UPS;
            h4 p1, p2;
            split4(pre[NDS + s], sb, p1, p2);
            load_u(s);
            if (umeta[k] >> 16) {
                *reinterpret_cast<h4*>(sB + cit * PB + 64 * 16 * k + dloff0) = p1;
                *reinterpret_cast<h4*>(sB + (CIT + cit) * PB + 64 * 16 * k + dloff0) = p2;
            }
        }
        __syncthreads();

        // ---- MFMA phase: two k-steps of 32 pixels.  The deal of groups is compile-time per rotated wavefront index ----
        auto mm3 = [&](f32x4& c, const h8 a1, const h8 a2, const h8 b1, const h8 b2) __attribute__((always_inline)) {
            c = mfma32h(a2, b1, c);        // small terms first
            c = mfma32h(a1, b2, c);
            c = mfma32h(a1, b1, c);
        };
        auto phase = [&](auto wv_c) __attribute__((always_inline)) {
            constexpr int WV = decltype(wv_c)::value;
            const h8 one8 = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};
            __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
            for (int step = 0; step < 2; ++step) {
                // k slot (read r, group kq, j) = pixel pi = 4 kq + j + 16 r of the step's 32 pixels; the step covers tile rows 32 / TW * step ..
                const _Float16* ap = sA + (32 * step + rpix) * 16 + rch;
                auto lda = [&](const int nt, h8& a1, h8& a2) __attribute__((always_inline)) {
                    a1 = ds_tr8(ap + nt * PA, 16 * 16);
                    a2 = ds_tr8(ap + (NT + nt) * PA, 16 * 16);
                };
                // region pixel of k slot (r, kq, j) for tap (tdy, tdx): row (32 step + pi) / TW + tdy, column (32 step + pi) % TW + tdx
                const int p0 = 32 * step + rpix, p1 = p0 + 16;
                const int rb0 = ((p0 / TW) * RW + (p0 % TW)) * 16 + rch, rb1 = ((p1 / TW) * RW + (p1 % TW)) * 16 + rch;
                auto ldb = [&](const int cit, const int tdy, const int tdx, h8& b1, h8& b2) __attribute__((always_inline)) {
                    const _Float16* bp = sB + cit * PB + (tdy * RW + tdx) * 16;
                    const h4 l1 = ds_tr(bp + rb0), u1 = ds_tr(bp + rb1);
                    const h4 l2 = ds_tr(bp + CIT * PB + rb0), u2 = ds_tr(bp + CIT * PB + rb1);
                    b1 = __builtin_shufflevector(l1, u1, 0, 1, 2, 3, 4, 5, 6, 7);
                    b2 = __builtin_shufflevector(l2, u2, 0, 1, 2, 3, 4, 5, 6, 7);
                };
                if constexpr (BAL) {
                    if constexpr (WV < 3) {
                        h8 b1[3], b2[3];
#pragma unroll
                        for (int t = 0; t < 3; ++t) ldb(0, WV, t, b1[t], b2[t]);
#pragma unroll
                        for (int nt = 0; nt < 6; ++nt) {
                            h8 a1, a2;
                            lda(nt, a1, a2);
                            if (nt < 5) {
#pragma unroll
                                for (int t = 0; t < 3; ++t) mm3(acc[t * 5 + nt], a1, a2, b1[t], b2[t]);
                            } else {
                                mm3(acc[15], a1, a2, b1[0], b2[0]);
                            }
                            if (do_bias && nt >= 2 * WV && nt < 2 * WV + 2 && nt < 5) {
                                f32x4& bc = bacc[nt - 2 * WV];
                                bc = mfma32h(a2, one8, bc);
                                bc = mfma32h(a1, one8, bc);
                            }
                        }
                    } else {
                        h8 a51, a52, a61, a62;
                        lda(5, a51, a52);
                        lda(6, a61, a62);
                        if (do_bias) {
                            bacc[0] = mfma32h(a52, one8, bacc[0]); bacc[0] = mfma32h(a51, one8, bacc[0]);
                            bacc[1] = mfma32h(a62, one8, bacc[1]); bacc[1] = mfma32h(a61, one8, bacc[1]);
                        }
#pragma unroll
                        for (int tdy = 0; tdy < 3; ++tdy) {
                            h8 b1[3], b2[3];
#pragma unroll
                            for (int t = 0; t < 3; ++t) ldb(0, tdy, t, b1[t], b2[t]);
#pragma unroll
                            for (int t = 0; t < 3; ++t) {
                                mm3(acc[tdy * 3 + t], a61, a62, b1[t], b2[t]);
                                if (t != 0) mm3(acc[9 + tdy * 2 + t - 1], a51, a52, b1[t], b2[t]);
                            }
                        }
                    }
                } else {
                    // groups in chunks of three: their fragments stay in registers while the n-tiles pass by
                    static_for<0, (GPW + 2) / 3>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int c0 = decltype(cc)::value * 3;
                        h8 b1[3], b2[3];
                        static_for<0, 3>([&](auto gc) __attribute__((always_inline)) {
                            constexpr int g = c0 + decltype(gc)::value, gi = WV + g * NW;
                            if constexpr (g < GPW && gi < G) ldb(gi % CIT, (gi / CIT) / 3, (gi / CIT) % 3, b1[g - c0], b2[g - c0]);
                        });
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            h8 a1, a2;
                            lda(nt, a1, a2);
                            static_for<0, 3>([&](auto gc) __attribute__((always_inline)) {
                                constexpr int g = c0 + decltype(gc)::value;
                                if constexpr (g < GPW && WV + g * NW < G) mm3(acc[g * NT + nt], a1, a2, b1[g - c0], b2[g - c0]);
                            });
                        }
                        if constexpr (GPW * NT >= 20) __builtin_amdgcn_sched_barrier(0);     // keep the next chunk's fragment reads behind this chunk (registers)
                    });
                }
            }
            __builtin_amdgcn_s_setprio(0);
        };
        if (my_first == 0) phase(std::integral_constant<int, 0>{});
        else if (my_first == 1) phase(std::integral_constant<int, 1>{});
        else if (my_first == 2) phase(std::integral_constant<int, 2>{});
        else phase(std::integral_constant<int, 3>{});
    }

    // partial [blockIdx.x][n][K = 9*Cin]: lane holds n = nt*16 + 4*kq + reg, k = tap*Cin + ci0 + cit*16 + ij
    const int K = 9 * Cin;
    float* out = partial + (size_t)blockIdx.x * N * K;
    const float ia = __uint_as_float((unsigned)(127 - ea) << 23), ib = __uint_as_float((unsigned)(127 - eb) << 23);
    if constexpr (BAL) {
        if (do_bias && blockIdx.y == 0 && ij == 0) {          // every column of a bias tile holds the same sums: column 0 writes them
            const int nt0 = wrot < 3 ? 2 * wrot : 5;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (wrot != 2 || i == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bias_partial[(size_t)blockIdx.x * N + (nt0 + i) * 16 + 4 * kq + r] = bacc[i][r] * ia;
                }
        }
        auto store = [&](const f32x4& v, int nt, int tap) {
            const int k = tap * Cin + ci0 + ij;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(nt * 16 + 4 * kq + r) * K + k] = v[r] * ia * ib;
        };
        if (wrot < 3) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) store(acc[t * 5 + nt], nt, 3 * wrot + t);
            store(acc[15], 5, 3 * wrot);
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                store(acc[tap], 6, tap);
                if (tap % 3 != 0) store(acc[9 + (tap / 3) * 2 + (tap % 3) - 1], 5, tap);
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < GPW; ++g) {
            const int gi = wrot + g * NW;
            if (gi >= G) continue;
            const int tap = gi / CIT, cit = gi % CIT;
            const int k = tap * Cin + ci0 + cit * 16 + ij;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(nt * 16 + 4 * kq + r) * K + k] = acc[g * NT + nt][r] * ia * ib;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same weight gradient for an UPSAMPLING block with 16 output channels, reading the block's own sources: U[p][ci] = bilinear x2
// (align_corners = False) of the concatenated, normalised + activated low-resolution sources, formed in the kernel instead of being
// materialised by gcpx_conv_stage (additional_conv_layer at c2: 1.07 GB written by the stage kernel and read back here, per step).
// Per tile the workgroup first stages the LOW-resolution source region (TH/2 + 2 rows x TW/2 + 2 columns x 32 channels, border
// pixels replicated = the clamped indices of the interpolation; affine + LeakyReLU applied once per source value) as f32 in LDS; every
// thread then blends its region pixels of the operand from four LDS reads (weights 9/16, 3/16, 3/16, 1/16), and the rest — maxima,
// scales, split, planes, MFMA phase — is the kernel above with NT = 1, CIT = 2.
struct WsUpSrc {
    const float *p0, *sc0, *sh0, *p1, *sc1, *sh1;      // sources [F / fdiv][Hs][Ws][C], per-channel affine (or NULL)
    int C0, C1, fdiv0, fdiv1, act0, act1;
};

template <int TW>
__global__ void __launch_bounds__(256, 2) wgrad_conv3x3_split_up_kernel(const float* __restrict__ dy, const WsUpSrc us, float* __restrict__ partial,
                                                                        const int F, const int H, const int W, const int Cin, const int ldy) {
    constexpr int NT = 1, CIT = 2;
    using Cfg = WSCfg<NT, CIT, TW>;
    constexpr int NW = Cfg::NW, N = Cfg::N, CC = Cfg::CC, TH = Cfg::TH, RH = Cfg::RH, RW = Cfg::RW, RPX = Cfg::RPX;
    constexpr int PA = Cfg::PA, PB = Cfg::PB, G = Cfg::G, GPW = Cfg::GPW, NDS = Cfg::NDS, UPS = Cfg::UPS, NUS = Cfg::NUS;
    constexpr int SRH = TH / 2 + 2, SRW = TW / 2 + 2, SPX = SRH * SRW, NSRC = (SPX + 31) / 32;
    extern __shared__ float4 smem4[];
    _Float16* sA = reinterpret_cast<_Float16*>(smem4);                     // [2][NT][PA]
    _Float16* sB = sA + Cfg::A_HALFS;                                      // [2][CIT][PB]
    float* red = reinterpret_cast<float*>(sB + Cfg::B_HALFS);              // [NW][2] tile maxima
    float4* sS = reinterpret_cast<float4*>(reinterpret_cast<char*>(smem4) + Cfg::LDS_BYTES);      // [SPX][8] source region (f32, 32 channels)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ij = lane & 15, kq = lane >> 4;
    const int wrot = __builtin_amdgcn_readfirstlane((wave + (int)blockIdx.x) & 3);
    const int ci0 = blockIdx.y * CC;
    const int ntx = W / TW, nty = H / TH;
    const int ntiles = F * nty * ntx;
    const int Hs = H / 2, Ws = W / 2;

    constexpr int NACC = GPW * NT;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};

    // dY staging slots as above; source slots: thread (sp_s = tid >> 3, sc8 = tid & 7) moves float4 sc8 (of the 32 channels) of region
    // pixels sp_s + 32 k
    const int sp = tid >> 2, sc4 = tid & 3;
    const unsigned dgoff0 = (unsigned)((((sp / TW) * W + (sp % TW)) * ldy + 4 * sc4) * 4);
    const int dloff0 = sp * 16 + 4 * sc4;
    const int sp_s = tid >> 3, sc8 = tid & 7;
    const int cg = ci0 + 4 * sc8;                                          // this thread's source channels
    const bool first = cg < us.C0;
    const float* sbase = first ? us.p0 : us.p1;
    const int sC = first ? us.C0 : us.C1, cl = first ? cg : cg - us.C0, fdiv = first ? us.fdiv0 : us.fdiv1;
    const float* scp = first ? us.sc0 : us.sc1;
    const float* shp = first ? us.sh0 : us.sh1;
    const bool aff = scp != nullptr, lre = (first ? us.act0 : us.act1) == GCPX_ACT_LRELU;
    float4 a_sc = make_float4(1.f, 1.f, 1.f, 1.f), a_sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (aff) { a_sc = *reinterpret_cast<const float4*>(scp + cl); a_sh = *reinterpret_cast<const float4*>(shp + cl); }
    unsigned smeta[NSRC];                                                  // region pixel -> (row, column), or no such pixel
#pragma unroll
    for (int k = 0; k < NSRC; ++k) {
        const int r = sp_s + 32 * k;
        smeta[k] = r < SPX ? (1u << 16) | ((unsigned)(r / SRW) << 8) | (unsigned)(r % SRW) : 0u;
    }
    // operand region pixels of this thread (as umeta above)
    unsigned umeta[Cfg::UPS];
#pragma unroll
    for (int k = 0; k < UPS; ++k) {
        const int r = sp + 64 * k;
        umeta[k] = r < RPX ? (1u << 16) | ((unsigned)(r / RW) << 8) | (unsigned)(r % RW) : 0u;
    }
    // A tile's origin is even in both directions, so which two region rows / columns an operand pixel interpolates between and with
    // which weights depends on the slot alone: the LDS offset of its first source value and the four products of the bilinear weights
    // are formed once (v = w00 a + w01 b + w10 c + w11 d: one multiply + three fma per value; 0.75 goes to the nearer source pixel)
    int uq[NUS];
    float uw[Cfg::UPS][4];
#pragma unroll
    for (int k = 0; k < UPS; ++k) {
        const int row = (umeta[k] >> 8) & 255, col = umeta[k] & 255;        // operand pixel (row - 1, col - 1) relative to the tile
        const float wy0 = (row & 1) ? 0.25f : 0.75f, wx0 = (col & 1) ? 0.25f : 0.75f;     // y = origin - 1 + row is odd for even row
        uw[k][0] = wy0 * wx0; uw[k][1] = wy0 * (1.f - wx0); uw[k][2] = (1.f - wy0) * wx0; uw[k][3] = (1.f - wy0) * (1.f - wx0);
#pragma unroll
        for (int cit = 0; cit < CIT; ++cit) uq[cit * UPS + k] = ((row >> 1) * SRW + (col >> 1)) * 8 + cit * 4 + sc4;
    }
    float4 pre[NDS + NSRC];
    const char* dyb = nullptr;
    const float* sfb = nullptr;
    int ty0 = 0, tx0 = 0;
    auto tile_base = [&](const int tile) __attribute__((always_inline)) {
        const int tx = tile % ntx;
        const int t2 = tile / ntx;
        const int f = t2 / nty;
        ty0 = (t2 % nty) * TH; tx0 = tx * TW;
        dyb = reinterpret_cast<const char*>(dy + (((size_t)f * H + ty0) * W + tx0) * ldy);
        sfb = sbase + (size_t)(f / fdiv) * Hs * Ws * sC + cl;
    };
    auto load_d = [&](const int s) __attribute__((always_inline)) { pre[s] = gload4(dyb, dgoff0 + 64u * s); };
    auto load_s = [&](const int k) __attribute__((always_inline)) {
        // clamped source pixel: the border replication of the interpolation's index clamp (pixels past the region are never read)
        const int row = (smeta[k] >> 8) & 255, col = smeta[k] & 255;
        const int ys = min(max(ty0 / 2 - 1 + row, 0), Hs - 1), xs = min(max(tx0 / 2 - 1 + col, 0), Ws - 1);
        pre[NDS + k] = gload4(reinterpret_cast<const char*>(sfb), (unsigned)((ys * Ws + xs) * sC * 4));
    };

    int ea = 0, eb = 0;
    bool have_a = false, have_b = false;
    const int my_first = wrot;
    const int rpix = 4 * kq + (ij >> 2), rch = 4 * (ij & 3);

    int tile = blockIdx.x;
    if (tile < ntiles) {
        tile_base(tile);
#pragma unroll
        for (int s = 0; s < NDS; ++s) load_d(s);
#pragma unroll
        for (int k = 0; k < NSRC; ++k) load_s(k);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        const int cty0 = ty0, ctx0 = tx0;                       // (tile_base below moves on to the next tile)
        // ---- source region -> LDS (affine + activation once per value); the registers take the next tile's loads at once ----
        tile_base(min(tile + (int)gridDim.x, ntiles - 1));
#pragma unroll
        for (int k = 0; k < NSRC; ++k) {
            float4 t = pre[NDS + k];
            if (aff) { t.x = fmaf(t.x, a_sc.x, a_sh.x); t.y = fmaf(t.y, a_sc.y, a_sh.y); t.z = fmaf(t.z, a_sc.z, a_sh.z); t.w = fmaf(t.w, a_sc.w, a_sh.w); }
            if (lre) { t.x = lrelu(t.x, 0.2f); t.y = lrelu(t.y, 0.2f); t.z = lrelu(t.z, 0.2f); t.w = lrelu(t.w, 0.2f); }
            load_s(k);
            if (smeta[k] >> 16) sS[(sp_s + 32 * k) * 8 + sc8] = t;
        }
        __syncthreads();                                      // source region complete (and the previous tile's operand reads are done)
        // ---- this thread's operand pixels: bilinear x2 from the region; positions outside the image are the conv's zero padding ----
        float4 uv[NUS];
#pragma unroll
        for (int s = 0; s < NUS; ++s) {
            const int k = s % UPS;
            const int row = (umeta[k] >> 8) & 255, col = umeta[k] & 255;
            const int y = cty0 - 1 + row, x = ctx0 - 1 + col;
            uv[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((umeta[k] >> 16) != 0 && y >= 0 && y < H && x >= 0 && x < W) {
                const float4* q0 = sS + uq[s];
                const float4 v00 = q0[0], v01 = q0[8], v10 = q0[SRW * 8], v11 = q0[SRW * 8 + 8];
                const float w00 = uw[k][0], w01 = uw[k][1], w10 = uw[k][2], w11 = uw[k][3];
                float4 r_;
                r_.x = fmaf(w11, v11.x, fmaf(w10, v10.x, fmaf(w01, v01.x, w00 * v00.x)));
                r_.y = fmaf(w11, v11.y, fmaf(w10, v10.y, fmaf(w01, v01.y, w00 * v00.y)));
                r_.z = fmaf(w11, v11.z, fmaf(w10, v10.z, fmaf(w01, v01.z, w00 * v00.z)));
                r_.w = fmaf(w11, v11.w, fmaf(w10, v10.w, fmaf(w01, v01.w, w00 * v00.w)));
                uv[s] = r_;
            }
        }
        // ---- largest |value| of the tile, per operand ----
        float ma = 0.f, mb = 0.f;
#pragma unroll
        for (int s = 0; s < NDS; ++s) { ma = vmax3abs(ma, pre[s].x, pre[s].y); ma = vmax3abs(ma, pre[s].z, pre[s].w); }
#pragma unroll
        for (int s = 0; s < NUS; ++s) { mb = vmax3abs(mb, uv[s].x, uv[s].y); mb = vmax3abs(mb, uv[s].z, uv[s].w); }
        ma = wave_max_nonneg(ma);
        mb = wave_max_nonneg(mb);
        if (lane == 0) { red[2 * wave] = ma; red[2 * wave + 1] = mb; }
        __syncthreads();
        ma = fmaxf(fmaxf(red[0], red[2]), fmaxf(red[4], red[6]));
        mb = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
        ma = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ma)));
        mb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, mb)));
        float resc = 1.f;
        if (ma > 0.f) {
            const int e = max(-100, min(100, 14 + 127 - (int)((__float_as_uint(ma) >> 23) & 0xff)));
            if (!have_a) { ea = e; have_a = true; }
            else if (e < ea) { resc *= __uint_as_float((unsigned)(127 + max(e - ea, -126)) << 23); ea = e; }
        }
        if (mb > 0.f) {
            const int e = max(-100, min(100, 14 + 127 - (int)((__float_as_uint(mb) >> 23) & 0xff)));
            if (!have_b) { eb = e; have_b = true; }
            else if (e < eb) { resc *= __uint_as_float((unsigned)(127 + max(e - eb, -126)) << 23); eb = e; }
        }
        if (resc != 1.f) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] *= resc;
        }
        const float sa = __uint_as_float((unsigned)(127 + ea) << 23), sb = __uint_as_float((unsigned)(127 + eb) << 23);
#pragma unroll
        for (int s = 0; s < NDS; ++s) {
            h4 p1, p2;
            split4(pre[s], sa, p1, p2);
            load_d(s);
            *reinterpret_cast<h4*>(sA + s * PA + dloff0) = p1;
            *reinterpret_cast<h4*>(sA + (NT + s) * PA + dloff0) = p2;
        }
#pragma unroll
        for (int s = 0; s < NUS; ++s) {
            const int cit = s / UPS, k = s % UPS;
            h4 p1, p2;
            split4(uv[s], sb, p1, p2);
            if (umeta[k] >> 16) {
                *reinterpret_cast<h4*>(sB + cit * PB + 64 * 16 * k + dloff0) = p1;
                *reinterpret_cast<h4*>(sB + (CIT + cit) * PB + 64 * 16 * k + dloff0) = p2;
            }
        }
        __syncthreads();

        // ---- MFMA phase (as above, generic deal with NT = 1) ----
        auto mm3 = [&](f32x4& c, const h8 a1, const h8 a2, const h8 b1, const h8 b2) __attribute__((always_inline)) {
            c = mfma32h(a2, b1, c);
            c = mfma32h(a1, b2, c);
            c = mfma32h(a1, b1, c);
        };
        auto phase = [&](auto wv_c) __attribute__((always_inline)) {
            constexpr int WV = decltype(wv_c)::value;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
            for (int step = 0; step < 2; ++step) {
                const _Float16* ap = sA + (32 * step + rpix) * 16 + rch;
                const h8 a1 = ds_tr8(ap, 16 * 16), a2 = ds_tr8(ap + NT * PA, 16 * 16);
                const int p0 = 32 * step + rpix, p1 = p0 + 16;
                const int rb0 = ((p0 / TW) * RW + (p0 % TW)) * 16 + rch, rb1 = ((p1 / TW) * RW + (p1 % TW)) * 16 + rch;
                static_for<0, GPW>([&](auto gc) __attribute__((always_inline)) {
                    constexpr int g = decltype(gc)::value, gi = WV + g * NW;
                    if constexpr (gi < G) {
                        constexpr int cit = gi % CIT, tdy = (gi / CIT) / 3, tdx = (gi / CIT) % 3;
                        const _Float16* bp = sB + cit * PB + (tdy * RW + tdx) * 16;
                        const h4 l1 = ds_tr(bp + rb0), u1 = ds_tr(bp + rb1);
                        const h4 l2 = ds_tr(bp + CIT * PB + rb0), u2 = ds_tr(bp + CIT * PB + rb1);
                        const h8 b1 = __builtin_shufflevector(l1, u1, 0, 1, 2, 3, 4, 5, 6, 7);
                        const h8 b2 = __builtin_shufflevector(l2, u2, 0, 1, 2, 3, 4, 5, 6, 7);
                        mm3(acc[g], a1, a2, b1, b2);
                    }
                });
            }
            __builtin_amdgcn_s_setprio(0);
        };
        if (my_first == 0) phase(std::integral_constant<int, 0>{});
        else if (my_first == 1) phase(std::integral_constant<int, 1>{});
        else if (my_first == 2) phase(std::integral_constant<int, 2>{});
        else phase(std::integral_constant<int, 3>{});
    }

    const int K = 9 * Cin;
    float* out = partial + (size_t)blockIdx.x * N * K;
    const float ia = __uint_as_float((unsigned)(127 - ea) << 23), ib = __uint_as_float((unsigned)(127 - eb) << 23);
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const int gi = wrot + g * NW;
        if (gi >= G) continue;
        const int tap = gi / CIT, cit = gi % CIT;
        const int k = tap * Cin + ci0 + cit * 16 + ij;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(4 * kq + r) * K + k] = acc[g][r] * ia * ib;
    }
}

template <int TW>
int launch_ws_up(const float* dy, const WsUpSrc& us, float* partial, int F, int H, int W, int Cin, int ldy, int grid, hipStream_t stream) {
    using Cfg = WSCfg<1, 2, TW>;
    constexpr int SPX = (Cfg::TH / 2 + 2) * (TW / 2 + 2);
    constexpr int LDS = Cfg::LDS_BYTES + SPX * 128;
    static_assert(LDS <= 64 * 1024, "operand planes + source region must fit 64 KiB");
    if (W % TW || H % Cfg::TH) return GCPX_ERR_UNSUPPORTED;
    auto kern = wgrad_conv3x3_split_up_kernel<TW>;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) {
            gcpx_set_error("%s: hipFuncSetAttribute(64 KiB LDS): %s", __func__, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid, Cin / Cfg::CC), dim3(256), LDS, stream, dy, us, partial, F, H, W, Cin, ldy);
    return GCPX_OK;
}

template <int NT, int CIT, int TW>
int launch_ws2(const float* dy, const float* u, float* partial, int F, int H, int W, int Cin, int ldy, int grid, hipStream_t stream,
               const int* fmap = nullptr, const float* usc = nullptr, const float* ush = nullptr, float* bias_partial = nullptr) {
    using Cfg = WSCfg<NT, CIT, TW>;
    if (W % TW || H % Cfg::TH) return GCPX_ERR_UNSUPPORTED;
    auto kern = wgrad_conv3x3_split_kernel<NT, CIT, TW>;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        if (e != hipSuccess) {
            gcpx_set_error("%s: hipFuncSetAttribute(64 KiB LDS): %s", __func__, hipGetErrorString(e));
            return GCPX_ERR_HIP;
        }
        attr_set = true;
    }
    static_assert(Cfg::LDS_BYTES <= 64 * 1024, "operand planes of one tile must fit 64 KiB");
    hipLaunchKernelGGL(kern, dim3(grid, Cin / Cfg::CC), dim3(256), Cfg::LDS_BYTES, stream, dy, u, partial, F, H, W, Cin, ldy, fmap, usc, ush, bias_partial);
    return GCPX_OK;
}

template <int NT, int CIT>
int launch_ws(const float* dy, const float* u, float* partial, int F, int H, int W, int Cin, int ldy, int grid, hipStream_t stream,
              const int* fmap = nullptr, const float* usc = nullptr, const float* ush = nullptr, float* bias_partial = nullptr) {
    const bool tall = getenv("GCPX_WS_TW32") == nullptr;             // 4 x 16 tiles (see launch_ws_up's caller): the head's weight gradient 794-836 -> 759-775 us
    if (W >= 32 && W % 32 == 0 && !(tall && H % 4 == 0)) return launch_ws2<NT, CIT, 32>(dy, u, partial, F, H, W, Cin, ldy, grid, stream, fmap, usc, ush, bias_partial);
    if (W % 16 == 0 && (W == 16 || tall)) return launch_ws2<NT, CIT, 16>(dy, u, partial, F, H, W, Cin, ldy, grid, stream, fmap, usc, ush, bias_partial);
    if (W == 8) return launch_ws2<NT, CIT, 8>(dy, u, partial, F, H, W, Cin, ldy, grid, stream, fmap, usc, ush, bias_partial);
    return GCPX_ERR_UNSUPPORTED;
}

}  // namespace

// Same arguments, grid and partial layout as gcpx_wgrad_conv3x3 (the exact f32 kernel, which also takes the shapes this one does not
// cover; the input-channel chunk per workgroup row — 16 for the 112-slot head, 32 otherwise — is the same in both).
extern "C" int gcpx_wgrad_conv3x3_split(const float* dy, int32_t ldy, const float* u, int32_t F, int32_t H, int32_t W, int32_t Cin,
                                        int32_t Cout, float* partial, int32_t grid, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(dy && u && partial && F > 0 && grid > 0, "bad arguments");
    GCPX_CHECK_ARG(ldy % 4 == 0 && Cin % 16 == 0, "ldy % 4, Cin % 16");
    const int NT = (Cout + 15) / 16;
    GCPX_CHECK_ARG(ldy >= NT * 16, "dy rows must hold Cout rounded up to 16 columns");
    int st = GCPX_ERR_UNSUPPORTED;
    if (NT == 7 && Cin == 16) st = launch_ws<7, 1>(dy, u, partial, F, H, W, Cin, ldy, grid, stream);
    else if (NT == 1 && Cin % 32 == 0) st = launch_ws<1, 2>(dy, u, partial, F, H, W, Cin, ldy, grid, stream);
    else if (NT == 2 && Cin % 32 == 0) st = launch_ws<2, 2>(dy, u, partial, F, H, W, Cin, ldy, grid, stream);
    else if (NT == 4 && Cin % 32 == 0) st = launch_ws<4, 2>(dy, u, partial, F, H, W, Cin, ldy, grid, stream);
    if (st == GCPX_ERR_UNSUPPORTED) return gcpx_wgrad_conv3x3(dy, ldy, u, F, H, W, Cin, Cout, partial, grid, stream_);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

// The same for an upsampling block with 16 output channels, reading the block's own (low-resolution) sources: `a` is the block's forward
// descriptor as gcpx_conv_stage takes it (src / nsrc / F / Hin / Win / Hout / Wout / Cin, upsample = 1; no src_row_map), the operand
// tensor of gcpx_wgrad_conv3x3_split is never materialised.  GCPX_ERR_UNSUPPORTED when the shape has no fused form (the caller then
// runs gcpx_conv_stage + gcpx_wgrad_conv3x3_split): 16 output channels, Cin a multiple of 32, 16-channel groups that do not straddle
// the two sources, W in {8, 16, 32k}.
extern "C" int gcpx_wgrad_conv3x3_split_up(const float* dy, int32_t ldy, const gcpx_conv_args* a, int32_t Cout, float* partial, int32_t grid,
                                           void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(dy && a && partial && grid > 0 && a->F > 0, "bad arguments");
    GCPX_CHECK_ARG(ldy % 4 == 0 && ldy >= 16, "dy rows must hold 16 columns");
    const bool fits = a->upsample == 1 && !a->src_row_map && (a->nsrc == 1 || a->nsrc == 2) && Cout <= 16 && a->Cin % 32 == 0 &&
                      a->src[0].C % 16 == 0 && (a->nsrc == 1 || a->src[1].C % 16 == 0) &&
                      a->Cin == a->src[0].C + (a->nsrc == 2 ? a->src[1].C : 0) && a->Hout == 2 * a->Hin && a->Wout == 2 * a->Win &&
                      a->src[0].frame_div >= 1 && (a->nsrc == 1 || a->src[1].frame_div >= 1);
    if (!fits) return GCPX_ERR_UNSUPPORTED;
    WsUpSrc us;
    us.p0 = a->src[0].ptr; us.sc0 = a->src[0].scale; us.sh0 = a->src[0].shift; us.C0 = a->src[0].C; us.fdiv0 = a->src[0].frame_div; us.act0 = a->src[0].act;
    const gcpx_conv_src& s1 = a->src[a->nsrc == 2 ? 1 : 0];
    us.p1 = s1.ptr; us.sc1 = s1.scale; us.sh1 = s1.shift; us.C1 = s1.C; us.fdiv1 = s1.frame_div; us.act1 = s1.act;
    const int H = a->Hout, W = a->Wout;
    int st = GCPX_ERR_UNSUPPORTED;
    // 64-pixel tiles as 4 rows x 16 columns wherever the image allows it: the interpolated region with its halo is 6 x 18 = 108 pixels
    // instead of 4 x 34 = 136 for a 2 x 32 tile (additional_conv_layer at c2: 725 -> 630-660 us, pyramid-0: 172 -> 150; GCPX_WS_TW32=1: wide tiles)
    const bool up_tall = getenv("GCPX_WS_TW32") == nullptr;                // (read per call: the tests run both tile shapes in one process)
    if (W >= 32 && W % 32 == 0 && !(up_tall && H % 4 == 0)) st = launch_ws_up<32>(dy, us, partial, a->F, H, W, a->Cin, ldy, grid, stream);
    else if (W % 16 == 0 && (W == 16 || up_tall)) st = launch_ws_up<16>(dy, us, partial, a->F, H, W, a->Cin, ldy, grid, stream);
    else if (W == 8) st = launch_ws_up<8>(dy, us, partial, a->F, H, W, a->Cin, ldy, grid, stream);
    if (st != GCPX_OK) return st;
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

// The same for a NON-upsampling conv whose operand is LeakyReLU(scale * x + shift) of a raw tensor x [Fx][H][W][Cin] read through a frame
// map (operand frame f = frame frame_map[f] of x; negative entries must carry a zero dy): the output head's weight gradient reads the
// last decoder block's raw output at the matched nodes directly — gcpx_conv_stage's gathered copy is never written.  frame_map /
// scale + shift may be NULL.  bias_partial (optional, the 112-column head form): [grid][112] per-workgroup column sums of dy — the bias
// gradient out of the same staged tiles (two more accumulator tiles per wavefront against a fragment of ones) instead of a pass of its
// own over dy.  GCPX_ERR_UNSUPPORTED (nothing launched) for shapes without a split form.
extern "C" int gcpx_wgrad_conv3x3_split_src(const float* dy, int32_t ldy, const float* x, const int32_t* frame_map, const float* scale,
                                            const float* shift, int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t Cout, float* partial,
                                            float* bias_partial, int32_t grid, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(dy && x && partial && F > 0 && grid > 0, "bad arguments");
    GCPX_CHECK_ARG(ldy % 4 == 0 && Cin % 16 == 0 && (scale == nullptr) == (shift == nullptr), "ldy % 4, Cin % 16, scale and shift together");
    const int NT = (Cout + 15) / 16;
    GCPX_CHECK_ARG(ldy >= NT * 16, "dy rows must hold Cout rounded up to 16 columns");
    int st = GCPX_ERR_UNSUPPORTED;
    GCPX_CHECK_ARG(!bias_partial || (NT == 7 && Cin == 16), "bias_partial: the 112-column head form only");
    if (NT == 7 && Cin == 16) st = launch_ws<7, 1>(dy, x, partial, F, H, W, Cin, ldy, grid, stream, frame_map, scale, shift, bias_partial);
    // the head's 80 leading slots only (the adaptive model's mixture-mean gradient: the log-scale slots behind them are zero, gcpx_dlm_mean_bwd)
    else if (NT == 5 && Cin == 16) st = launch_ws<5, 1>(dy, x, partial, F, H, W, Cin, ldy, grid, stream, frame_map, scale, shift);
    else if (NT == 1 && Cin % 32 == 0) st = launch_ws<1, 2>(dy, x, partial, F, H, W, Cin, ldy, grid, stream, frame_map, scale, shift);
    else if (NT == 2 && Cin % 32 == 0) st = launch_ws<2, 2>(dy, x, partial, F, H, W, Cin, ldy, grid, stream, frame_map, scale, shift);
    else if (NT == 4 && Cin % 32 == 0) st = launch_ws<4, 2>(dy, x, partial, F, H, W, Cin, ldy, grid, stream, frame_map, scale, shift);
    if (st != GCPX_OK) return st;
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
