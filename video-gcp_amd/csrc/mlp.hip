// Fused Predictor MLP on gfx950: input(in->mid, LReLU), n_mid x (mid->mid, GroupNorm(8), LReLU), head(mid->out)
// with an optional reparametrised-Gaussian epilogue.  One 256-thread workgroup carries 16 rows through every
// layer: the four wavefronts split the output columns of each layer, hidden activations live in LDS, GroupNorm
// statistics are reduced with in-lane adds + two 4-lane-column shuffles (a group = one 16-column MFMA tile).
// The (wide) head layer can additionally be split over blockIdx.y; the cheap hidden layers are then recomputed.
//
// Replaces blox.torch.subnetworks.Predictor (absent; spec in DESIGN.md) at the call sites
//   /root/reference/gcp/prediction/models/tree/tree_module.py:77   (prior p(z | e_l, e_r))
//   /root/reference/gcp/prediction/models/tree/inference.py:27-35  (gather e_tilde + posterior q)
//   /root/reference/gcp/prediction/models/tree/tree_module.py:79-94 (sample / reparametrize)
//   /root/reference/gcp/prediction/models/tree/tree_module.py:105  (MLP LSTM initialiser)
//   misc.py:48, frame_binding.py:71, base_gcp.py:256, inverse_mdl.py:126, cost_mdl.py:63 (heads).
#include "common.h"
#include "gemm_tile.h"

#include <cstdlib>

bool gcpx_gemm_tile_plan(const gcpx_gemm_args* a, int* pr, int* cr, int* ks, int* gx, int* gy, int* gz);      // gemm.hip

namespace {

// one workgroup = 16 rows (block bx) of one Predictor; by / ny: this workgroup's share of the head's column tiles
// LEAN: the launch brings more workgroups than CUs, so it is throughput-bound: smaller load sets and no cross-layer weight prefetch
// keep the kernel at <= 128 registers (two workgroups per CU) instead of 256 (one) — prior + posterior at 1024 rows ran 49 us with
// the deep pipeline (512 workgroups, two rounds) against 20 us at 16 rows.
template <int MID, bool LEAN>
__device__ __forceinline__ void mlp_rows(const gcpx_mlp_args& a, const int bx, const int by, const int ny, float* hid) {
    constexpr int NTM = MID / 16;                    // column tiles of a hidden layer
    constexpr int NW = NTM >= 4 ? 4 : NTM;           // wavefronts that own hidden-layer tiles
    constexpr int TPW = NTM / NW;                    // tiles per wavefront
    constexpr int PITCH = MID + 4;
    constexpr int CPG = MID / 8;                     // channels per GroupNorm group (gn_groups = 8)
    static_assert(CPG == 4 || CPG == 16, "GroupNorm group must be one lane (4) or one 16-channel tile");

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int r = bx * 16 + j;
    const bool rv = r < a.M;
    const int rs = rv ? r : 0;
    const int rb = rs / a.rpb, rj = rs % a.rpb;
    const float slope = a.lrelu_slope;
    const bool owner = wave < NW;
    const int nt0 = wave * TPW;

    f32x4 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0, 0, 0, 0};

    // weights of the first hidden layer: requested before anything else, they arrive during the input layer
    auto load_hidden = [&](const int l, float4 (&w)[NTM][TPW]) {
        const float4* wbase = reinterpret_cast<const float4*>(a.w_mid) + ((size_t)l * NTM * NTM + nt0) * 64 + lane;
#pragma unroll
        for (int kg = 0; kg < NTM; ++kg)
#pragma unroll
            for (int t = 0; t < TPW; ++t) w[kg][t] = wbase[(kg * NTM + t) * 64];
    };
    float4 wA[NTM][TPW], wB[NTM][TPW];
    if (!LEAN && owner && a.n_mid > 0) load_hidden(0, wA);

    // ---- input layer: gathered global sources ----
    if (owner) {
        const float4* wbase = reinterpret_cast<const float4*>(a.w_in) + (size_t)nt0 * 64 + lane;
        int kg0 = 0;
        for (int s = 0; s < a.nsrc; ++s) {
            const gcpx_row_src src = a.src[s];
            bool ok = rv;
            size_t off;
            if (src.rowidx) {
                off = (size_t)src.rowidx[rs] * src.sr;
            } else {
                const int jj = rj + src.shift;
                ok = ok && jj >= 0 && jj < a.rpb;
                off = (size_t)rb * src.sb + (size_t)(ok ? jj : 0) * src.sr;
            }
            const float* bp = src.ptr + off + q * 4;
            const float mask = ok ? 1.f : 0.f;
            const int nkg = src.width / 16;
            const bool xf = src.scale || src.act;
            // Every load of this layer is independent of the MFMAs, and one workgroup is all there is on its CU: what bounds the
            // layer is the number of DEPENDENT global round trips.  Sets of UKI k-groups in ping-pong (explicit register sets): the
            // loads of one set are in flight during the MFMAs of the other (in_dim = 384: 3 sets instead of 6 serial batches).
            constexpr int UKI = LEAN ? 4 : 8;
            auto load_set = [&](const int kg, float4 (&b)[UKI], float4 (&w)[UKI][TPW]) {
#pragma unroll
                for (int u = 0; u < UKI; ++u) {
                    const int k = (kg + u < nkg) ? kg + u : nkg - 1;
                    b[u] = *reinterpret_cast<const float4*>(bp + k * 16);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) w[u][t] = wbase[((size_t)(kg0 + k) * NTM + t) * 64];
                }
            };
            auto mfma_set = [&](const int kg, float4 (&b)[UKI], float4 (&w)[UKI][TPW]) {
#pragma unroll
                for (int u = 0; u < UKI; ++u) {
                    if (kg + u < nkg) {
                        float4 bb = b[u];
                        if (xf) bb = affine_act4(bb, src.scale, src.shiftv, ((kg + u) * 16 + q * 4) & (src.cmod - 1), src.act);
                        bb.x *= mask; bb.y *= mask; bb.z *= mask; bb.w *= mask;
#pragma unroll
                        for (int t = 0; t < TPW; ++t) {
                            acc[t] = mfma16(w[u][t].x, bb.x, acc[t]);
                            acc[t] = mfma16(w[u][t].y, bb.y, acc[t]);
                            acc[t] = mfma16(w[u][t].z, bb.z, acc[t]);
                            acc[t] = mfma16(w[u][t].w, bb.w, acc[t]);
                        }
                    }
                }
            };
            float4 b0[UKI], w0[UKI][TPW], b1[UKI], w1[UKI][TPW];
            int kg = 0;
            load_set(0, b0, w0);
            while (true) {
                const int kn = kg + UKI;
                if (kn < nkg) load_set(kn, b1, w1);
                mfma_set(kg, b0, w0);
                if (kn >= nkg) break;
                kg = kn + UKI;
                if (kg < nkg) load_set(kg, b0, w0);
                mfma_set(kn, b1, w1);
                if (kg >= nkg) break;
            }
            kg0 += nkg;
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int c = (nt0 + t) * 16 + q * 4;
            const float4 bv = *reinterpret_cast<const float4*>(a.b_in + c);
            float4 v = make_float4(lrelu(acc[t][0] + bv.x, slope), lrelu(acc[t][1] + bv.y, slope),
                                   lrelu(acc[t][2] + bv.z, slope), lrelu(acc[t][3] + bv.w, slope));
            *reinterpret_cast<float4*>(hid + j * PITCH + c) = v;
            if (a.save && rv && by == 0) *reinterpret_cast<float4*>(a.save + (size_t)r * MID + c) = v;
        }
    }
    __syncthreads();

    // ---- hidden layers: mid -> mid, GroupNorm, LReLU ----
    // The weights of layer l + 1 are requested while layer l is still being computed (two explicit register sets), so a hidden
    // layer costs its MFMAs + GroupNorm + one barrier instead of a global round trip on top.
    int cur = 0;
    auto hidden = [&](const int l, float4 (&w)[NTM][TPW], float4 (&wnext)[NTM][TPW]) {
        const float* hin = hid + cur * 16 * PITCH;
        float* hout = hid + (cur ^ 1) * 16 * PITCH;
        if (owner) {
            float4 b[NTM];
#pragma unroll
            for (int kg = 0; kg < NTM; ++kg) b[kg] = *reinterpret_cast<const float4*>(hin + j * PITCH + kg * 16 + q * 4);
            if constexpr (LEAN) load_hidden(l, w);
            else if (l + 1 < a.n_mid) load_hidden(l + 1, wnext);
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kg = 0; kg < NTM; ++kg) {
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    acc[t] = mfma16(w[kg][t].x, b[kg].x, acc[t]);
                    acc[t] = mfma16(w[kg][t].y, b[kg].y, acc[t]);
                    acc[t] = mfma16(w[kg][t].z, b[kg].z, acc[t]);
                    acc[t] = mfma16(w[kg][t].w, b[kg].w, acc[t]);
                }
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int c = (nt0 + t) * 16 + q * 4;
                const float4 bv = *reinterpret_cast<const float4*>(a.b_mid + l * MID + c);
                const float4 gv = *reinterpret_cast<const float4*>(a.gn_gamma + l * MID + c);
                const float4 be = *reinterpret_cast<const float4*>(a.gn_beta + l * MID + c);
                const float v0 = acc[t][0] + bv.x, v1 = acc[t][1] + bv.y, v2 = acc[t][2] + bv.z, v3 = acc[t][3] + bv.w;
                float sum = (v0 + v1) + (v2 + v3);
                if (CPG == 16) { sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32); }
                const float mean = sum * (1.f / CPG);
                const float d0 = v0 - mean, d1 = v1 - mean, d2 = v2 - mean, d3 = v3 - mean;
                float ss = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                if (CPG == 16) { ss += __shfl_xor(ss, 16); ss += __shfl_xor(ss, 32); }
                const float rstd = rsqrtf(ss * (1.f / CPG) + a.gn_eps);
                float4 o = make_float4(lrelu(d0 * rstd * gv.x + be.x, slope), lrelu(d1 * rstd * gv.y + be.y, slope),
                                       lrelu(d2 * rstd * gv.z + be.z, slope), lrelu(d3 * rstd * gv.w + be.w, slope));
                *reinterpret_cast<float4*>(hout + j * PITCH + c) = o;
                if (a.save && rv && by == 0) {
                    float* sv = a.save + (size_t)(1 + 2 * l) * a.M * MID + (size_t)r * MID + c;
                    *reinterpret_cast<float4*>(sv) = make_float4(v0, v1, v2, v3);
                    *reinterpret_cast<float4*>(sv + (size_t)a.M * MID) = o;
                }
            }
        }
        cur ^= 1;
        __syncthreads();
    };
    if constexpr (LEAN) {
        for (int l = 0; l < a.n_mid; ++l) hidden(l, wA, wA);
    } else {
        for (int l = 0; l < a.n_mid; l += 2) {
            hidden(l, wA, wB);
            if (l + 1 < a.n_mid) hidden(l + 1, wB, wA);
        }
    }

    // ---- head: mid -> out; column tiles are dealt round-robin to (blockIdx.y, wave) ----
    const float* hin = hid + cur * 16 * PITCH;
    float4 b[NTM];
#pragma unroll
    for (int kg = 0; kg < NTM; ++kg) b[kg] = *reinterpret_cast<const float4*>(hin + j * PITCH + kg * 16 + q * 4);
    const int out_pad = (a.out_dim + 15) & ~15;
    const int NTO = out_pad / 16;
    const float4* wbase = reinterpret_cast<const float4*>(a.w_out) + lane;
    const int split = a.out_split > 0 ? a.out_split : a.out_dim;
    float* orow = a.out ? a.out + (size_t)rb * a.ob + (size_t)rj * a.orow : nullptr;
    const int tstart = by * 4 + wave, tstride = ny * 4;

    if (a.epi == GCPX_MLP_GAUSS) {
        const int nz = a.out_dim / 2;
        const int NTZ = nz / 16;
        const float* erow = a.eps + (size_t)rb * a.eb + (size_t)rj * a.erow;
        float* zrow = a.z + (size_t)rb * a.zb + (size_t)rj * a.zrow;
        for (int nt = tstart; nt < NTZ; nt += tstride) {
            float4 wm[NTM], wl[NTM];
#pragma unroll
            for (int kg = 0; kg < NTM; ++kg) {
                wm[kg] = wbase[(kg * NTO + nt) * 64];
                wl[kg] = wbase[(kg * NTO + nt + NTZ) * 64];
            }
            f32x4 am = f32x4{0, 0, 0, 0}, al = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kg = 0; kg < NTM; ++kg) {
                am = mfma16(wm[kg].x, b[kg].x, am); al = mfma16(wl[kg].x, b[kg].x, al);
                am = mfma16(wm[kg].y, b[kg].y, am); al = mfma16(wl[kg].y, b[kg].y, al);
                am = mfma16(wm[kg].z, b[kg].z, am); al = mfma16(wl[kg].z, b[kg].z, al);
                am = mfma16(wm[kg].w, b[kg].w, am); al = mfma16(wl[kg].w, b[kg].w, al);
            }
            if (rv) {
                const int n = nt * 16 + q * 4;
                const float4 bm = *reinterpret_cast<const float4*>(a.b_out + n);
                const float4 bl = *reinterpret_cast<const float4*>(a.b_out + nz + n);
                const float4 mu = make_float4(am[0] + bm.x, am[1] + bm.y, am[2] + bm.z, am[3] + bm.w);
                const float4 ls = make_float4(al[0] + bl.x, al[1] + bl.y, al[2] + bl.z, al[3] + bl.w);
                if (orow) {
                    *reinterpret_cast<float4*>(orow + n) = mu;
                    *reinterpret_cast<float4*>(orow + nz + n) = ls;
                }
                const float4 e = *reinterpret_cast<const float4*>(erow + n);
                float4 z;
                z.x = mu.x + expf(ls.x) * e.x; z.y = mu.y + expf(ls.y) * e.y;
                z.z = mu.z + expf(ls.z) * e.z; z.w = mu.w + expf(ls.w) * e.w;
                *reinterpret_cast<float4*>(zrow + n) = z;
            }
        }
    } else {
        for (int nt = tstart; nt < NTO; nt += tstride) {
            float4 w[NTM];
#pragma unroll
            for (int kg = 0; kg < NTM; ++kg) w[kg] = wbase[(kg * NTO + nt) * 64];
            f32x4 ao = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int kg = 0; kg < NTM; ++kg) {
                ao = mfma16(w[kg].x, b[kg].x, ao);
                ao = mfma16(w[kg].y, b[kg].y, ao);
                ao = mfma16(w[kg].z, b[kg].z, ao);
                ao = mfma16(w[kg].w, b[kg].w, ao);
            }
            if (rv) {
                const int n = nt * 16 + q * 4;
                if (n < a.out_dim) {
                    const float4 bv = *reinterpret_cast<const float4*>(a.b_out + n);
                    float v[4] = {ao[0] + bv.x, ao[1] + bv.y, ao[2] + bv.z, ao[3] + bv.w};
                    if (a.epi == GCPX_MLP_TANH) { v[0] = tanhf(v[0]); v[1] = tanhf(v[1]); v[2] = tanhf(v[2]); v[3] = tanhf(v[3]); }
                    const int blk = n / split, nn = n % split;
                    float* op = orow + (size_t)blk * a.oblk + nn;
                    if (n + 3 < a.out_dim && nn + 3 < split && (((uintptr_t)op) & 15) == 0) {
                        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int nk = n + k;
                            if (nk < a.out_dim) orow[(size_t)(nk / split) * a.oblk + nk % split] = v[k];
                        }
                    }
                }
            }
        }
    }
}

// Helper workgroups of a small launch.  A Predictor at 16 rows is ONE workgroup pulling its 0.65 MB of weights through one CU at the
// ~10 B / cycle a CU gets from HBM / MALL (29 us, NOTEBOOK.md section 6c) while 240 CUs idle.  The L2 is shared by the 32 CUs of an
// XCD, and an L2 hit is served several times faster than that: so the launch brings one extra workgroup for every idle CU, and those
// do nothing but LOAD the weights of the launch's problems — each XCD's helpers share the byte range between them (helper h is
// assumed to sit on XCD h % 8, the observed dispatch order; when that does not hold some lines are fetched twice and others by the
// compute workgroup itself: slower, never wrong) — in the order the compute workgroups consume them.  Nothing is stored.
__device__ __forceinline__ void prefetch_range(const float* base, const size_t n_floats, const int slot, const int nslots) {
    const size_t n4 = n_floats / 4;                                        // 16-byte units
    const float4* p = reinterpret_cast<const float4*>(base);
    for (size_t i = (size_t)slot * 256 + threadIdx.x; i < n4; i += (size_t)nslots * 256) {
        const float4 v = p[i];
        asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));            // keeps the load; its value is not needed
    }
}

__device__ __forceinline__ void prefetch_weights(const gcpx_mlp_args& a, const int helper, const int nhelpers) {
    const int nslots = nhelpers >= 8 ? nhelpers / 8 : 1;
    const int slot = nhelpers >= 8 ? helper / 8 : helper;
    if (slot >= nslots) return;
    const int out_pad = (a.out_dim + 15) & ~15;
    prefetch_range(a.w_in, (size_t)a.in_dim * a.mid, slot, nslots);
    if (a.n_mid) prefetch_range(a.w_mid, (size_t)a.n_mid * a.mid * a.mid, slot, nslots);
    prefetch_range(a.w_out, (size_t)a.mid * out_pad, slot, nslots);
}

// blocks (bx, by) with bx < gx compute; bx >= gx (launched only with gy == 1 ... see gcpx_mlp) are helpers
template <int MID, bool LEAN>
__global__ void __launch_bounds__(256, LEAN ? 2 : 1) mlp_kernel(const gcpx_mlp_args a, const int nblocks) {
    __shared__ float4 hid4[2 * 16 * (MID + 4) / 4];
    const int gx = (a.M + 15) / 16;
    if ((int)blockIdx.x >= nblocks) {
        prefetch_weights(a, blockIdx.x - nblocks, gridDim.x - nblocks);
        return;
    }
    mlp_rows<MID, LEAN>(a, blockIdx.x % gx, blockIdx.x / gx, nblocks / gx, reinterpret_cast<float*>(hid4));
}

// several Predictors in ONE launch (the prior next to the posterior of a tree level; the latent-space heads): independent
// problems, so what used to be parallel graph branches or a chain of ~20 us launches is one kernel boundary.
// dims[p] = {first block, row blocks gx, head splits gy}; blocks of problem p are (bx, by) = (local % gx, local / gx).
template <int MID, bool LEAN>
__global__ void __launch_bounds__(256, LEAN ? 2 : 1) mlp_group_kernel(const gcpx_mlp_args* __restrict__ tab, const int4* __restrict__ dims, const int n,
                                                                      const int nblocks) {
    __shared__ float4 hid4[2 * 16 * (MID + 4) / 4];
    if ((int)blockIdx.x >= nblocks) {                     // helper workgroups: pull every problem's weights into this XCD's L2
        for (int p = 0; p < n; ++p) prefetch_weights(tab[p], blockIdx.x - nblocks, gridDim.x - nblocks);
        return;
    }
    int p = 0;
    while (p + 1 < n && (int)blockIdx.x >= dims[p + 1].x) ++p;
    const int4 d = dims[p];
    const int local = blockIdx.x - d.x;
    mlp_rows<MID, LEAN>(tab[p], local % d.y, local / d.y, d.z, reinterpret_cast<float*>(hid4));
}

// A level of the tree in front of its recurrent cell: the prior and posterior Predictors of level l AND the merge of the parents'
// hidden states (tree_lstm.py:43-48) need nothing but level l - 1's results and not each other — one launch.  Workgroups
// [0, nblocks) are the Predictors' (as mlp_group_kernel), the next gx * gy * gz carry the GEMM's blocks as gemm_kernel<PR, CR, false, KS>
// would (the merge GEMM was a launch of its own on the dependent chain: 14 - 35 us per level at c2), the rest are helpers.
template <int MID, int PR, int CR, bool KS>
__global__ void __launch_bounds__(256, 1) level_pre_kernel(const gcpx_mlp_args* __restrict__ tab, const int4* __restrict__ dims, const int n,
                                                           const int nblocks, const gcpx_gemm_args g, const int gx, const int gy, const int gz) {
    __shared__ float4 hid4[2 * 16 * (MID + 4) / 4];
    const int blk = blockIdx.x;
    if (blk >= nblocks) {
        const int local = blk - nblocks, ng = gx * gy * gz;
        if (local < ng) {
            gemm_tile<PR, CR, false, KS>(g, local % gx, (local / gx) % gy, local / (gx * gy));
        } else {
            for (int p = 0; p < n; ++p) prefetch_weights(tab[p], local - ng, gridDim.x - nblocks - ng);
        }
        return;
    }
    int p = 0;
    while (p + 1 < n && blk >= dims[p + 1].x) ++p;
    const int4 d = dims[p];
    const int local = blk - d.x;
    mlp_rows<MID, false>(tab[p], local % d.y, local / d.y, d.z, reinterpret_cast<float*>(hid4));
}

typedef void (*level_pre_fn)(const gcpx_mlp_args*, const int4*, int, int, const gcpx_gemm_args, int, int, int);
level_pre_fn level_pre_variant(const int pr, const int cr, const int ks) {
    if (ks) {
        if (pr == 1 && cr == 1) return level_pre_kernel<128, 1, 1, true>;
        if (pr == 1 && cr == 2) return level_pre_kernel<128, 1, 2, true>;
        if (pr == 2 && cr == 2) return level_pre_kernel<128, 2, 2, true>;
        if (pr == 4 && cr == 2) return level_pre_kernel<128, 4, 2, true>;
        if (pr == 2 && cr == 4) return level_pre_kernel<128, 2, 4, true>;
        return nullptr;
    }
    if (pr == 2 && cr == 2) return level_pre_kernel<128, 2, 2, false>;
    if (pr == 4 && cr == 2) return level_pre_kernel<128, 4, 2, false>;
    return nullptr;
}

int mlp_check(const gcpx_mlp_args* a) {
    GCPX_CHECK_ARG(a != nullptr, "null args");
    GCPX_CHECK_ARG(a->nsrc >= 1 && a->nsrc <= 6, "nsrc out of range");
    GCPX_CHECK_ARG(a->M > 0 && a->rpb > 0, "bad M/rpb");
    int ksum = 0;
    for (int s = 0; s < a->nsrc; ++s) {
        GCPX_CHECK_ARG(a->src[s].ptr != nullptr, "source pointer is NULL");
        GCPX_CHECK_ARG(a->src[s].width > 0 && a->src[s].width % 16 == 0, "source width must be a multiple of 16");
        ksum += a->src[s].width;
    }
    GCPX_CHECK_ARG(ksum == a->in_dim, "in_dim != sum of source widths");
    GCPX_CHECK_ARG(a->w_in && a->b_in && a->w_out && a->b_out, "weights missing");
    GCPX_CHECK_ARG(a->n_mid == 0 || (a->w_mid && a->b_mid && a->gn_gamma && a->gn_beta), "hidden-layer weights missing");
    if (a->epi == GCPX_MLP_GAUSS) {
        GCPX_CHECK_ARG(a->out_dim % 32 == 0 && a->eps && a->z, "GAUSS epilogue needs out_dim % 32 == 0, eps, z");
    } else {
        GCPX_CHECK_ARG(a->out != nullptr, "out is NULL");
    }
    return GCPX_OK;
}

// split a wide head over blockIdx.y when there are few row blocks (the LSTM initialiser: 16 rows x 6144 cols)
void mlp_grid(const gcpx_mlp_args* a, int* gx, int* gy) {
    *gx = (a->M + 15) / 16;
    const int head_tiles = ((a->epi == GCPX_MLP_GAUSS ? a->out_dim / 2 : a->out_dim) + 15) / 16;
    static const int ymax = [] { const char* e = getenv("GCPX_MLP_GY_MAX"); return e ? atoi(e) : 32; }();     // tuning aid
    int y = 1;
    while (y < ymax && *gx * y < 256 && head_tiles / (4 * y) >= 2) y *= 2;
    *gy = y;
}

// helper workgroups (prefetch_weights) a launch of nb compute workgroups brings along: one per CU the launch leaves idle, while the
// launch is small enough to be bound by what ONE CU pulls (GCPX_MLP_HELPERS=0 switches them off, =n caps them)
int mlp_helpers(const int nb, const int cus) {
    static const int cap = [] { const char* e = getenv("GCPX_MLP_HELPERS"); return e ? atoi(e) : 1 << 30; }();
    if (nb > cus / 4) return 0;
    const int h = ((cus - nb) / 8) * 8;
    return h < cap ? (h > 0 ? h : 0) : cap;
}

}  // namespace

extern "C" int gcpx_mlp_group_dims(const gcpx_mlp_args* host_table, int32_t n, int32_t* dims, int32_t* total_blocks) {
    GCPX_CHECK_ARG(host_table && dims && total_blocks && n >= 1 && n <= 16, "bad arguments");
    int start = 0;
    for (int p = 0; p < n; ++p) {
        const int st = mlp_check(host_table + p);
        if (st != GCPX_OK) return st;
        GCPX_CHECK_ARG(host_table[p].mid == host_table[0].mid, "all problems of a group share the hidden width");
        int gx, gy;
        mlp_grid(host_table + p, &gx, &gy);
        dims[4 * p] = start; dims[4 * p + 1] = gx; dims[4 * p + 2] = gy; dims[4 * p + 3] = 0;
        start += gx * gy;
    }
    *total_blocks = start;
    return GCPX_OK;
}

extern "C" int gcpx_mlp_group(const gcpx_mlp_args* dev_table, const int32_t* dev_dims, int32_t n, int32_t total_blocks, int32_t mid,
                              void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(dev_table && dev_dims && n >= 1 && n <= 16 && total_blocks > 0, "bad arguments");
    GCPX_CHECK_ARG((((uintptr_t)dev_dims) & 15) == 0, "dims must be 16-byte aligned");
    const int cus = gcpx_conv_grid() / 2;
    const bool lean = total_blocks > cus;                            // more workgroups than CUs
    const int grid = total_blocks + mlp_helpers(total_blocks, cus);
    const int4* dd = reinterpret_cast<const int4*>(dev_dims);
    if (mid == 128 && lean) hipLaunchKernelGGL((mlp_group_kernel<128, true>), dim3(grid), dim3(256), 0, stream, dev_table, dd, n, total_blocks);
    else if (mid == 128) hipLaunchKernelGGL((mlp_group_kernel<128, false>), dim3(grid), dim3(256), 0, stream, dev_table, dd, n, total_blocks);
    else if (mid == 32 && lean) hipLaunchKernelGGL((mlp_group_kernel<32, true>), dim3(grid), dim3(256), 0, stream, dev_table, dd, n, total_blocks);
    else if (mid == 32) hipLaunchKernelGGL((mlp_group_kernel<32, false>), dim3(grid), dim3(256), 0, stream, dev_table, dd, n, total_blocks);
    else {
        gcpx_set_error("gcpx_mlp_group: unsupported mid=%d (128 or 32)", mid);
        return GCPX_ERR_UNSUPPORTED;
    }
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

// 1 when gcpx_mlp_group_gemm can carry this GEMM next to a group of total_blocks Predictor workgroups of hidden width mid
extern "C" int gcpx_mlp_group_gemm_supported(const gcpx_gemm_args* g, int32_t total_blocks, int32_t mid) {
    if (!g || mid != 128 || g->epi == GCPX_EPI_LSTM || g->epi == GCPX_EPI_GAUSS_SAMPLE) return 0;
    const int cus = gcpx_conv_grid() / 2;
    if (total_blocks > cus) return 0;                                // (the throughput-bound Predictor launches run the lean kernel)
    int pr, cr, ks, gx, gy, gz;
    if (!gcpx_gemm_tile_plan(g, &pr, &cr, &ks, &gx, &gy, &gz)) return 0;
    return level_pre_variant(pr, cr, ks) != nullptr;
}

extern "C" int gcpx_mlp_group_gemm(const gcpx_mlp_args* dev_table, const int32_t* dev_dims, int32_t n, int32_t total_blocks, int32_t mid,
                                   const gcpx_gemm_args* g, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(dev_table && dev_dims && g && n >= 1 && n <= 16 && total_blocks > 0, "bad arguments");
    GCPX_CHECK_ARG((((uintptr_t)dev_dims) & 15) == 0, "dims must be 16-byte aligned");
    int pr, cr, ks, gx, gy, gz;
    level_pre_fn kern = nullptr;
    if (mid == 128 && g->epi != GCPX_EPI_LSTM && g->epi != GCPX_EPI_GAUSS_SAMPLE && gcpx_gemm_tile_plan(g, &pr, &cr, &ks, &gx, &gy, &gz))
        kern = level_pre_variant(pr, cr, ks);
    if (!kern) {
        gcpx_set_error("gcpx_mlp_group_gemm: this GEMM / hidden width has no combined launch (ask gcpx_mlp_group_gemm_supported)");
        return GCPX_ERR_UNSUPPORTED;
    }
    const int cus = gcpx_conv_grid() / 2;
    const int work = total_blocks + gx * gy * gz;
    const int grid = work + mlp_helpers(work, cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, stream, dev_table, reinterpret_cast<const int4*>(dev_dims), n, total_blocks, *g, gx, gy, gz);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

extern "C" int gcpx_mlp(const gcpx_mlp_args* a, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int st = mlp_check(a);
    if (st != GCPX_OK) return st;
    int gx, gy;
    mlp_grid(a, &gx, &gy);
    const int cus = gcpx_conv_grid() / 2;
    const bool lean = gx * gy > cus;                                 // more workgroups than CUs
    const int nb = gx * gy, grid = nb + mlp_helpers(nb, cus);
    if (a->mid == 128 && lean) hipLaunchKernelGGL((mlp_kernel<128, true>), dim3(grid), dim3(256), 0, stream, *a, nb);
    else if (a->mid == 128) hipLaunchKernelGGL((mlp_kernel<128, false>), dim3(grid), dim3(256), 0, stream, *a, nb);
    else if (a->mid == 32 && lean) hipLaunchKernelGGL((mlp_kernel<32, true>), dim3(grid), dim3(256), 0, stream, *a, nb);
    else if (a->mid == 32) hipLaunchKernelGGL((mlp_kernel<32, false>), dim3(grid), dim3(256), 0, stream, *a, nb);
    else {
        gcpx_set_error("gcpx_mlp: unsupported mid=%d (128 or 32)", a->mid);
        return GCPX_ERR_UNSUPPORTED;
    }
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
