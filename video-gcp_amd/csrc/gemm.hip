// Row GEMM with gathered / concatenated / shifted inputs and fused epilogues, on f32 MFMA (gfx950).
//
//   out[r, n] = epi( sum_s sum_k X_s[map_s(r), k] * W[n, koff_s + k] + bias[n] )
//
// Replaces the Linear / LSTMCell / Conv1d / 1x1->4x4 ConvTranspose / 4x4-valid Conv launches of
//   /root/reference/gcp/prediction/models/tree/tree_lstm.py:43-49   (split_linear merge, HiddenStatePredictorModel)
//   /root/reference/gcp/prediction/models/base_gcp.py:199           (ConvSeqEncodingModule, conv over time)
//   encoder head / decoder input block (blox, absent; spec in DESIGN.md).
//
// Layout: output columns n sit on the MFMA i side (A operand = weights, pre-packed in fragment order so a
// wavefront's load of 16 columns x 16 k is one coalesced 1 KiB read), rows sit on the j side (B operand: lane
// (j, kk) reads 16 B = 4 consecutive k of its row, with the producer's BatchNorm affine + LeakyReLU applied on
// load).  A lane ends with 4 consecutive columns of one row: with gate-interleaved LSTM weights (n = 4u + gate)
// the whole cell update for (row, unit u) is lane-local.  No LDS, no barriers: wavefronts are independent.
#include "common.h"
#include "gemm_tile.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

bool gcpx_gemm_split_applies(const gcpx_gemm_args* a);                  // gemm_split.hip
int gcpx_launch_gemm_split(const gcpx_gemm_args* a, hipStream_t stream);
bool gcpx_gemm_planes_applies(const gcpx_gemm_args* a);                 // gemm_planes.hip
int gcpx_launch_gemm_planes(const gcpx_gemm_args* a, hipStream_t stream);

namespace {

template <int PR, int CR, bool LSTM, bool KS = false>
__global__ void __launch_bounds__(256) gemm_kernel(const gcpx_gemm_args a) {
    gemm_tile<PR, CR, LSTM, KS>(a, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several independent small-M problems (the split-K <1,1> regime) in ONE launch: the same layer of the three recurrent nets of a
// VRNN step, or two GEMMs of a tree level that do not depend on each other.  A dependent chain of ~5 us launches is bound by the
// number of launches, not by their work.  dims[p] = {first block, row blocks, column tiles, batches}.
__global__ void __launch_bounds__(256) gemm_group_kernel(const gcpx_gemm_args* __restrict__ tab, const int4* __restrict__ dims, const int n) {
    int p = 0;
    while (p + 1 < n && (int)blockIdx.x >= dims[p + 1].x) ++p;
    const int4 d = dims[p];
    const int local = blockIdx.x - d.x;
    const int bx = local % d.y, by = (local / d.y) % d.z, bz = local / (d.y * d.z);
    const gcpx_gemm_args& a = tab[p];
    if (a.epi == GCPX_EPI_GAUSS_SAMPLE) {                      // 16 rows x 16 latent dimensions per block (formula of gauss_sample_kernel)
        const int r = bx * 16 + ((int)threadIdx.x >> 4), d = by * 16 + ((int)threadIdx.x & 15);
        if (r < a.M) {
            const int b = r / a.rpb, j = r % a.rpb;
            const float* m = a.src[0].ptr + (size_t)b * a.src[0].sb + (size_t)j * a.src[0].sr;
            const float e = a.src[1].ptr[(size_t)b * a.src[1].sb + (size_t)j * a.src[1].sr + d];
            a.out[(size_t)b * a.ob + (size_t)j * a.orow + d] = m[d] + expf(m[a.N + d]) * e;
        }
        return;
    }
    if (a.epi == GCPX_EPI_LSTM) gemm_tile<1, 1, true, true>(a, bx, by, bz);
    else gemm_tile<1, 1, false, true>(a, bx, by, bz);
}

struct TileChoice { int pr, cr; };

TileChoice choose_tile(int M, int N, int nb = 1) {
    if (const char* ov = getenv("GCPX_GEMM_TILE")) {           // tuning aid: "pr,cr,minM" (also seen by gcpx_gemm_row_blocks)
        int pr = 0, cr = 0, mm = 0;
        if (sscanf(ov, "%d,%d,%d", &pr, &cr, &mm) == 3 && M >= mm && N % (16 * cr) == 0) return TileChoice{pr, cr};
    }
    // Every wavefront owns a PR x CR block of 16 x 16 tiles and walks all of K alone.  Cost model fitted to measurements on MI355X
    // (tools/run_gemm_tiles.sh, K = 1024; microseconds per launch, up to a common constant):
    //   compute = rounds x block area x e(area),  rounds = ceil(wavefronts / SIMDs): one wavefront per SIMD is the sweet spot
    //             (M = 512, N = 2048: <2,2> = 1024 wavefronts 26 us; <4,2> = half the SIMDs idle 45 us); a second round costs more
    //             than twice the first (co-resident wavefronts share the MFMA pipe AND thrash L1): x1.35;
    //             e = 4.2 / 4.9 / 5.0 / 6.0 / 8.0 us per tile for areas 16 / 8 / 4 / 2 / 1 (bigger blocks: fewer loads per MFMA);
    //   L2      = tiles x 256 MFMAs x bytes per MFMA / ~10 TB/s, bytes per MFMA = 256 (PR + CR) / (PR CR): <4,1> blocks of the batched
    //             merge at M = 1024 ran 114 us against 74 us for <4,4>, L2-bound.
    // The launch is priced at max(compute, L2); ties go to the block with fewer bytes per MFMA.
    static const long simds = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return 4L * (cus > 0 ? cus : 256);
    }();
    const int prs[3] = {4, 2, 1}, crs[3] = {4, 2, 1};
    TileChoice best{1, 1};
    double best_cost = -1.0, best_bytes = 0.0;
    const long NT = N / 16, RT = (M + 15) / 16;
    // a problem whose largest blocks already give every SIMD a wavefront is throughput-bound: largest blocks, fewest operand bytes
    // (the batched merge data gradient, 24576 tiles: <2,2> in six rounds 250 us — the fitted model below underrates how badly small
    // blocks do once several rounds queue up behind L2)
    if (M >= 64 && N % 64 == 0 && ((RT + 3) / 4) * (NT / 4) * nb >= simds) return TileChoice{4, 4};
    for (int pi = 0; pi < 3; ++pi)
        for (int ci = 0; ci < 3; ++ci) {
            const int pr = prs[pi], cr = crs[ci];
            if (N % (16 * cr)) continue;
            if (pr > 1 && 16 * pr > ((M + 15) & ~15)) continue;      // row tiles that would be entirely masked
            const long waves = ((RT + pr - 1) / pr) * ((NT + cr - 1) / cr) * nb;
            const long rounds = (waves + simds - 1) / simds;
            const int area = pr * cr;
            const double e = area >= 16 ? 4.2 : area >= 8 ? 4.9 : area >= 4 ? 5.0 : area >= 2 ? 6.0 : 8.0;
            const double compute = (double)rounds * area * e * (rounds > 1 ? 1.35 : 1.0);
            const double bytes = 256.0 * (pr + cr) / area;
            const double l2 = (double)RT * NT * nb * 256.0 * bytes / 1.0e7;
            const double cost = compute > l2 ? compute : l2;
            if (best_cost < 0 || cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && bytes < best_bytes)) {
                best_cost = cost; best_bytes = bytes; best = TileChoice{pr, cr};
            }
        }
    return best;
}

template <int PR, int CR>
void launch_t(const gcpx_gemm_args* a, hipStream_t stream) {
    const int rbk = (a->M + 16 * PR - 1) / (16 * PR);
    const int cbk = (a->N / 16 + 4 * CR - 1) / (4 * CR);
    const int nb = a->nbatch > 1 ? a->nbatch : 1;
    if (a->epi == GCPX_EPI_LSTM)
        hipLaunchKernelGGL((gemm_kernel<PR, CR, true>), dim3(rbk, cbk, nb), dim3(256), 0, stream, *a);
    else
        hipLaunchKernelGGL((gemm_kernel<PR, CR, false>), dim3(rbk, cbk, nb), dim3(256), 0, stream, *a);
}

}  // namespace

extern "C" int gcpx_gemm_row_blocks(int32_t M, int32_t N) {
    const TileChoice t = choose_tile(M, N);
    return (M + 16 * t.pr - 1) / (16 * t.pr);
}

static int gemm_check(const gcpx_gemm_args* a);

// From 64 rows: a split-K workgroup (four wavefronts share a PR x CR block of tiles and split K) sized by the bytes a CU pulls.
// These launches are bound by operand delivery (~10 B / cycle and CU): a workgroup pulls (PR + CR) x 16 x K x 4 bytes, so the
// cost of a block shape is rounds x (PR + CR), rounds = workgroups / CUs rounded up.  Measured (K = 1024, us): cost 3: 9.3, 4:
// 10.5 - 12.4, 6: 16.3 - 18.8, 8: 21.5, 12: 29 - 36 — against 20.8 / 24.6 for the 128 / 256-row LSTM GEMM on one-wavefront blocks
// (cost model of choose_tile).  A multi-tile block is taken when the best shape costs <= 6 — the mid levels of the tree, and the
// narrow GEMMs at any row count (encoder head 1280 x 128 x 2048: 19.8 -> 14.4 us; embedding 1024 x 512 x 768: 18.8 -> 13.0); the
// single-tile split-K path keeps cost <= 2.
static bool ks_block_choice(const gcpx_gemm_args* a, int* pr_out, int* cr_out) {
    static const int max_rows = [] { const char* e = getenv("GCPX_GEMM_KS_MAX_ROWS"); return e ? atoi(e) : 4096; }();
    static const int max_cost = [] { const char* e = getenv("GCPX_GEMM_KS_MAX_COST"); return e ? atoi(e) : 6; }();
    if (!(a->M >= 64 && a->M <= max_rows && a->K >= 256 && !a->stats_partial) || getenv("GCPX_GEMM_NO_KS_BLOCKS")) return false;
    static const long cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return (long)(n > 0 ? n : 256);
    }();
    const long nb = a->nbatch > 1 ? a->nbatch : 1;
    // (a 4 x 4 block is MFMA-bound, not delivery-bound: 512 x 2048 x 1024 at cost 8 ran 31 us against 26 on one-wavefront blocks)
    const int shapes[5][2] = {{1, 1}, {1, 2}, {2, 2}, {4, 2}, {2, 4}};
    int best = 0;
    long best_cost = -1;
    for (int i = 0; i < 5; ++i) {
        const int pr = shapes[i][0], cr = shapes[i][1];
        if (a->N % (16 * cr)) continue;
        const long wgs = (long)((a->M + 16 * pr - 1) / (16 * pr)) * (a->N / (16 * cr)) * nb;
        const long cost = ((wgs + cus - 1) / cus) * (pr + cr);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = i; }
    }
    if (best == 0 || best_cost > max_cost) return false;
    *pr_out = shapes[best][0];
    *cr_out = shapes[best][1];
    return true;
}

// true when gcpx_gemm would run the problem as split-K <1,1> tiles (what gcpx_gemm_group launches)
static bool gemm_is_small(const gcpx_gemm_args* a) {
    int pr_ = 1, cr_ = 1;
    if (ks_block_choice(a, &pr_, &cr_)) return false;          // a multi-tile split-K block beats sharing a launch
    const int nb = a->nbatch > 1 ? a->nbatch : 1;
    const TileChoice t = choose_tile(a->M, a->N, nb);
    const long rbk = (a->M + 15) / 16, nt = a->N / 16;
    return t.pr == 1 && t.cr == 1 && a->K >= 256 && rbk * nt * nb <= 1024 && !a->stats_partial;
}

extern "C" int gcpx_gemm_group_dims(const gcpx_gemm_args* host_table, int32_t n, int32_t* dims, int32_t* total_blocks) {
    GCPX_CHECK_ARG(host_table && dims && total_blocks && n >= 1 && n <= 16, "bad arguments");
    int start = 0;
    for (int p = 0; p < n; ++p) {
        if (host_table[p].epi == GCPX_EPI_GAUSS_SAMPLE) {      // the reparametrised draw: 16 x 16 elements per block
            const gcpx_gemm_args& a = host_table[p];
            GCPX_CHECK_ARG(a.nsrc == 2 && a.src[0].ptr && a.src[1].ptr && a.out && a.M > 0 && a.N > 0 && a.N % 16 == 0 && a.rpb > 0 &&
                               a.src[0].width == 2 * a.N && a.src[1].width == a.N, "gauss sample: [mu | log_sigma] rows, eps rows, out");
            dims[4 * p] = start; dims[4 * p + 1] = (a.M + 15) / 16; dims[4 * p + 2] = a.N / 16; dims[4 * p + 3] = 1;
            start += dims[4 * p + 1] * dims[4 * p + 2];
            continue;
        }
        const int st = gemm_check(host_table + p);
        if (st != GCPX_OK) return st;
        if (!gemm_is_small(host_table + p)) {
            gcpx_set_error("gcpx_gemm_group_dims: problem %d (M=%d N=%d K=%d) is not in the small-M split-K regime", p, host_table[p].M,
                           host_table[p].N, host_table[p].K);
            return GCPX_ERR_UNSUPPORTED;
        }
        const int nb = host_table[p].nbatch > 1 ? host_table[p].nbatch : 1;
        const int rbk = (host_table[p].M + 15) / 16, nt = host_table[p].N / 16;
        dims[4 * p] = start; dims[4 * p + 1] = rbk; dims[4 * p + 2] = nt; dims[4 * p + 3] = nb;
        start += rbk * nt * nb;
    }
    *total_blocks = start;
    return GCPX_OK;
}

extern "C" int gcpx_gemm_group(const gcpx_gemm_args* dev_table, const int32_t* dev_dims, int32_t n, int32_t total_blocks, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(dev_table && dev_dims && n >= 1 && n <= 16 && total_blocks > 0, "bad arguments");
    GCPX_CHECK_ARG((((uintptr_t)dev_dims) & 15) == 0, "dims must be 16-byte aligned");
    hipLaunchKernelGGL(gemm_group_kernel, dim3(total_blocks), dim3(256), 0, stream, dev_table, reinterpret_cast<const int4*>(dev_dims), n);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}

static int gemm_check(const gcpx_gemm_args* a) {
    GCPX_CHECK_ARG(a != nullptr, "null args");
    GCPX_CHECK_ARG(a->epi != GCPX_EPI_GAUSS_SAMPLE, "GCPX_EPI_GAUSS_SAMPLE is a gcpx_gemm_group problem");
    GCPX_CHECK_ARG(a->nsrc >= 1 && a->nsrc <= 6, "nsrc out of range");
    GCPX_CHECK_ARG(a->M > 0 && a->N > 0 && a->N % 16 == 0 && a->rpb > 0, "bad M/N/rpb");
    int ksum = 0;
    for (int s = 0; s < a->nsrc; ++s) {
        GCPX_CHECK_ARG(a->src[s].ptr != nullptr, "source pointer is NULL");
        GCPX_CHECK_ARG(a->src[s].width > 0 && a->src[s].width % 16 == 0, "source width must be a multiple of 16");
        GCPX_CHECK_ARG(!(a->src[s].scale || a->src[s].act) ||
                           (a->src[s].cmod > 0 && (a->src[s].cmod & (a->src[s].cmod - 1)) == 0),
                       "cmod must be a power of two when scale/act is set");
        ksum += a->src[s].width;
    }
    GCPX_CHECK_ARG(ksum == a->K, "K != sum of source widths");
    GCPX_CHECK_ARG(a->wpk != nullptr, "weights missing");
    if (a->epi == GCPX_EPI_LSTM) {
        GCPX_CHECK_ARG(a->c_prev && a->h_out && a->c_out, "LSTM epilogue needs c_prev/h_out/c_out");
    } else {
        GCPX_CHECK_ARG(a->out != nullptr, "out is NULL");
    }
    GCPX_CHECK_ARG(a->nbatch <= 1 || (a->epi != GCPX_EPI_LSTM && !a->stats_partial), "nbatch > 1 only for plain epilogues");
    GCPX_CHECK_ARG(!a->lstm_bwd || (a->epi == GCPX_EPI_NONE && a->nbatch <= 1 && !a->stats_partial),
                   "lstm_bwd: plain epilogue, single problem");
    return GCPX_OK;
}

// How gcpx_gemm would tile this problem, for launches that carry its workgroups next to other work (mlp.hip: level_pre_kernel):
// block shape, split-K or one-wavefront blocks, grid.  false: bad arguments, or the problem runs on the split-f16 kernel.
bool gcpx_gemm_tile_plan(const gcpx_gemm_args* a, int* pr, int* cr, int* ks, int* gx, int* gy, int* gz) {
    if (gemm_check(a) != GCPX_OK || a->stats_partial) return false;
    const int nb = a->nbatch > 1 ? a->nbatch : 1;
    *gz = nb;
    if (ks_block_choice(a, pr, cr)) {
        *ks = 1; *gx = (a->M + 16 * *pr - 1) / (16 * *pr); *gy = a->N / (16 * *cr);
        return true;
    }
    if (gcpx_gemm_split_applies(a)) return false;
    const TileChoice t = choose_tile(a->M, a->N, nb);
    const long rbk = (a->M + 15) / 16, nt = a->N / 16;
    if (t.pr == 1 && t.cr == 1 && a->K >= 256 && !getenv("GCPX_GEMM_NOKS") && rbk * nt * nb <= 1024) {
        *pr = 1; *cr = 1; *ks = 1; *gx = (int)rbk; *gy = (int)nt;
        return true;
    }
    *pr = t.pr; *cr = t.cr; *ks = 0;
    *gx = (a->M + 16 * t.pr - 1) / (16 * t.pr); *gy = (a->N / 16 + 4 * t.cr - 1) / (4 * t.cr);
    return true;
}

extern "C" int gcpx_gemm(const gcpx_gemm_args* a, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int st = gemm_check(a);
    if (st != GCPX_OK) return st;
    {
        int pr_ = 1, cr_ = 1;
        if (!ks_block_choice(a, &pr_, &cr_)) {
            if (gcpx_gemm_planes_applies(a)) return gcpx_launch_gemm_planes(a, stream);
            if (gcpx_gemm_split_applies(a)) return gcpx_launch_gemm_split(a, stream);
        }
    }
    const TileChoice t = choose_tile(a->M, a->N, a->nbatch > 1 ? a->nbatch : 1);
    {
        int pr = 1, cr = 1;
        if (ks_block_choice(a, &pr, &cr)) {
            const long nb = a->nbatch > 1 ? a->nbatch : 1;
            const dim3 grid((a->M + 16 * pr - 1) / (16 * pr), a->N / (16 * cr), nb);
            const bool lstm = a->epi == GCPX_EPI_LSTM;
#define GCPX_KS_LAUNCH(PR_, CR_)                                                                                              \
            do {                                                                                                              \
                if (lstm) hipLaunchKernelGGL((gemm_kernel<PR_, CR_, true, true>), grid, dim3(256), 0, stream, *a);            \
                else hipLaunchKernelGGL((gemm_kernel<PR_, CR_, false, true>), grid, dim3(256), 0, stream, *a);                \
            } while (0)
            if (pr == 1 && cr == 2) GCPX_KS_LAUNCH(1, 2);
            else if (pr == 2 && cr == 2) GCPX_KS_LAUNCH(2, 2);
            else if (pr == 4 && cr == 2) GCPX_KS_LAUNCH(4, 2);
            else GCPX_KS_LAUNCH(2, 4);
#undef GCPX_KS_LAUNCH
            GCPX_CHECK_LAUNCH();
            return GCPX_OK;
        }
    }
    if (t.pr == 1 && t.cr == 1 && a->K >= 256 && !getenv("GCPX_GEMM_NOKS")) {
        // few rows: split K over the wavefronts of a workgroup when that still leaves the launch small
        const long nb = a->nbatch > 1 ? a->nbatch : 1;
        const long rbk = (a->M + 15) / 16, nt = a->N / 16;
        if (rbk * nt * nb <= 1024) {
            if (a->epi == GCPX_EPI_LSTM)
                hipLaunchKernelGGL((gemm_kernel<1, 1, true, true>), dim3(rbk, nt, nb), dim3(256), 0, stream, *a);
            else
                hipLaunchKernelGGL((gemm_kernel<1, 1, false, true>), dim3(rbk, nt, nb), dim3(256), 0, stream, *a);
            GCPX_CHECK_LAUNCH();
            return GCPX_OK;
        }
    }
    switch (t.pr * 10 + t.cr) {
        case 44: launch_t<4, 4>(a, stream); break;
        case 42: launch_t<4, 2>(a, stream); break;
        case 41: launch_t<4, 1>(a, stream); break;
        case 24: launch_t<2, 4>(a, stream); break;
        case 22: launch_t<2, 2>(a, stream); break;
        case 21: launch_t<2, 1>(a, stream); break;
        case 14: launch_t<1, 4>(a, stream); break;
        case 12: launch_t<1, 2>(a, stream); break;
        default: launch_t<1, 1>(a, stream); break;
    }
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
