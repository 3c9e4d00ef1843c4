// Image metrics of the evaluation harness (gcp/evaluation/compute_metrics.py:123-130: mse / psnr / ssim of a generated sequence
// against the ground truth, from blox.torch.evaluation — absent; spec in video-gcp_amd/evaluation.py and oracle/metrics_oracle.py).
//   per frame f:  sq[f]   = sum over (c, y, x) of (a - b)^2            (images in [-1, 1])
//                 ssim[f] = mean over channels of the mean SSIM map, 7x7 uniform window, K1 = 0.01, K2 = 0.03, data range 1 on
//                           the [0, 1]-scaled images, sample covariance (n / (n - 1)) — skimage.metrics.structural_similarity
//                           defaults, valid windows only
// HBM-bound byte work: one workgroup per (frame, channel), both planes staged in LDS once.
#include "common.h"

namespace {

constexpr int WIN = 7;

__global__ void __launch_bounds__(256) image_metrics_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                            const int* __restrict__ frame_map, const int C, const int H, const int W,
                                                            double* __restrict__ sq_part, double* __restrict__ ssim_part) {
    extern __shared__ float lds[];
    float* A = lds;
    float* Bm = lds + H * W;
    __shared__ double red[2][256];
    const int f = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    const int fe = frame_map ? frame_map[f] : f;           // frame of `est` compared with frame f of `tgt` (negative: skip)
    if (fe < 0) {
        if (tid == 0) { sq_part[(size_t)f * C + c] = 0.0; ssim_part[(size_t)f * C + c] = 0.0; }
        return;
    }
    const size_t plane = (size_t)H * W;
    const float* ap = est + ((size_t)fe * C + c) * plane;
    const float* bp = tgt + ((size_t)f * C + c) * plane;
    double sq = 0.0;
    for (int i = tid; i < H * W; i += 256) {
        const float a = ap[i], b = bp[i];
        const float d = a - b;
        sq += (double)d * d;
        A[i] = 0.5f * (a + 1.f);
        Bm[i] = 0.5f * (b + 1.f);
    }
    __syncthreads();
    const int oh = H - WIN + 1, ow = W - WIN + 1;
    const double np = WIN * WIN, cov_norm = np / (np - 1.0);
    const double C1 = 0.01 * 0.01, C2 = 0.03 * 0.03;
    double ss = 0.0;
    for (int i = tid; i < oh * ow; i += 256) {
        const int y = i / ow, x = i % ow;
        float sa = 0.f, sb = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
        for (int dy = 0; dy < WIN; ++dy) {
            const float* ra = A + (y + dy) * W + x;
            const float* rb = Bm + (y + dy) * W + x;
#pragma unroll
            for (int dx = 0; dx < WIN; ++dx) {
                const float a = ra[dx], b = rb[dx];
                sa += a; sb += b; saa = fmaf(a, a, saa); sbb = fmaf(b, b, sbb); sab = fmaf(a, b, sab);
            }
        }
        const double ux = sa / np, uy = sb / np;
        const double vx = cov_norm * (saa / np - ux * ux), vy = cov_norm * (sbb / np - uy * uy), vxy = cov_norm * (sab / np - ux * uy);
        ss += ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
    red[0][tid] = sq;
    red[1][tid] = ss;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        sq_part[(size_t)f * C + c] = red[0][0];
        ssim_part[(size_t)f * C + c] = red[1][0] / (double)(oh * ow);
    }
}

// per sequence b: mean over its frames [first, last) of the per-frame metrics -> mse, psnr, ssim
__global__ void sequence_metrics_kernel(const double* __restrict__ sq_part, const double* __restrict__ ssim_part,
                                        const int* __restrict__ first, const int* __restrict__ last, const int T, const int C,
                                        const long long elems_per_frame, float* __restrict__ out) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    double sq = 0.0, ps = 0.0, ss = 0.0;
    int n = 0;
    for (int t = first[b]; t < last[b]; ++t) {
        double fsq = 0.0, fss = 0.0;
        for (int c = 0; c < C; ++c) {
            fsq += sq_part[((size_t)b * T + t) * C + c];
            fss += ssim_part[((size_t)b * T + t) * C + c];
        }
        sq += fsq;
        const double mse01 = fsq / (double)elems_per_frame * 0.25;      // on the [0, 1]-scaled images
        ps += 10.0 * log10(1.0 / fmax(mse01, 1e-30));
        ss += fss / C;
        ++n;
    }
    const double nn = n > 0 ? (double)n : 1.0;
    out[b * 3 + 0] = n > 0 ? (float)(sq / (nn * (double)elems_per_frame)) : NAN;
    out[b * 3 + 1] = n > 0 ? (float)(ps / nn) : NAN;
    out[b * 3 + 2] = n > 0 ? (float)(ss / nn) : NAN;
}

}  // namespace

extern "C" int gcpx_image_metrics(const float* est, const float* tgt, const int32_t* frame_map, const int32_t* first, const int32_t* last,
                                  int32_t B, int32_t T, int32_t C, int32_t H, int32_t W, double* scratch, float* out, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    GCPX_CHECK_ARG(est && tgt && first && last && scratch && out, "null pointer");
    GCPX_CHECK_ARG(B > 0 && T > 0 && C > 0 && H >= WIN && W >= WIN, "bad sizes (frames must be at least 7 x 7)");
    const size_t lds = 2 * (size_t)H * W * sizeof(float);
    GCPX_CHECK_ARG(lds <= 64 * 1024, "frame too large for the LDS staging (H * W <= 8192)");
    double* sq = scratch;
    double* ss = scratch + (size_t)B * T * C;
    hipLaunchKernelGGL(image_metrics_kernel, dim3(B * T, C), dim3(256), lds, stream, est, tgt, frame_map, C, H, W, sq, ss);
    hipLaunchKernelGGL(sequence_metrics_kernel, dim3(B), dim3(64), 0, stream, sq, ss, first, last, T, C, (long long)C * H * W, out);
    GCPX_CHECK_LAUNCH();
    return GCPX_OK;
}
