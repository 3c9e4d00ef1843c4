/*
 * gcpx.h — C-ABI of libgcpx.so: the MI355X (gfx950) hot path of the goal-conditioned hierarchical video
 * predictor (gcp_tree).  Plain pointers and sizes only; every pointer marked "dev" is a device (HBM) pointer
 * owned by the caller.  The library allocates nothing on the hot path and never synchronises; every launch
 * is enqueued on the caller's stream.  Every function returns 0 on success, a negative gcpx_status otherwise;
 * gcpx_last_error() gives the text.
 *
 * The reference (orybkin/video-gcp, mounted at /root/reference) has no FFI/plugin layer — its boundary is
 * Python duck typing.  Each entry point therefore cites the reference call site whose device
 * work it replaces.
 */
#ifndef GCPX_H
#define GCPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCPX_VERSION 1

typedef enum gcpx_status {
    GCPX_OK = 0,
    GCPX_ERR_INVALID_ARG = -1,
    GCPX_ERR_UNSUPPORTED = -2,
    GCPX_ERR_HIP = -3,
    GCPX_ERR_COMM = -4
} gcpx_status;

typedef enum gcpx_act { GCPX_ACT_NONE = 0, GCPX_ACT_LRELU = 1, GCPX_ACT_TANH = 2 } gcpx_act;

int gcpx_version(void);
const char* gcpx_last_error(void);
/* number of workgroups a persistent conv launch will use (size of the stats_partial first dim) */
int gcpx_conv_grid(void);

/* ---------------------------------------------------------------------------------------------------
 * Convolution stacks.
 *   replaces blox.torch.encoder_decoder.Encoder / DecoderModule as called at
 *   gcp/prediction/models/base_gcp.py:188,208-209 (encoder) and
 *   gcp/prediction/models/tree/tree_dense_rec.py:42 (decoder.decode_seq over all tree nodes).
 * Activations are fp32 NHWC [F][H][W][C].  A conv's input is the channel concatenation of up to two sources
 * (decoder features ++ skip activations of I_0, broadcast over the node axis with frame_div); each source is
 * passed RAW (pre-normalisation) with the per-channel affine (BatchNorm folded to scale/shift) and
 * LeakyReLU applied while the tile is staged ("normalise on load").
 * ------------------------------------------------------------------------------------------------- */
typedef struct gcpx_conv_src {
    const float* ptr;    /* dev: NHWC [F / frame_div][Hin][Win][C] */
    const float* scale;  /* dev: [C] or NULL (identity) */
    const float* shift;  /* dev: [C] or NULL */
    int32_t C;           /* channels of this source (multiple of 16) */
    int32_t frame_div;   /* source frame = f / frame_div (1 = per-frame; N = broadcast over N nodes) */
    int32_t act;         /* gcpx_act applied after the affine */
    int32_t _pad;
} gcpx_conv_src;

typedef enum gcpx_head_mode {
    GCPX_HEAD_RAW = 0,      /* store conv output NHWC [F][H][W][CT*16] (kernel channel order), + bias, + out_act */
    GCPX_HEAD_DLM_MEAN = 1, /* discrete-logistic-mixture mean -> images NCHW [F][3][H][W]; nothing else written */
    GCPX_HEAD_DLM_BOTH = 2, /* both of the above */
    GCPX_HEAD_TANH_NCHW = 3, /* gaussian head: tanh(first 3 channels) -> images NCHW */
    GCPX_HEAD_DLM_NLL = 4,  /* mixture mean -> images, and for every frame f with raw_row_map[f] >= 0 the discretised-logistic-mixture
                               negative log-likelihood of target row raw_row_map[f] (decoder.nll on the matched frames,
                               frame_binding.py:88-99), evaluated in the epilogue: nll_partial[i][row] = sum over the pixels of item i
                               (4 rows x 16 columns; (H / 4) * (W / 16) items per frame) — reduce over i with gcpx_reduce_partials.
                               No raw parameters are stored.  Split-f16 head only (wpk_split set) */
    GCPX_HEAD_DLM_NLL_GRAD = 5 /* training forward: as GCPX_HEAD_DLM_NLL, and row raw_row_map[f] of `out` ([rows][H][W][112]) receives
                               d (nll_scale * nll_row_weight[row] * NLL) / d parameters — what gcpx_dlm_nll_bwd computes from stored
                               parameters.  Rows no frame maps to are not written (zero them: gcpx_zero_unmapped_rows) */
} gcpx_head_mode;

typedef struct gcpx_conv_args {
    gcpx_conv_src src[2];
    int32_t nsrc;
    int32_t F;              /* output frames */
    int32_t Hin, Win;       /* source spatial size */
    int32_t Hout, Wout;     /* output spatial size (2x source when upsample, 1/2 for the stride-2 encoder conv) */
    int32_t Cin;            /* sum of source channels */
    int32_t Cout;           /* real output channels; stored channel pitch is out_pitch */
    int32_t out_pitch;      /* floats between consecutive pixels of `out` */
    int32_t upsample;       /* 1: bilinear x2 (align_corners=False) before the 3x3 conv */
    int32_t out_act;        /* gcpx_act in the epilogue (layers without a norm) */
    int32_t head_mode;      /* gcpx_head_mode (conv3x3 only) */
    const float* wpk;       /* dev: weights in MFMA fragment order (see video-gcp_amd/packing.py) */
    const float* bias;      /* dev: [CT*16], zero padded */
    float* out;             /* dev: NHWC raw output (may be NULL for GCPX_HEAD_DLM_MEAN) */
    float* images;          /* dev: NCHW [F][3][H][W] for the head modes that produce images */
    float* stats_partial;   /* dev: [gcpx_conv3x3_grid(a) or gcpx_conv4x4s2_grid()][2][CT*16] per-workgroup sum / sum-of-squares of the
                               raw output for training-mode BatchNorm, or NULL */
    const int32_t* raw_row_map; /* dev: [F] or NULL (output head only): frame f stores its raw parameters at row
                               raw_row_map[f] of `out`, or not at all when the entry is negative — only the nodes matched
                               to a ground-truth frame need their distribution parameters (frame_binding.py:91-92) */
    const int32_t* src_row_map; /* dev: [F] or NULL (non-upsampling convs and gcpx_conv_stage): frame f reads source frame
                               src_row_map[f]; a negative entry reads zeros (backward of the matched-frame gather) */
    const int32_t* src_row_frames; /* dev: [n_src_rows] or NULL: the inverse of src_row_map (row r is read by frame src_row_frames[r]).
                               Optional accelerator for the plain 3x3 conv: the kernel then walks the rows that exist instead of
                               testing every frame, and zero-fills the frames whose src_row_map entry is negative up front */
    int32_t n_src_rows;
    int32_t w_split_log2;   /* with wpk_split: the power of two the packed weights were scaled by (the kernel undoes it) */
    const void* wpk_split;  /* dev or NULL: the same weights as two f16 pieces in 16x16x32 fragment order
                               (packing.pack_dlm_head_split / pack_conv3x3_split).  When set, kernels that have a split-f16 form run
                               it: f32-equivalent results (error of the order of one f32 rounding per product) on the f16 matrix
                               pipes, see csrc/split_mfma.h.  NULL selects the exact f32 MFMA kernels */
    const int32_t* w_split_log2_dev; /* dev or NULL: when set, the scale exponent is read from here instead of w_split_log2
                               (weights re-split on the device after every optimizer step, gcpx_split_pack) */
    int32_t split_layout;   /* layout of wpk_split: GCPX_SPLIT_PLAIN (the conv's own 3x3 taps) or GCPX_SPLIT_ROWFOLD (upsampling blocks
                               with 32 -> 16 channels: the vertical half of the bilinear x2 folded into the weights, see
                               gcpx_fold_upsample_weights) */
    int32_t nll_rows;       /* GCPX_HEAD_DLM_NLL: rows of nll_target / row pitch of nll_partial */
    const float* nll_target; /* dev: NCHW [nll_rows][3][H][W] ground-truth frames in [-1, 1] (traj_seq) */
    float* nll_partial;     /* dev: [(H / 4) * (W / 16)][nll_rows]; rows no frame maps to are not written */
    const float* nll_row_weight; /* dev: [nll_rows] or NULL (GCPX_HEAD_DLM_NLL_GRAD): per-row factor of the gradient (pad_mask) */
    float nll_scale;        /* GCPX_HEAD_DLM_NLL_GRAD: d total / d nll_bt = w_rec / (B * prod(traj_seq.shape[1:])) (base_gcp.py:299-301) */
    int32_t _pad3;
    float* images_rows;     /* dev or NULL (split-f16 mixture heads): NCHW [rows][3][H][W]; every frame f with raw_row_map[f] >= 0 stores its
                               image at that row as well — the matched / kept frames in sequence order (tree_dense_rec.py:56-60,
                               tree.py:62-65) without a gather pass behind the head.  Rows no frame maps to are not written */
    int64_t images_rows_dup; /* != 0: a second copy of those rows at images_rows + images_rows_dup (floats) */
    /* Data gradient of a 16-channel layer with the NEXT step of the backward pass in its epilogue (split-f16 plain 3x3 conv with 16
       output channels, out_pitch 16): the layer whose output gradient this conv produces is BatchNorm + LeakyReLU(0.2) of the raw
       tensor bwd_r [F][H][W][16], so the kernel stores  g = conv * (bwd_scale * r + bwd_shift > 0 ? 1 : 0.2)  and leaves the
       per-workgroup sums of g and of g * (r - bwd_mean) * bwd_rstd in stats_partial [gcpx_conv3x3_grid(a)][2][16] — what gcpx_act_bwd
       (up = 0, no add) computes in a pass of its own over the same tensors.  NULL: plain data gradient */
    const float* bwd_r;
    const float *bwd_scale, *bwd_shift, *bwd_mean, *bwd_rstd;
    /* Upsampling blocks with 32 / 64 output channels in split-f16 (conv3x3_up32_split_kernel): a raw NHWC tensor
       [F / addend_frame_div][Hout][Wout][out_pitch] added to the conv's result (before the BatchNorm partial sums and the store), frame f
       reading frame f / addend_frame_div.  The decoder concatenates the skip activations of I_0 — the same for the N nodes of a sequence
       — to every node's features (tree_dense_rec.py:42 -> blox ConvDecoder with skips); a conv is linear in its input channels, so the
       skip half is convolved ONCE per sequence and arrives here, and the per-node launch walks the node's own channels only.  NULL: none */
    const float* addend;
    int32_t addend_frame_div;
    int32_t _pad4;
} gcpx_conv_args;

/* GCPX_SPLIT_HEAD32: the 100-channel mixture head's weights in 32x32x16 A-fragment order [9 taps][4 tiles][2 pieces][64][8]
   (packing.pack_head32_split / head32_index) for csrc/conv3x3_head32.hip — modes GCPX_HEAD_DLM_MEAN / _NLL / _NLL_GRAD only */
/* GCPX_SPLIT_ROWFOLD16: the row-folded weights of a 16 -> 16 channel upsampling block, two horizontal taps per k-step
   (packing.pack_conv3x3_fold16 / conv3x3_fold16_index over the [24][16][Cin] scratch of gcpx_fold_upsample_weights): the node half / the
   skip half of additional_conv_layer once the skip half is hoisted (gcpx_conv_args.addend) */
typedef enum gcpx_split_layout { GCPX_SPLIT_PLAIN = 0, GCPX_SPLIT_ROWFOLD = 1, GCPX_SPLIT_HEAD32 = 2, GCPX_SPLIT_ROWFOLD16 = 3 } gcpx_split_layout;

/* decoder block: (bilinear x2 upsample +) 3x3 conv, pad 1.  gcpx_conv3x3_grid(a) = number of workgroups that launch
   will use = rows of a->stats_partial (a->stats_partial must already be non-NULL in the query if it will be). */
int gcpx_conv3x3(const gcpx_conv_args* a, void* stream);
int gcpx_conv3x3_grid(const gcpx_conv_args* a);
/* encoder block: 4x4 conv, stride 2, pad 1 (NHWC sources); its stats_partial has gcpx_conv4x4s2_grid() rows */
int gcpx_conv4x4s2(const gcpx_conv_args* a, void* stream);
int gcpx_conv4x4s2_grid(void);
/* first encoder block: same conv on the NCHW 3-channel image tensor the model API receives.
   x: dev NCHW [F][3][Hin][Win]; out NHWC [F][Hin/2][Win/2][Cout] with bias + out_act applied. */
int gcpx_conv4x4s2_image(const float* x, const float* wpk, const float* bias, float* out,
                         int32_t F, int32_t Hin, int32_t Win, int32_t Cout, int32_t out_act, void* stream);

/* BatchNorm statistics -> (scale, shift):  scale = gamma * rsqrt(var + eps), shift = beta - mean * scale.
   partial: dev [n_partial][2][pitch] (fp32 partial sums written by the conv / gemm epilogues); column n
   belongs to channel n % C (pitch is a multiple of C); count = elements per channel.  If running_mean/var are non-NULL they are updated with `momentum` (training). */
int gcpx_bn_finalize(const float* partial, int32_t n_partial, int32_t pitch, int32_t C, double count,
                     const float* gamma, const float* beta, float eps, float* scale, float* shift,
                     float* running_mean, float* running_var, float momentum, float* mean_out, float* rstd_out,
                     void* stream);   /* mean_out / rstd_out: [C] batch statistics kept for the backward pass, or NULL */
/* n standard-normal numbers (Philox4x32-10 + Box-Muller) into out (16-byte aligned); state: dev uint64 [2] = {seed, offset}, the offset
   advanced by a second launch behind the draw, so a replayed graph draws fresh numbers.  Replaces the torch.randn behind
   Gaussian.sample() (tree_module.py:79-94) for models that draw their latent noise themselves. */
int gcpx_randn(float* out, int64_t n, uint64_t* state, void* stream);
/* eval-mode fold: scale/shift from running statistics */
int gcpx_bn_fold(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                 float eps, int32_t C, float* scale, float* shift, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Row GEMM with gathered, concatenated inputs:  out[r, :] = epi( concat_s X_s[map_s(r), :] @ W^T + bias ).
 *   replaces the Linear/LSTMCell/Conv1d/ConvTranspose(1x1->4x4)/Conv(4x4 valid) launches issued from
 *   gcp/prediction/models/tree/tree_lstm.py:43-49 (split_linear merge + HiddenStatePredictorModel),
 *   gcp/prediction/models/base_gcp.py:199 (ConvSeqEncodingModule) and the encoder head / decoder input.
 * Row r = (b, j), b = r / rpb, j = r % rpb.  A source row is ptr + b*sb + (j+shift)*sr  (zeros when j+shift is
 * outside [0, rpb)), or ptr + rowidx[r]*sr when rowidx is given (batchwise_index gather, inference.py:27-33).
 * ------------------------------------------------------------------------------------------------- */
typedef struct gcpx_row_src {
    const float* ptr;
    const int32_t* rowidx;  /* dev: [M] absolute row indices, or NULL */
    const float* scale;     /* dev: per-channel affine on load, channel = k % cmod; NULL = identity */
    const float* shiftv;
    int64_t sb, sr;         /* strides in floats */
    int32_t width;          /* K extent of this source (multiple of 16) */
    int32_t shift;          /* row shift inside the batch (conv1d taps) */
    int32_t act;
    int32_t cmod;
} gcpx_row_src;

typedef enum gcpx_gemm_epi {
    GCPX_EPI_NONE = 0,
    GCPX_EPI_LRELU = 1,
    GCPX_EPI_LSTM = 2,   /* N = 4H gate-interleaved (n = 4u+g, g in i,f,g,o): writes h, c; see below */
    GCPX_EPI_GAUSS_SAMPLE = 3 /* gcpx_gemm_group only — not a GEMM: out[r, n] = mu + exp(log_sigma) * eps with [mu | log_sigma] = the 2N-wide
                            rows of src[0] and eps = the N-wide rows of src[1] (Gaussian.sample / reparametrize, sequential.py:52): lets
                            the draw of the next VRNN step ride in a launch of this step instead of being a launch of its own */
} gcpx_gemm_epi;

struct gcpx_lstm_bwd_args;
typedef struct gcpx_gemm_args {
    gcpx_row_src src[6];
    int32_t nsrc;
    int32_t M, N, K, rpb;
    const float* wpk;       /* dev: [K/16][N/16][64][4] fragment-packed W[N][K] */
    const float* bias;      /* dev: [N] or NULL */
    float* out;             /* dev: row r at out + b*ob + j*orow (NONE / LRELU) */
    int64_t ob, orow;
    int32_t epi;
    int32_t nbatch;         /* 0/1: single problem; >1: blockIdx.z = b runs the same M x N x K problem with every source
                               pointer advanced by b*z_src_off, weights by b*z_w_off, bias by b*z_bias_off and out by
                               b*z_out_off floats (the 2*n_lstm_layers split_linear projections in one launch) */
    float* stats_partial;   /* dev: [gcpx_gemm_row_blocks(M, N)][2][N] per-row-block sum / sum of squares
                               of the output columns (training-mode BatchNorm), or NULL */
    /* LSTM epilogue: c' = sig(f)*c + sig(i)*tanh(g); h' = sig(o)*tanh(c') */
    const float* c_prev;    /* row r at c_prev + r*c_prev_stride */
    int64_t c_prev_stride;
    float* h_out;           /* row r at h_out + b*hb + j*hrow ; c_out likewise */
    float* c_out;
    int64_t hb, hrow;
    float* h_copy;          /* optional dense copy of h: row r at h_copy + r*H (input of the next layer) */
    int64_t z_src_off, z_w_off, z_bias_off, z_out_off;
    float* gates_out;       /* LSTM epilogue, optional: activated gates [M][H][4] = (i, f, g, o) kept for the backward pass */
    const void* wpk_split;  /* dev or NULL: the same weights as two f16 pieces, [K/32][N/16][2][64][8] (packing.pack_gemm_split; batch b at
                               byte offset b * z_w_off * 4).  When set, problems with enough rows to be bound by the f32 MFMA rate run
                               the split-f16 kernel (csrc/gemm_split.hip): f32-equivalent results, see split_mfma.h */
    const int32_t* w_split_log2_dev; /* dev or NULL: [nbatch] powers of two the packed pieces were scaled by; NULL = w_split_log2 */
    int32_t w_split_log2;
    int32_t _pad_split;
    void* x_planes;         /* dev or NULL: workspace for the activations as two f16 pieces in fragment order + x_exp [padded rows] row
                               exponents (sizes: gcpx_gemm_planes_workspace).  When set (with wpk_split) problems with many rows run as a
                               conversion pass + an LDS-DMA fed GEMM (csrc/gemm_planes.hip); same results class as the split kernel */
    int32_t* x_exp;
    int64_t x_planes_bytes; /* size of the x_planes allocation (checked against the problem) */
    const struct gcpx_lstm_bwd_args* lstm_bwd;
                            /* DEVICE copy of a gcpx_lstm_bwd_args, or NULL (GCPX_EPI_NONE, single problem, dense rows r = b*rpb + j of the
                               M rows the cell backward has too): the epilogue runs the LSTM cell backward of the layer this GEMM feeds —
                               output column u < H of row r IS that cell's d h from above, so dgates / dc_prev of (r, u) are written next
                               to out[r, u] and the separate gcpx_lstm_bwd launch (one of two per layer on a latency-bound chain)
                               disappears.  dh_dense of the struct is ignored; columns >= H (the d h_prev half of a [dx | dh] GEMM) are
                               plain outputs.  Same arithmetic as gcpx_lstm_bwd. */
} gcpx_gemm_args;

int gcpx_gemm(const gcpx_gemm_args* a, void* stream);
/* workspace of gcpx_gemm_args.x_planes / x_exp for an M x K activation matrix (x nbatch): bytes of the planes, number of int32 exponents */
int gcpx_gemm_planes_workspace(int32_t M, int32_t K, int32_t nbatch, int64_t* planes_bytes, int64_t* n_exp);
/* Several independent small-M problems (those gcpx_gemm runs as split-K 16 x 16 tiles) in ONE launch: the same layer of the
   prior / inference / generator nets of a VRNN step (sequential.py:49-54), or independent GEMMs of a tree level.
   gcpx_gemm_group_dims validates a HOST table of n (<= 16) problems (GCPX_ERR_UNSUPPORTED when one is not in that regime) and
   fills dims [n][4] + the grid size; the caller uploads table and dims once and replays gcpx_gemm_group. */
int gcpx_gemm_group_dims(const gcpx_gemm_args* host_table, int32_t n, int32_t* dims, int32_t* total_blocks);
int gcpx_gemm_group(const gcpx_gemm_args* dev_table, const int32_t* dev_dims, int32_t n, int32_t total_blocks, void* stream);
/* number of row blocks gcpx_gemm uses for an M x N problem (first dim of stats_partial) */
int gcpx_gemm_row_blocks(int32_t M, int32_t N);

/* ---------------------------------------------------------------------------------------------------
 * Fused Predictor MLP:  input(in->mid, LReLU), n_mid x (mid->mid, GroupNorm(8), LReLU), head(mid->out).
 *   replaces blox.torch.subnetworks.Predictor as called at tree_module.py:77 (prior), inference.py:35
 *   (posterior), tree_module.py:105 (MLP LSTM initialiser), misc.py:48 (length), frame_binding.py:71
 *   (existence), base_gcp.py:256 (state regressor), inverse_mdl.py:126 and cost_mdl.py:63.
 * One wavefront owns 16 rows end to end; hidden activations never leave LDS.
 * GCPX_MLP_GAUSS: out = 2*nz, mu = out[:nz], log_sigma = out[nz:]; additionally z = mu + exp(log_sigma) * eps
 * (Gaussian.sample / reparametrize, tree_module.py:79-94).
 * ------------------------------------------------------------------------------------------------- */
typedef enum gcpx_mlp_epi { GCPX_MLP_PLAIN = 0, GCPX_MLP_GAUSS = 1,
                             GCPX_MLP_TANH = 2 /* out = tanh(head): the non-LSTM subgoal predictor, tree_module.py:109-110 */ } gcpx_mlp_epi;

typedef struct gcpx_mlp_args {
    gcpx_row_src src[6];
    int32_t nsrc;
    int32_t M, rpb;
    int32_t in_dim, mid, n_mid, out_dim;
    const float* w_in;      /* dev: fragment-packed [in_dim/16][mid/16][64][4] */
    const float* b_in;      /* [mid] */
    const float* w_mid;     /* dev: n_mid x fragment-packed [mid/16][mid/16][64][4] */
    const float* b_mid;     /* [n_mid][mid] */
    const float* gn_gamma;  /* [n_mid][mid] */
    const float* gn_beta;   /* [n_mid][mid] */
    const float* w_out;     /* dev: fragment-packed [mid/16][out_pad/16][64][4] */
    const float* b_out;     /* [out_pad] */
    float gn_eps;
    float lrelu_slope;
    int32_t epi;
    int32_t out_split;      /* columns per output block (0 = out_dim): n -> block n / out_split */
    float* out;             /* row r, col n at out + (n/out_split)*oblk + b*ob + j*orow + n % out_split */
    int64_t ob, orow, oblk;
    /* GAUSS */
    const float* eps;       /* row r at eps + b*eb + j*erow, [nz] */
    int64_t eb, erow;
    float* z;               /* row r at z + b*zb + j*zrow, [nz] */
    int64_t zb, zrow;
    float* save;            /* optional, training: hidden activations kept for the backward pass, dense rows:
                               [0]: a_0 = LReLU(input layer) [M][mid]; then per hidden layer l: u_l (pre-GroupNorm) [M][mid]
                               and a_l (post GroupNorm + LReLU) [M][mid]  ->  (1 + 2*n_mid) * M * mid floats */
} gcpx_mlp_args;

int gcpx_mlp(const gcpx_mlp_args* a, void* stream);
/* Several independent Predictors of the same hidden width in ONE launch: the prior next to the posterior of a tree level
   (tree_module.py:77 + inference.py:27-35), the latent-space heads of run_auxilliary_models (base_gcp.py:234-262).
   gcpx_mlp_group_dims validates a HOST table of n (<= 16) problems and fills dims [n][4] = {first block, row blocks, head
   splits, 0} + the grid size; the caller uploads table and dims once (plan build) and replays gcpx_mlp_group. */
int gcpx_mlp_group_dims(const gcpx_mlp_args* host_table, int32_t n, int32_t* dims, int32_t* total_blocks);
int gcpx_mlp_group(const gcpx_mlp_args* dev_table, const int32_t* dev_dims, int32_t n, int32_t total_blocks, int32_t mid,
                   void* stream);
/* The same launch carrying one row GEMM (a HOST gcpx_gemm_args, plain epilogue) as additional workgroups: a level of the tree in front
   of its recurrent cell — prior + posterior Predictors (tree_module.py:77, inference.py:27-35) and the split_linear merge of the
   parents' hidden states (tree_lstm.py:43-48) depend on the previous level only, not on each other.  Results are those of
   gcpx_mlp_group + gcpx_gemm.  gcpx_mlp_group_gemm_supported: 1 when the GEMM's tiling and the group's size have a combined kernel
   (hidden width 128, at most one workgroup per CU, GEMM not on the split-f16 kernel); otherwise launch the two separately. */
int gcpx_mlp_group_gemm_supported(const gcpx_gemm_args* g, int32_t total_blocks, int32_t mid);
int gcpx_mlp_group_gemm(const gcpx_mlp_args* dev_table, const int32_t* dev_dims, int32_t n, int32_t total_blocks, int32_t mid,
                        const gcpx_gemm_args* g, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Balanced frame binding — integer bookkeeping, bit-exact with
 *   gcp/prediction/models/tree/frame_binding.py:42-65 (BalancedBinding, torch-1.3 Long/Long truncation),
 *   gcp/prediction/models/tree/frame_binding.py:28-34 (argmax gather) and
 *   gcp/evaluation/evaluation_matching.py:192-206 (BalancedEvalBinding.get_all_samples).
 * Tree nodes are addressed by depth-first position p in [0, N) (N = 2^L - 1).
 *   node_t      [B][N]  timestep of node p (int32)
 *   leave       [B][N]  1 if the node is kept (its c_n_prime row is non-zero)
 *   frame2node  [B][T]  depth-first position matched to frame t; for padded frames the ROOT (argmax of an
 *                       all-zero column = bf index 0)
 *   etilde_row  [L-level blocks, bf order: level l holds B*2^l entries, b-major]  b*T + node timestep:
 *               absolute row of inf_enc_seq[B*T] gathered for the posterior (inference.py:27-33)
 *   seq_len     [B]     end_ind + 1
 *   node2row    [B][N]  b*T + timestep for kept nodes, -1 otherwise (row of the matched-frame arrays; may be NULL)
 * ------------------------------------------------------------------------------------------------- */
int gcpx_balanced_binding(const int64_t* end_ind, int32_t B, int32_t L, int32_t T, int32_t* node_t,
                          int32_t* leave, int32_t* frame2node, int32_t* etilde_row, int32_t* seq_len,
                          int32_t* node2row, void* stream);

/* out[b][t] = src[b][idx[b][t] + idx_offset] for rows of `row_floats` floats (matched / pruned sequences);
   src has N rows per batch element; rows with idx < 0 are zero-filled (pad_sequence, base_gcp.py:242). */
int gcpx_gather_rows(const float* src, const int32_t* idx, float* out, int32_t B, int32_t T, int32_t N,
                     int32_t idx_offset, int64_t row_floats, void* stream);
/* gcpx_gather_rows for the rows its producer has NOT written: rows r with written[r] >= 0 are left untouched (the output head stores
   the frames matched to a row itself, gcpx_conv_args.images_rows; what remains are the padded rows: zeros, or the frame idx names) */
int gcpx_gather_rows_rest(const float* src, const int32_t* idx, float* out, int32_t B, int32_t T, int32_t N, int32_t idx_offset,
                          int64_t row_floats, const int32_t* written, void* stream);
/* Learned pairwise cost over a rollout (LearnedCostEstimate, gcp/planning/cem/cost_fcn.py:88-95):
   gcpx_seq_pairs builds the "next" operand nxt[i][t] = lat[i][t+1] (t+1 < len_i) else goal[i] (goal == NULL: the
   sequence's own last latent); the cost MLP runs on (lat, nxt) rows; gcpx_masked_row_sum adds the first len_i
   per-step costs of every candidate. */
int gcpx_seq_pairs(const float* lat, const int32_t* lengths, const float* goal, float* nxt, int32_t n, int32_t T,
                   int32_t nz, void* stream);
int gcpx_masked_row_sum(const float* vals, const int32_t* lengths, float* out, int32_t n, int32_t T, void* stream);
/* Hand-written planner costs over padded device rollouts (CostFcn subclasses, gcp/planning/cem/cost_fcn.py:10-77): x [n][T][ld]
   (the first D columns of a row are used), lengths [n], goal row of candidate i at goal + i*goal_stride (0: one goal for all) -> out [n].
   kind 0 EuclideanDistance, 1 EuclideanPathLength (dense only), 2 StepPathLength, 3 L2ImageCost (D = the image columns);
   final_step_weight multiplies the last step, dense = sum over steps instead of the last step's value (cost_fcn.py:16-23). */
int gcpx_rollout_cost(const float* x, int64_t ld, const int32_t* lengths, const float* goal, int64_t goal_stride, float* out, int32_t n,
                      int32_t T, int32_t D, int32_t kind, int32_t dense, float final_step_weight, void* stream);
/* dst[b][r] = src[b][r] for r < rows, rows of row_floats floats, independent batch strides (in rows): assembling
   images = cat(I_0, decoded) of the sequential model (sequential.py:57) without a torch.cat */
int gcpx_copy_rows(const float* src, float* dst, int32_t B, int32_t rows, int64_t row_floats, int64_t src_batch_rows,
                   int64_t dst_batch_rows, void* stream);
/* Gaussian.sample() for a recurrent cell's [mu | log_sigma] rows: z = mu + exp(log_sigma) * eps, rows r = (b, j) at
   base + b*{m,e,z}b + j*{m,e,z}r  (VRNN step of sequential.py:51-54) */
int gcpx_gauss_sample(const float* muls, int64_t mb, int64_t mr, const float* eps, int64_t eb, int64_t er, float* z,
                      int64_t zb, int64_t zr, int32_t M, int32_t rpb, int32_t nz, void* stream);
/* idx[b][t] = t (t <= end_ind[b]) else -1; seq_len[b] = end_ind[b] + 1: prefix selection of the sequential model
   (sequential.py:88-93,130-131) in the same form as gcpx_compact_index */
int gcpx_seq_index(const int64_t* end_ind, int32_t B, int32_t T, int32_t* idx, int32_t* seq_len, void* stream);
/* graph-capturable memset(0) (recurrent state reset, lstm_init='zero') */
int gcpx_fill_zero(void* ptr, int64_t nbytes, void* stream);
/* compaction of kept nodes: dst_idx[b][k] = k-th depth-first position with leave==1 (k < seq_len[b]), else -1 */
int gcpx_compact_index(const int32_t* leave, int32_t B, int32_t N, int32_t T, int32_t* dst_idx, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Losses (deterministic reductions).
 *   gcpx_dlm_nll:  decoder.nll(matched distr, traj_seq) of frame_binding.py:88-99 for the discrete-logistic-mixture head:
 *                  params [rows][npix][pitch] in the head kernel's channel order, target NCHW [rows][3][npix];
 *                  nll_out[row] = sum over pixels of the negative log-likelihood; rows whose row_weight (pad_mask) is 0
 *                  are skipped and report 0.
 *   gcpx_gauss_nll: the same for the gaussian head (mu = images).
 *   gcpx_kl_gauss: KLDivLoss2(q_z, p_z) of inference.py:38-43: per batch element, sum over nodes and dims of
 *                  max(KL, free_nats); q/p rows are [mu | log_sigma] at base + b*batch_stride + n*node_stride.
 *   gcpx_loss_combine: length CE (misc.py:53-56), existence BCE (frame_binding.py:80-86), state L2
 *                  (base_gcp.py:281-286), the weighted total of get_total_loss (base_gcp.py:294-304).
 *                  out[0..8] = dense_img_rec, kl, len_pred, existence_predictor, state_regression, total, nll,
 *                  action_reconst (inverse_mdl.py:181-191), cost_estimation (cost_mdl.py:59-62).
 * ------------------------------------------------------------------------------------------------- */
typedef struct gcpx_loss_args {
    const float* nll_bt;          /* [B*T] */
    const float* pad_mask;        /* [B*T] */
    const float* kl_b;            /* [B] or NULL */
    const float* len_logits;      /* [B][T] or NULL */
    const int64_t* end_ind;       /* [B] */
    const float* existence;       /* [B][N] logits or NULL */
    const int32_t* leave;         /* [B][N] */
    const float* regressed_state; /* [B][T][state_dim] or NULL */
    const float* state_target;    /* [B][T][state_dim] or NULL */
    const int32_t* seq_len;       /* [B] */
    float* out;                   /* [16] */
    int32_t B, T, N, state_dim;
    float w_rec, w_kl, w_len, w_exist, w_state;
    float total_div;
    /* inverse model, sampled pair (inverse_mdl.py:136-191): action_pred [B][n_actions] against actions[b][inv_t0[b]] */
    const float* action_pred;     /* [B][n_actions] or NULL */
    const float* action_seq;      /* inputs.actions [B][T-1][n_actions] */
    const int64_t* inv_t0;        /* [B] */
    /* cost model (cost_mdl.py:42-62): cost_pred [B] against the ground-truth path cost [B] */
    const float* cost_pred;       /* [B] or NULL */
    const float* cost_target;     /* [B] */
    int32_t n_actions;
    float w_action, w_cost;
    /* weights of the state regression when they differ from pad_mask (gcp_sequential: pad_mask above is the reconstruction
       weight with frame 0 zeroed, sequential.py:63-66, while the regressor still sees frame 0, base_gcp.py:281-286); NULL = pad_mask */
    const float* state_mask;      /* [B*T] or NULL */
    /* KL weight under a schedule (kl_weight_burn_in, base_gcp.py:121-128: the weight ramps from 0 to its target over the first
       iterations): when set, the weight is read from this device scalar instead of w_kl, so a captured graph sees the current value */
    const float* w_kl_dev;
} gcpx_loss_args;

int gcpx_dlm_nll(const float* params, const float* target, const float* row_weight, float* nll_out, int32_t rows,
                 int32_t npix, int32_t pitch, int32_t n_mix, void* stream);
int gcpx_gauss_nll(const float* mu, const float* target, const float* log_sigma, float* nll_out, int32_t rows,
                   int32_t nelem, void* stream);
int gcpx_kl_gauss(const float* qz, const float* pz, int32_t B, int32_t N, int32_t nz, int64_t batch_stride,
                  int64_t node_stride, float free_nats, const float* node_weight /* b*weight_bstride + n, or NULL */,
                  int64_t weight_bstride, float* kl_out, void* stream);
int gcpx_loss_combine(const gcpx_loss_args* a, void* stream);
/* The same in two launches for a forward whose serial tail matters: gcpx_loss_pre = gcpx_kl_gauss (same arguments) + the five terms
   that need no decoded frame (length cross entropy, existence BCE, state / action / cost regression -> out[2], [3], [4], [7], [8]),
   one launch that can be issued before the decoder; gcpx_loss_final = reconstruction + KL sums and the weighted total
   (base_gcp.py:264-304) from nll_bt, kl_b and those five values. */
int gcpx_loss_pre(const gcpx_loss_args* a, const float* qz, const float* pz, int32_t N, int32_t nz, int64_t batch_stride,
                  int64_t node_stride, float free_nats, const float* node_weight, int64_t weight_bstride, float* kl_out, void* stream);
int gcpx_loss_final(const gcpx_loss_args* a, void* stream);


/* ---------------------------------------------------------------------------------------------------
 * Auxiliary models, training-time paths, and the sampled sequence length (csrc/aux.hip).
 * ------------------------------------------------------------------------------------------------- */
/* InverseModel.sample_offsets (inverse_mdl.py:84-104) + CostModel._general_cost's index draws (cost_mdl.py:105-107) from four
   uniform numbers per sequence, u [4][B] in [0,1):  inv_t0 in {0..end_ind-temp_dist}, inv_t1 = inv_t0 + {1..temp_dist},
   cost_start in {0..end_ind-1}, cost_end in {cost_start+1..end_ind}.  (The reference draws with np.random; callers that
   want its exact stream feed the indices instead.) */
int gcpx_aux_sample_indices(const int64_t* end_ind, const float* u, int32_t B, int32_t temp_dist, int64_t* inv_t0, int64_t* inv_t1,
                            int64_t* cost_start, int64_t* cost_end, void* stream);
/* the same from four STANDARD-NORMAL numbers per sequence, n [4][B] (u = Phi(n) is uniform on (0, 1)): the draws then come out of the
   one generator launch that also fills the latent noise of Gaussian.sample(), and this launch sits inside the forward's graph */
int gcpx_aux_sample_indices_gauss(const int64_t* end_ind, const float* n, int32_t B, int32_t temp_dist, int64_t* inv_t0, int64_t* inv_t1,
                                  int64_t* cost_start, int64_t* cost_end, void* stream);
/* rows [4][B] (int32): absolute rows b*T + inv_t0, b*Wd + inv_t1, b*Wd + cost_start, b*Wd + cost_end — the gather-on-load
   sources of `inv_mdl.action_pred(enc_traj_seq[b,t0], model_enc_seq[b,t1])` (inverse_mdl.py:149-169) and
   `cost_mdl.cost_pred(model_enc_seq[b,start], model_enc_seq[b,end])` (cost_mdl.py:108-109, :50) */
int gcpx_aux_index_rows(const int64_t* inv_t0, const int64_t* inv_t1, const int64_t* cost_start, const int64_t* cost_end, int32_t B,
                        int32_t T, int32_t Wd, int32_t* rows, void* stream);
/* ground-truth cost of CostModel._general_cost with EuclideanPathLength(dense_cost=True) (cost_mdl.py:110-111,
   cost_fcn.py:14-21,49-54): out[b] = sum_{t=start}^{end-1} sum_row || x[b,t+1,row,:] - x[b,t,row,:] ||_2 over x [B][T][rows][row_len]
   (images: rows = 3*H, row_len = W).  partial: scratch [B][rows]. */
int gcpx_path_cost(const float* x, const int64_t* start_idx, const int64_t* end_idx, int32_t B, int32_t T, int32_t rows,
                   int32_t row_len, float* partial, float* out, void* stream);
/* get_end_ind under val_mode(pred_length=True) (base_gcp.py:219-226): end_ind[b] = max(min_len, inverse CDF of
   softmax(logits[b]) at u[b]) — the OneHotCategorical draw (misc.py:49) with the uniform number fed in */
int gcpx_sample_length(const float* logits, const float* u, int32_t B, int32_t T, int32_t min_len, int64_t* end_ind, void* stream);


/* ---------------------------------------------------------------------------------------------------
 * Evaluation metrics (csrc/metrics.hip): mse / psnr / ssim of generated against ground-truth sequences —
 * Evaluator.compute_metrics (gcp/evaluation/compute_metrics.py:123-130; blox.torch.evaluation is absent, spec in
 * video-gcp_amd/evaluation.py).  tgt [B][T][C][H][W] in [-1,1]; est: a pool of frames [R][C][H][W]; frame_map [B*T] = pool
 * frame compared with target frame (b, t) (negative: skipped; NULL: identity); per sequence the frames t in [first[b], last[b])
 * are averaged (the harness crops the two conditioning frames).  scratch: 2*B*T*C doubles.  out [B][3] = mse, psnr, ssim.
 * ------------------------------------------------------------------------------------------------- */
int gcpx_image_metrics(const float* est, const float* tgt, const int32_t* frame_map, const int32_t* first, const int32_t* last,
                       int32_t B, int32_t T, int32_t C, int32_t H, int32_t W, double* scratch, float* out, void* stream);


/* ===================================================================================================
 * Training step: explicit backward pass, optimizer.
 *   replaces `losses.total.value.backward(); optimizer.step()` of gcp/prediction/train.py:155-163 (torch autograd +
 *   blox.torch.radam.RAdam, gcp_builder.py:178-179).  There is no autograd here: every gradient is a launch.
 * Data gradients of Linear / conv layers reuse gcpx_gemm / gcpx_conv3x3 with transposed weight packs.
 * ================================================================================================= */

/* ---- weight gradients: dW[n][k] = sum_r dY[r][n] * X[r][k] ---- */
typedef enum gcpx_wgrad_mode {
    GCPX_WG_ROWS = 0,      /* X row r at x + b*sb + (j+shift)*sr (b = r / rpb, j = r % rpb; zero outside [0,rpb)) or x + rowidx[r]*sr */
    GCPX_WG_CONV1D = 1,    /* k = (tap, ci), 3 taps along j: X[r][k] = x[b][j + tap - 1][ci] */
    GCPX_WG_CONV3X3 = 2,   /* k = (tap, ci): rows are pixels (f, y, x) of an NHWC [F][H][W][Cin] tensor, pad 1 */
    GCPX_WG_CONV4X4S2 = 3  /* k = (tap, ci), 16 taps: rows are output pixels (f, oy, ox) of the stride-2 conv over NHWC [F][H][W][Cin] */
} gcpx_wgrad_mode;
typedef enum gcpx_wgrad_map {
    GCPX_WMAP_LINEAR = 0,  /* dst[n*ldw + k_off + k] */
    GCPX_WMAP_CONV = 1,    /* k = (tap, ci): dst[(n_map[n]*Cin + ci)*ntap + tap]   (torch conv weight [Cout][Cin][taps]) */
    GCPX_WMAP_CONVT = 2    /* n = (tap, co), k = ci: dst[(k*Cout + co)*ntap + tap] (ConvTranspose2d weight [Cin][Cout][taps]) */
} gcpx_wgrad_map;

typedef struct gcpx_wgrad_args {
    const float* dy;        /* dev: [R][ldy] output gradient rows (dense) */
    const float* x;         /* dev: input operand (see mode) */
    const int32_t* rowidx;  /* ROWS: absolute row gather, or NULL */
    const int32_t* frame_map; /* conv modes: X frame = frame_map[f] (negative: zeros), or NULL */
    const float* scale;     /* per-channel affine + activation applied to X on load (channel = k % cmod, or ci), or NULL */
    const float* shiftv;
    float* out;             /* direct: dW[n*ldw + k_off + k]; partial: [nsplit][n_valid][K] */
    int64_t ldy, sb, sr, ldw;
    int64_t dy_sb;          /* 0: dy rows are dense (row r at dy + r*ldy); else row (b, j) at dy + b*dy_sb + j*ldy, b = r / dy_rpb */
    int32_t R, N, n_valid;  /* rows; columns of dy to read (N % 4 == 0 when > 16); rows of dW to write */
    int32_t K, mode, Cin, H, W, rpb, shift, act, cmod;
    int32_t k_off, accumulate, partial, nsplit;
    int32_t dy_rpb;
    int32_t nbatch;         /* > 1 (direct mode): blockIdx.z = b runs the same problem with dy advanced by b*z_dy_off, x by b*z_x_off,
                               out by b*z_out_off and dbias by b*z_bias_off floats (the 2*n_lstm_layers split_linear projections) */
    float* dbias;           /* optional (direct mode): dbias[n] (+)= sum_r dy[r][n], fused bias gradient */
    float* dbias2;          /* optional second destination of the same sums (LSTM b_ih / b_hh; not batched) */
    int64_t z_dy_off, z_x_off, z_out_off, z_bias_off;
    int32_t split_f16;      /* != 0: a direct-mode ROWS problem of whole 128 x 128 blocks of dW with >= 256 plain rows runs on the f16 matrix
                               pipes with f32-equivalent arithmetic (csrc/wgrad_rows_split.hip: both operands split into two f16 pieces in
                               the kernel, three MFMAs per product, f32 accumulate); everything else, and 0, runs the exact f32 kernel */
    int32_t _pad_split;
} gcpx_wgrad_args;

int gcpx_wgrad(const gcpx_wgrad_args* a, void* stream);
/* Grouped launch of up to 64 independent direct-mode problems that use the same kernel variant (the small per-level weight gradients of
   tree_module.py:67-114 / tree_lstm.py:43-49): gcpx_wgrad_classify gives the variant and workgroup count of one problem (host query);
   tab: DEVICE copy of the descriptors, block_start: DEVICE [nprob] first workgroup of each problem, total_blocks = their sum.
   The problems must write disjoint outputs. */
int gcpx_wgrad_classify(const gcpx_wgrad_args* a, int32_t row_split, int32_t* variant, int32_t* nblocks);   /* row_split: -1 = this problem's own heuristic, 0 / 1 = force */
int gcpx_wgrad_group(const gcpx_wgrad_args* tab, const int32_t* block_start, int32_t nprob, int32_t total_blocks, int32_t variant,
                     void* stream);
int gcpx_wgrad_reduce(const float* partial, int32_t nsplit, int32_t N, int32_t K, float* dst, int32_t map_mode, int32_t Cin,
                      int32_t ntap, int32_t Cout, const int32_t* n_map, int64_t ldw, int32_t k_off, int32_t accumulate,
                      void* stream);
/* LDS-tiled weight gradient of a 3x3 conv (pad 1): dy [F*H*W][ldy] (Cout rounded up to 16 columns are read), u NHWC
   [F][H][W][Cin]; writes partial [grid][ceil16(Cout)][9*Cin] (k = tap*Cin + ci), one row block per persistent workgroup,
   to be combined by gcpx_wgrad_reduce(map CONV).  H, W powers of two, W in {8, 16, 32k}. */
int gcpx_wgrad_conv3x3(const float* dy, int32_t ldy, const float* u, int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                       float* partial, int32_t grid, void* stream);
/* The same contract on the f16 matrix pipes with f32-equivalent arithmetic (csrc/wgrad_conv_split.hip: both operands split into two
   f16 pieces in the kernel, three MFMAs per product, f32 accumulate, running power-of-two scales per workgroup).  Shapes it does not
   cover run on gcpx_wgrad_conv3x3. */
int gcpx_wgrad_conv3x3_split(const float* dy, int32_t ldy, const float* u, int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                             float* partial, int32_t grid, void* stream);
/* The same for an UPSAMPLING block with 16 output channels, reading the block's own low-resolution sources: `a` is the block's forward
   descriptor as gcpx_conv_stage takes it (src / nsrc / F / Hin / Win / Hout / Wout / Cin, upsample = 1, no src_row_map); the kernel forms
   the bilinear x2 of the concatenated, normalised + activated sources per tile, so the operand tensor gcpx_conv_stage would write
   (additional_conv_layer at c2: 1.07 GB per step, read back by the weight gradient) never exists.  GCPX_ERR_UNSUPPORTED (nothing
   launched) when the shape has no fused form: Cout <= 16, Cin % 32 == 0, source widths % 16 == 0, W in {8, 16, 32 k}. */
int gcpx_wgrad_conv3x3_split_up(const float* dy, int32_t ldy, const gcpx_conv_args* a, int32_t Cout, float* partial, int32_t grid,
                                void* stream);
/* ... and for a NON-upsampling conv whose operand is LeakyReLU(0.2)(scale * x + shift) of a raw tensor x [Fx][H][W][Cin] read through a
   frame map (operand frame f = frame frame_map[f] of x; a negative entry must carry a zero dy): the output head's weight gradient reads
   the last decoder block's raw output at the matched nodes — the gathered, activated copy gcpx_conv_stage would write (335 MB at c2)
   never exists.  frame_map, or scale + shift, may be NULL.  bias_partial (optional; the 112-column head form only): [grid][112]
   per-workgroup column sums of dy over every pixel = the bias gradient, from the same staged tiles (sum over the grid with
   gcpx_wgrad_reduce): no gcpx_colsum pass over dy (2.35 GB at c2). */
int gcpx_wgrad_conv3x3_split_src(const float* dy, int32_t ldy, const float* x, const int32_t* frame_map, const float* scale,
                                 const float* shift, int32_t F, int32_t H, int32_t W, int32_t Cin, int32_t Cout, float* partial,
                                 float* bias_partial, int32_t grid, void* stream);
/* bias gradient: dst[n] (+)= sum_r dy[r][n] (rows addressed like gcpx_wgrad_args.dy); dst2 = optional second destination
   (LSTM b_ih and b_hh); with nsplit > 1 the row range is split and partial [nsplit][N] is written instead of dst */
int gcpx_colsum(const float* dy, int64_t ldy, int32_t R, int32_t N, int32_t dy_rpb, int64_t dy_sb, int32_t nsplit, float* partial,
                float* dst, float* dst2, int32_t accumulate, void* stream);
/* dst[i] (+)= sum_{p < n} partial[p*stride + i], i < len (per-workgroup partial sums of norm parameters) */
int gcpx_reduce_partials(const float* partial, int32_t n, int64_t stride, int32_t len, float* dst, int32_t accumulate, void* stream);

/* ---- LSTM cell backward (HiddenStatePredictorModel inside tree_lstm.py:43-49) ---- */
typedef struct gcpx_lstm_bwd_args {
    const float* gates;     /* [M][H][4] activated gates saved by the forward (gcpx_gemm_args.gates_out) */
    const float* c_prev;    /* row r at c_prev + r*c_prev_stride */
    const float* c_new;     /* row (b, j) at c_new + b*pb + j*prow */
    const float* dh_dense;  /* [M] rows at dh_dense + r*dh_stride: gradient from the layer above, or NULL */
    const float* dh_pos;    /* gradient w.r.t. the stored hidden state h (from the children), at dh_pos + b*pb + j*prow, or NULL */
    const float* dc_pos;    /* same for the stored cell state c, or NULL */
    float* dgates;          /* [M][4H] pre-activation gate gradients, torch gate-major order (i | f | g | o) */
    float* dc_prev;         /* row r at dc_prev + r*dcp_stride */
    int64_t c_prev_stride, pb, prow, dh_stride, dcp_stride;
    int32_t M, H, rpb, _pad;
} gcpx_lstm_bwd_args;
int gcpx_lstm_bwd(const gcpx_lstm_bwd_args* a, void* stream);

/* ---- Predictor MLP pieces ---- */
/* GroupNorm + LeakyReLU backward on dense rows: du = dGN(u) applied to da * lrelu'; partial: [gcpx_gn_bwd_blocks(M)][2][C]
   per-workgroup sums of (d gamma, d beta) */
int gcpx_gn_lrelu_bwd(const float* u, const float* da, const float* gamma, const float* beta, float* du, float* partial,
                      int32_t M, int32_t C, int32_t groups, float eps, float slope, void* stream);
int gcpx_gn_bwd_blocks(int32_t M);
/* Fused data-gradient chain of one Predictor MLP (mirror of gcpx_mlp; backward of tree_module.py:77 / inference.py:27-35, which
   the reference gets from autograd): dout [M][ldo] -> head^T -> (GroupNorm + LReLU backward -> mid_l^T) x n_mid -> LReLU
   backward -> per input split dx_i = du0 @ W_in[:, split i].  `save` = the buffer gcpx_mlp_args.save filled in the forward.
   Writes every du_l ([M][mid]; operands of the weight-gradient GEMMs), the per-workgroup GroupNorm parameter sums
   gn_partial[l] [gcpx_mlp_bwd_blocks(M)][2][mid] (reduce with gcpx_reduce_partials) and the dx_i rows (b, j) at
   out + b*ob + j*orow (b = r / rpb).  wT_*: gcpx_gemm packs of the transposed weights. */
typedef struct gcpx_mlp_bwd_dx {
    const float* wT;        /* pack of W_in[:, split]^T: [mid / 16][width / 16][64][4] */
    float* out;
    int64_t ob, orow;
    int32_t width, _pad;
} gcpx_mlp_bwd_dx;
typedef struct gcpx_mlp_bwd_args {
    const float* dout;
    const float* save;
    const float* wT_out;    /* [out_pad / 16][mid / 16][64][4] */
    const float* wT_mid[4];
    const float* gn_gamma[4];
    const float* gn_beta[4];
    float* du[5];           /* du[0]: input layer (after its LReLU backward); du[1 + l]: hidden layer l */
    float* gn_partial[4];
    gcpx_mlp_bwd_dx dx[4];
    int64_t ldo;
    int32_t M, rpb, mid, n_mid, out_pad, ndx;
    float gn_eps, lrelu_slope;
} gcpx_mlp_bwd_args;
int gcpx_mlp_bwd(const gcpx_mlp_bwd_args* a, void* stream);
/* nprob (<= GCPX_MLP_BWD_GROUP_MAX) Predictors of one hidden width whose chains are independent, in ONE launch (blockIdx.y = the
   Predictor; tab: HOST array, copied into the kernel arguments): a tree level's posterior and prior share the level's latency-bound
   chain.  Bit for bit what one gcpx_mlp_bwd per Predictor writes. */
#define GCPX_MLP_BWD_GROUP_MAX 4
int gcpx_mlp_bwd_group(const gcpx_mlp_bwd_args* tab, int32_t nprob, void* stream);
int gcpx_mlp_bwd_blocks(int32_t M);
/* dx[r][c] = dy[r][c] * (1 - y[r][c]^2), backward of y = tanh(u) (GCPX_MLP_TANH; tree_module.py:109-110): dy and y rows (b, j) at
   base + b*sb + j*sr, dx dense [B*rpb][width] */
int gcpx_tanh_bwd_rows(const float* dy, const float* y, float* dx, int64_t sb, int64_t sr, int32_t B, int32_t rpb, int32_t width,
                       void* stream);
/* dx[i] = dy[i] * (a[i] > 0 ? 1 : slope) */
int gcpx_lrelu_bwd(const float* a, const float* dy, float* dx, int64_t n, float slope, void* stream);

/* ---- latent variables ---- */
/* d KL(q||p) (inference.py:38-43): writes d q / d p rows ([mu | log_sigma]) at the same addresses as qz / pz rows */
int gcpx_kl_bwd(const float* qz, const float* pz, float* dqz, float* dpz, int32_t B, int32_t N, int32_t nz, int64_t batch_stride,
                int64_t node_stride, float free_nats, float coef, void* stream);
/* the same with a per-node weight (the flat VRNN's KL term carries pad_mask[:, 1:], sequential.py:63-66):
   gradient rows scaled by node_weight[b * weight_bstride + n] */
int gcpx_kl_bwd_weighted(const float* qz, const float* pz, float* dqz, float* dpz, int32_t B, int32_t N, int32_t nz, int64_t batch_stride,
                         int64_t node_stride, float free_nats, float coef, const float* node_weight, int64_t weight_bstride, void* stream);
/* the general form with the scheduled part of the weight in device memory: gradient rows scaled by coef * (*coef_dev) (* node_weight
   when given) — the KL weight burn-in (base_gcp.py:121-128) */
int gcpx_kl_bwd_scheduled(const float* qz, const float* pz, float* dqz, float* dpz, int32_t B, int32_t N, int32_t nz, int64_t batch_stride,
                          int64_t node_stride, float free_nats, float coef, const float* node_weight, int64_t weight_bstride,
                          const float* coef_dev, void* stream);
/* one tree level: dq_out[r] = dqz_pos[r] + [dz, dz * exp(log_sigma_q) * eps] (reparametrised sample backward,
   tree_module.py:86-94), dp_out[r] = dpz_pos[r]; rows r = (b, j) of the level; *_pos at base + b*pb + j*prow;
   dz = dz0[r*ldz0 ..] (+ dz1[r*ldz1 ..]); eps at eps + b*eb + j*erow */
int gcpx_latent_bwd(const float* dqz_pos, const float* dpz_pos, const float* qz_pos, int64_t pb, int64_t prow, const float* eps,
                    int64_t eb, int64_t erow, const float* dz0, int64_t ldz0, const float* dz1, int64_t ldz1, float* dq_out,
                    float* dp_out, int32_t M, int32_t rpb, int32_t nz, void* stream);

/* gradients of one level's inputs accumulated into the parents' slots of a position-layout buffer (the backward of
   tree_utils.py:37-44 interleave + the context broadcast of tree_module.py:97-101) */
typedef struct gcpx_tree_accum_src {
    const float* ptr;       /* dense [B*n][ld] */
    int64_t ld;
    int32_t off_left, off_right, off_ctx0, off_ctxg;   /* column offsets of the e_l / e_r / e_0 / e_g parts, -1 = absent */
    int32_t dst_col;        /* column offset inside the destination row */
    int32_t _pad;
} gcpx_tree_accum_src;
typedef struct gcpx_tree_accum_args {
    gcpx_tree_accum_src src[6];
    float* dst;             /* slot 0 of batch element 0 */
    int64_t dst_sb;         /* floats between batch elements */
    int64_t slot_stride;    /* floats between consecutive parent slots of this level */
    int32_t nsrc, B, n, width;
} gcpx_tree_accum_args;
int gcpx_tree_accum(const gcpx_tree_accum_args* a, void* stream);
/* backward of the posterior's batchwise_index gather (inference.py:27-33): out[b][t] = sum over nodes p with node_t[b][p] == t
   of det[b][p] (rows of nz floats at det + b*db + p*dp) */
int gcpx_timestep_scatter(const float* det, int64_t db, int64_t dp, const int32_t* node_t, float* out, int32_t B, int32_t N,
                          int32_t T, int32_t nz, void* stream);

/* dst row (b, j) at dst + b*dst_sb + j*dst_sr  +=  src1[r] (+ src2[r]), dense sources [B*rpb][width] */
int gcpx_add_rows(float* dst, int64_t dst_sb, int64_t dst_sr, const float* src1, const float* src2, int32_t B, int32_t rpb,
                  int32_t width, void* stream);
/* rows (b, j), j < rpb, of `width` floats between two strided layouts (row at base + b*sb + j*sr): mode 0 dst = src, 1 dst += src,
   2 dst[b] += sum over j of src[b][j] (dst_sr unused; fixed summation order).  Gradient bookkeeping of the recurrent rollout
   (sequential.py:49-54 backward): x_t / context slices of the per-step embedding gradients */
int gcpx_rows_strided(float* dst, int64_t dst_sb, int64_t dst_sr, const float* src, int64_t src_sb, int64_t src_sr, int32_t B, int32_t rpb,
                      int32_t width, int32_t mode, void* stream);
/* rows r of `row_floats` floats with row2frame[r] < 0 are set to zero (rows of the matched-frame gradient no decoded frame maps to:
   padded frames t > end_ind; the head kernel in GCPX_HEAD_DLM_NLL_GRAD mode writes only mapped rows) */
int gcpx_zero_unmapped_rows(float* ptr, int64_t row_floats, const int32_t* row2frame, int32_t rows, void* stream);
/* out[b][t] = idx[b][t] + b*stride (per-sequence node index -> absolute frame index) */
int gcpx_index_offset(const int32_t* idx, int32_t* out, int32_t B, int32_t T, int32_t stride, void* stream);
/* inv[r] = i for every i < n with fwd[i] = r >= 0 (fwd injective on its non-negative entries), -1 for rows nobody maps to:
   the src_row_frames companion of a src_row_map (gcpx_conv_args) */
int gcpx_index_inverse(const int32_t* fwd, int32_t n, int32_t* inv, int32_t n_inv, void* stream);

/* ---- conv stacks ---- */
/* gradient w.r.t. the raw (pre-norm) output of a layer from the gradient w.r.t. its activated output:
   dy = (T(da) + add) * act'(scale*r + shift), where T is identity, the transposed bilinear x2 upsample (up = 1) and / or
   the sum over `fsum` consecutive frames (backward of the skip broadcast); with mean/rstd also the per-workgroup partial
   sums [gcpx_act_bwd_blocks()][2][C] of (dy, dy * x_hat) the BatchNorm backward needs. */
typedef struct gcpx_actbwd_args {
    const float* da;        /* NHWC [F*fsum][H*(1+up)][W*(1+up)][ldc], channels c_off .. c_off+C */
    const float* add;       /* [F][H][W][C] or NULL */
    const float* r;         /* [F][H][W][C] raw output (or activated output when scale == NULL) or NULL (no activation) */
    const float* scale;
    const float* shift;
    const float* mean;      /* [C] batch statistics, or NULL (no norm: no partial sums) */
    const float* rstd;
    float* dy;              /* [F][H][W][C] */
    float* stats_partial;
    int64_t ldc;
    int32_t c_off, up, fsum, act, F, H, W, C;
} gcpx_actbwd_args;
int gcpx_act_bwd(const gcpx_actbwd_args* a, void* stream);
int gcpx_act_bwd_blocks(void);
/* gcpx_act_bwd (up = 1, fsum = 1) for the channels a->c_off .. + a->C of an upsampling decoder block's input gradient AND, from the same pass
   over the tensor, the skip-connection half: ds [F / rpb][H][W][Cs] = sum over the rpb frames of a sequence of the (bilinear-transposed)
   channels c_off_s .. + Cs (the two gcpx_act_bwd launches of training.py's decoder backward, tree_dense_rec.py:42 / base_gcp.py:190) */
int gcpx_act_skip_bwd(const gcpx_actbwd_args* a, float* ds, int32_t c_off_s, int32_t Cs, int32_t rpb, void* stream);
/* BatchNorm backward, second half: coef[0..C) = gamma*rstd, coef[C..2C) = mean(dy), coef[2C..3C) = mean(dy*x_hat);
   d gamma / d beta written (accumulated) to the gradient buffers */
int gcpx_bn_bwd_finalize(const float* partial, int32_t n_partial, int32_t C, double count, const float* gamma, const float* rstd,
                         float* coef, float* dgamma, float* dbeta, int32_t accumulate, void* stream);
/* dr = coef0 * (dy - coef1 - x_hat * coef2), in place over dy; channel = index % C */
int gcpx_bn_bwd_apply(float* dy, const float* r, const float* mean, const float* rstd, const float* coef, int64_t n, int32_t C,
                      void* stream);
/* materialise the (upsampled, concatenated, normalised + activated) input of a 3x3 conv block for its weight gradient:
   uses src / nsrc / F / Hin / Win / Hout / Wout / Cin / upsample / src_row_map / out of gcpx_conv_args; out NHWC [F][Hout][Wout][Cin] */
int gcpx_conv_stage(const gcpx_conv_args* a, void* stream);
/* backward of the stride-2 4x4 conv w.r.t. its input: dx[f][iy][ix][ci] = sum of the (at most 4) taps of
   dcol [F*H/2*W/2][16*Cin] (k = (tap, ci)) that touched the pixel */
int gcpx_col2im4x4s2(const float* dcol, float* dx, int32_t F, int32_t H, int32_t W, int32_t Cin, void* stream);
/* im2col of the NCHW 3-channel image for the first encoder conv's weight gradient: col [F*H/2*W/2][48], k = (ci, ky, kx) */
int gcpx_im2col_image(const float* x, float* col, int32_t F, int32_t H, int32_t W, void* stream);
/* Weight + bias gradient of the encoder's first layer (4x4 stride-2 pad-1 conv on the NCHW image, LeakyReLU, no norm) in one launch:
   (da + add) * slope(r) against the image patches (csrc/wgrad_image.hip; replaces gcpx_act_bwd + gcpx_im2col_image + gcpx_wgrad +
   gcpx_colsum for that layer).  da / add (or NULL) / r: NHWC [F][S/2][S/2][16]; image [F][3][S][S]; partial [grid][16*48 + 16]
   (weights [co][ci*16 + ky*4 + kx], then the bias), to be summed by gcpx_reduce_partials(stride 784). */
int gcpx_wgrad_image4x4s2(const float* da, const float* add, const float* r, const float* image, int32_t F, int32_t S, float* partial,
                          int32_t grid, void* stream);

/* ---- loss gradients ---- */
/* d NLL / d params of the discretised logistic mixture (same layouts as gcpx_dlm_nll); row gradient scaled by
   row_weight[row] * scale; rows with weight 0 are zero-filled.  colsum (optional): [rows][pitch] per-row sums over the
   pixels of dparams (the output head's bias gradient, summed over rows by gcpx_colsum).  nll_out (optional): [rows] the same
   values gcpx_dlm_nll writes — the training step evaluates loss and gradient in ONE pass over the 2.3 GB of parameters */
int gcpx_dlm_nll_bwd(const float* params, const float* target, const float* row_weight, float scale, float* dparams, float* colsum,
                     float* nll_out, int32_t rows, int32_t npix, int32_t pitch, int32_t n_mix, void* stream);
/* d total / d logits of the length CE, existence BCE and state L2 heads (same arguments as gcpx_loss_combine);
   dlen [B][ceil16(T)] (pad columns zero), dexist [B*N][16] (column 0), dstate [B*T][16] (columns < state_dim); NULL outputs are skipped */
int gcpx_loss_heads_bwd(const gcpx_loss_args* a, float* dlen, float* dexist, float* dstate, void* stream);
/* same for the inverse-model and cost-model L2 heads: daction [B][16] (columns < n_actions), dcost [B][16] (column 0) */
int gcpx_loss_aux_heads_bwd(const gcpx_loss_args* a, float* daction, float* dcost, void* stream);

/* ---- parameters ---- */
/* dst[i] = theta[idx0[i]] (+ theta[idx1[i]]), negative index = 0: every fragment-packed weight arena is a gather of the
   canonical flat parameter vector (video-gcp_amd/packing.py), refreshed once per optimizer step */
int gcpx_repack(const float* theta, const int32_t* idx0, const int32_t* idx1, float* dst, int64_t n, void* stream);
/* the same, held to max_blocks 256-thread workgroups (0 = no limit): see gcpx_optim_range */
int gcpx_repack_blocks(const float* theta, const int32_t* idx0, const int32_t* idx1, float* dst, int64_t n, int32_t max_blocks, void* stream);
/* split-f16 weights from the flat parameter vector (one launch per tensor, after gcpx_repack): v[i] = theta[idx[i]] (negative = 0),
   e = 14 - floor(log2 max|v|) (0 when all zero), *log2_out = e, and the two f16 pieces of v[i] 2^e in the layout the split kernels
   read: out[((i / 512) * 2 + p) * 512 + i % 512], p = 0 (rn16(v 2^e)) and 1 (rn16 of the remainder); n % 512 == 0.  Same pieces as
   packing.split_f16 on the host */
int gcpx_split_pack(const float* theta, const int32_t* idx, int32_t n, void* out, int32_t* log2_out, void* stream);
/* The same for several tensors in one launch (one workgroup per tensor; csrc/split_pack.hip): tab = DEVICE array of descriptors */
typedef struct gcpx_split_pack_desc {
    const float* src;        /* flat parameter vector (or the folded weights of a row-folded block) */
    const int32_t* idx;      /* [n] gather indices into src, -1 = zero */
    void* out;               /* [n / 512][2][512] f16 */
    int32_t* log2_out;       /* exponent of the tensor */
    int32_t n, _pad;
} gcpx_split_pack_desc;
int gcpx_split_pack_group(const gcpx_split_pack_desc* tab, int32_t nprob, void* stream);
/* The same result, bit for bit, from two launches with 32 workgroups per tensor (largest magnitudes first, then the pieces): the form
   the trainer runs behind every optimizer step.  scratch: DEVICE [nprob] uint32, overwritten */
int gcpx_split_pack_group2(const gcpx_split_pack_desc* tab, int32_t nprob, uint32_t* scratch, void* stream);
/* Row-folded weights of an upsampling decoder block (bilinear x2, align_corners=False, then 3x3 conv, pad 1 — DecoderModule's
   pyramid / additional_conv_layer blocks, blox; called through gcp/prediction/models/tree/tree_dense_rec.py:42).  Output row
   2 y + py of the block reads the three low-resolution rows y - 1, y, y + 1 (clamped) of the horizontally interpolated input with
   weights that are fixed linear combinations of the conv's three tap rows:
       py = 0:  [0.75 W0 + 0.25 W1,  0.25 W0 + 0.75 W1 + 0.75 W2,  0.25 W2]
       py = 1:  [0.25 W0,  0.75 W0 + 0.75 W1 + 0.25 W2,  0.25 W1 + 0.75 W2]
   and the conv's zero padding above row 0 / below the last row is restored by the correction sets -W0 (on the clamped row above)
   and -W2 (on the clamped row below).  w: dev [Cout][Cin][3][3]; out: dev f32 [24][Cout][Cin], set t = py 9 + dyl 3 + tx for
   t < 18, 18 + tx = -W0[tx], 21 + tx = -W2[tx]; sums formed in float64 in tap-row order (bit-identical to packing.fold_up_weights). */
int gcpx_fold_upsample_weights(const float* w, int32_t Cout, int32_t Cin, float* out, void* stream);
/* RAdam (Liu et al. 2019; blox.torch.radam.RAdam as used by gcp_builder.py:178-179): state[0] = step counter (float),
   incremented by this call; rectified update when the variance is tractable (rho_t > 5), momentum SGD otherwise */
int gcpx_radam_step(float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, float* state, int64_t n, float lr,
                    float beta1, float beta2, float eps, float grad_scale, void* stream);
/* The trainer's other optimizers (gcp_builder.py:174-186; torch.optim.Adam / RMSprop / SGD formulas): kind 1 = adam (p1, p2 = betas),
   2 = rmsprop (p1 = momentum, p2 = alpha), 3 = sgd (p1 = momentum).  m / v: the optimizer's two state vectors; state[0] = step
   counter (incremented), state[1] = gradient-clipping coefficient applied to the (scaled) gradient (0 = unset = 1; set per step by
   gcpx_grad_clip_coef; gcpx_radam_step applies it too). */
int gcpx_optim_step(float* theta, const float* grad, float* m, float* v, float* state, int64_t n, int32_t kind, float lr, float p1,
                    float p2, float eps, float grad_scale, void* stream);
/* One slice of an optimizer step: kind 0 = RAdam (p1, p2 = betas), 1 .. 3 as gcpx_optim_step, over n elements of the four vectors;
   state[0] is incremented only when tick != 0.  A trainer whose gradient becomes final slice by slice (the tree levels of the backward
   pass, leaves first) applies each slice while the rest is still being differentiated and ticks with the last one: every slice of a
   step must read the same counter, so all tick == 0 calls of a step are ordered before its tick != 0 call.  The same arithmetic per
   element as gcpx_radam_step / gcpx_optim_step (a step cut into slices gives the same bits as one call).  max_blocks > 0 holds the
   launch to that many 256-thread workgroups: a slice applied beside the backward's critical chain must leave it the chip (a full
   grid's wavefronts fill every CU and stretched the chain's 40 us GEMMs to 100 us). */
int gcpx_optim_range(float* theta, const float* grad, float* m, float* v, float* state, int64_t n, int32_t kind, float lr, float p1,
                     float p2, float eps, float grad_scale, int32_t tick, int32_t max_blocks, void* stream);
/* gradient_clip (gcp_builder.py:186 -> blox get_clipped_optimizer; spec here: torch.nn.utils.clip_grad_norm_ over all parameters):
   state[2] = || grad_scale * grad ||_2, state[1] = min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0).  partial: [n_partial]
   scratch (deterministic two-stage sum). */
int gcpx_grad_clip_coef(const float* grad, int64_t n, float grad_scale, float max_norm, float* partial, int32_t n_partial, float* state,
                        void* stream);

/* ---------------------------------------------------------------------------------------------------
 * hipGraph helpers: capture a launch sequence once, replay it per step.
 * ------------------------------------------------------------------------------------------------- */
int gcpx_graph_begin(void* stream);
int gcpx_graph_end(void* stream, void** graph_exec);
int gcpx_graph_launch(void* graph_exec, void* stream);
int gcpx_graph_destroy(void* graph_exec);

/* side streams for the independent branches of the forward (the I_0 / I_g encoder passes, the per-level
   split_linear merge and prior next to the posterior chain, the aux heads next to the decoder).  Under
   gcpx_graph_begin/_end a gcpx_event_record on the capturing stream followed by gcpx_stream_wait_event on a side
   stream forks the capture; the reverse joins it, so the branches become parallel paths of the hipGraph. */
int gcpx_stream_create(void** stream);
/* level < 0: highest priority of the device, 0: middle, > 0: lowest (side lanes that must not delay the critical chain) */
int gcpx_stream_create_priority(void** stream, int level);
int gcpx_stream_destroy(void* stream);
int gcpx_stream_wait_event(void* stream, void* ev);

/* event timing on the caller's stream (bench.py measures the dominant kernel with these) */
int gcpx_event_create(void** ev);
int gcpx_event_create_sync(void** ev);   /* ordering only: no timing, device-scope release (lane fork / join) */
int gcpx_event_record(void* ev, void* stream);
int gcpx_event_elapsed_ms(void* start, void* stop, float* ms);
int gcpx_event_destroy(void* ev);

/* ===================================================================================================
 * Adaptive (soft-DTW) frame binding + attentive inference — the long-horizon configuration
 *   (experiments/prediction/base_configs/gcp_adaptive.py:6-11: matching_type='dtw_image', attentive_inference=True).
 * ================================================================================================= */

/* Masked single-query multi-head attention of the attentive posterior
 *   (blox MultiheadAttention as called at adaptive_binding/attentive_inference.py:81; q / k / v already projected):
 *   row r = (b, j), b = r / rpb:  score[h][t] = q[r][h] . k[b][t][h] / sqrt(dk / heads) / temperature[0],
 *   frames outside [start_ind[b], end_ind[b]] masked out, softmax over t, out[r][h] = sum_t a[h][t] v[b][t][h].
 *   q [M][dk], k [B][T][dk], v [B][T][nz], out [M][nz] dense; att [M][T] = head-averaged weights or NULL;
 *   start_ind NULL = 0. */
int gcpx_attention(const float* q, const float* k, const float* v, const int64_t* start_ind, const int64_t* end_ind,
                   const float* temperature, float* out, float* att, int32_t M, int32_t rpb, int32_t T, int32_t dk, int32_t nz,
                   int32_t heads, void* stream);

/* batch_cdist(x, y, reduction='sum') of adaptive.py:44 / binding_loss.py:24 (blox.torch.ops.batch_cdist): squared L2 distance
 *   out[b][n][t] = max(0, |x[b][n]|^2 + |y[b][t]|^2 - 2 x[b][n] . y[b][t]),  x [B][N][K], y [B][T][K], K % 32 == 0.
 *   f32 MFMA GEMM with gcpx_cdist_splits(K) deterministic split-K partials: partial [splits][B][N][T], xnorm [B*N], ynorm [B*T]
 *   are caller-owned scratch. */
int gcpx_cdist_splits(int64_t K);
int gcpx_cdist(const float* x, const float* y, int32_t B, int32_t N, int32_t T, int64_t K, float* partial, float* xnorm,
               float* ynorm, float* out, void* stream);

/* soft_dtw(cost / temp, end_ind) + normalize(w, 1) of adaptive.py:50-58 (probabilistic_dtw.py:11-122), float64 inside:
 *   cost[b][n][t] = (dsum[b][n][t] / D) / temp[0]  ('mean' reduction of the cdist, then the matching temperature);
 *   w[b][n][t] = P(node n is aligned with frame t) under the Gibbs distribution over monotone alignments without horizontal
 *   moves that start at (0, 0) and end at (N-1, end_ind[b]), divided by max(sum over nodes, 1e-7).  Depth-first node order.
 *   acc: caller-owned scratch, float64 [2*B][N][T] (forward and backward accumulators).  Needs N >= T, T <= 1024. */
int gcpx_soft_dtw(const float* dsum, float D, const float* temp, const int64_t* end_ind, int32_t B, int32_t N, int32_t T,
                  double* acc, float* w, void* stream);

/* d loss / d temp of a LEARNED matching temperature (hyperparameters.py:132 learn_matching_temp = True; adaptive.py:19-21 keeps
 * `temp` trainable, adaptive.py:51 divides the detached cost by it inside the autograd graph): the averaging criterion
 * (binding_loss.py:24-35)  coef * sum_{b,n,t} w[b][n][t] * pad[b][t] * (0.5 dsum exp(-2 ls) + D (ls + 0.5 log 2 pi))  differentiated
 * through w = normalize(soft_dtw(cost / temp)) in forward mode along the alignment lattice, float64 inside.
 *   acc: the [2*B][N][T] accumulators gcpx_soft_dtw left for the same dsum / temp / end_ind;
 *   tangent [2*B][N][T] and partial [B]: caller-owned float64 scratch;  dtemp[0] += the gradient. */
int gcpx_soft_dtw_dtemp(const float* dsum, float D, const float* temp, const int64_t* end_ind, const double* acc,
                        const float* pad_mask, const float* log_sigma, float coef, int32_t B, int32_t N, int32_t T,
                        double* tangent, double* partial, float* dtemp, void* stream);

/* Bookkeeping on the matching distribution w [B][N = 2^L - 1][T] (depth-first):
 *   frame2node [B][T]  depth-first position of argmax over nodes, first maximum in BREADTH-first order
 *                      (tree.bf.match_dist.argmax(1), frame_binding.py:30; all-zero column -> root)
 *   matched_idx [B][T] the same for t <= end_ind[b], -1 beyond (get_matched_pruned_seqs, base_gcp.py:358-366), or NULL
 *   best_t [B][N]      argmax over frames (adaptive.py:119); entropy [B][N] = -sum p log p; p_n [B][N] = clamp(sum p, 0, 1)
 *                      (tree_module.py:145-147) */
int gcpx_match_stats(const float* w, const int64_t* end_ind, int32_t B, int32_t L, int32_t T, int32_t* frame2node,
                     int32_t* matched_idx, int32_t* best_t, float* entropy, float* p_n, void* stream);

/* AdaptiveBinding.prune_sequence (adaptive.py:62-77): node p > 0 is dropped when sigmoid(dist[b][p-1]) > threshold.
 *   leave [B][N]; kept_idx [B][N] = depth-first positions of kept nodes then -1; count [B]; target [B][N-1] (or NULL) =
 *   best_t[p] == best_t[p-1], the BCE target of adaptive.py:118-122. */
int gcpx_distance_prune(const float* dist, float threshold, const int32_t* best_t, int32_t B, int32_t N, int32_t* leave,
                        int32_t* kept_idx, int32_t* count, int32_t* target, void* stream);

/* LossAveragingCriterion.loss (binding_loss.py:19-42) before pad_mask / batch mean:
 *   nll_bt[b][t] = sum_n w[b][n][t] * (0.5 * dsum[b][n][t] * exp(-log_sigma)^2 + D * (log_sigma + 0.5 log 2 pi)) */
int gcpx_averaging_nll(const float* dsum, const float* w, const float* log_sigma, float D, int32_t B, int32_t N, int32_t T,
                       float* nll_bt, void* stream);

/* LossAveragingCriterion.get_soft_estimates (binding_loss.py:44-58): out[b][t][:] = sum_n w[b][n][t] * x[b][n][:], rows of D floats */
int gcpx_soft_average(const float* w, const float* x, float* out, int32_t B, int32_t N, int32_t T, int64_t D, void* stream);

/* Evaluation alignment: basic_dtw + _traceback of gcp/evaluation/dtw_utils.py:77-95,201-218 (the C version is
 *   gcp/evaluation/cutils.pyx:22-29) as used by DTWEvalBinding.get_single_matches (gcp/evaluation/evaluation_matching.py:133-146).
 *   cost [B][N][T] float32 (mean squared distance estimate n vs target t); sequence b uses rows < n_len[b] and columns < t_len[b]
 *   (NULL = N / T).  acc [B][N][T] float64 accumulated cost; inds [B][T] = for every target frame the estimate with the smallest
 *   accumulated cost among the path cells of its column (-1 beyond t_len); path [B][2][N+T] = (rows | columns) from the END of
 *   the path to its start, path_len [B]; dist [B] = acc[n-1][t-1] / (n + t). */
int gcpx_dtw_align(const float* cost, const int32_t* n_len, const int32_t* t_len, int32_t B, int32_t N, int32_t T, double* acc,
                   int32_t* inds, int32_t* path, int32_t* path_len, double* dist, void* stream);

/* ---- backward of the adaptive path (training step of config c5) ---- */
/* d images of LossAveragingCriterion.loss (binding_loss.py:19-42), the matching weights being constants (adaptive.py:50 detaches):
 *   dimg[b][n][:] = coef * exp(-2 log_sigma) * (images[b][n][:] * sum_t wp - sum_t wp[b][n][t] * traj[b][t][:]),  wp = w * pad_mask;
 *   dlog_sigma[0] += coef * sum w * pad * (D - dsum * exp(-2 log_sigma)).  coef = d total / d (per-sequence loss value). */
int gcpx_averaging_nll_bwd(const float* w, const float* pad_mask, const float* images, const float* traj, const float* dsum,
                           const float* log_sigma, float coef, int32_t B, int32_t N, int32_t T, int64_t D, float* dimg,
                           float* dlog_sigma, void* stream);
/* backward of the mixture mean (`images` of the discrete-logistic-mixture head): params / dparams [rows][npix][pitch] in the head's
 *   slot order, dimg NCHW [rows][3][npix]; colsum [rows][pitch] (per-frame column sums for the bias gradient) or NULL.
 *   used_slots = pitch: every slot of dparams is written (zeros where the mean does not depend on the parameter: the log-scales);
 *   used_slots = 8 * n_mix: only slots 0 .. 8 n_mix - 1 are read and written (the green / blue log-scales and the padding behind them
 *   carry no gradient) — for a head backward that walks the leading 8 n_mix channels only (gcpx_conv3x3 with Cin < src[0].C,
 *   gcpx_wgrad_conv3x3_split_src with Cout = 8 n_mix); slots >= used_slots of dparams are left untouched, of colsum written as 0 */
int gcpx_dlm_mean_bwd(const float* params, const float* dimg, float* dparams, float* colsum, int32_t rows, int32_t npix, int32_t pitch,
                      int32_t n_mix, int32_t used_slots, void* stream);
/* backward of gcpx_attention (one head): from d_out [M][nz] and the saved weights att [M][T]:
 *   dS [M][T] (scratch, gradient w.r.t. the scaled scores), dq [M][dk], dtemp_row [M] (per-row terms of d temperature),
 *   dK [B][T] rows of dk floats with leading dimension ldk, dV [B][T] rows of nz floats with leading dimension ldv
 *   (gradients of the projected keys / values summed over the rows of a sequence). */
int gcpx_attention_bwd(const float* q, const float* k, const float* v, const float* att, const float* d_out, const int64_t* end_ind,
                       const float* temperature, float* dS, float* dq, float* dtemp_row, float* dK, int64_t ldk, float* dV, int64_t ldv,
                       int32_t M, int32_t rpb, int32_t T, int32_t dk, int32_t nz, void* stream);

/* ---- collectives of the data-parallel paths: RCCL over xGMI, one communicator per process (one process
 *   per GPU), enqueued on the caller's stream, no synchronisation.  Replaces nn.DataParallel's gradient reduce
 *   (/root/reference/gcp/prediction/training/gcp_builder.py:71-78) and the per-GPU split of planning work (gcp/planning/run.py:108-120:
 *   here the costs of one sharded CEM population are gathered).  The Python host reaches the same RCCL through torch.distributed
 *   (video-gcp_amd/dist.py); these entry points serve a binder without torch.  RCCL is resolved at run time (GCPX_ERR_COMM when absent). */
#define GCPX_COMM_ID_BYTES 128
/* rank 0: a fresh rendezvous id (GCPX_COMM_ID_BYTES bytes) to hand to every rank by any host channel */
int gcpx_comm_unique_id(void* id_out);
/* every rank (its GPU current): *comm = communicator of `world` ranks */
int gcpx_comm_init(void** comm, int32_t rank, int32_t world, const void* id);
/* in-place sum of buf[0..n) over the ranks (the flat fp32 gradient or one of its buckets; 1 / world is folded into gcpx_optim_step) */
int gcpx_comm_allreduce(void* comm, float* buf, int64_t n, void* stream);
/* recv[r * n .. (r + 1) * n) = rank r's send[0..n)  (the candidates' costs of a sharded CEM population) */
int gcpx_comm_allgather(void* comm, const float* send, float* recv, int64_t n, void* stream);
int gcpx_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* GCPX_H */
