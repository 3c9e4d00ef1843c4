"""ctypes binding of libgcpx.so (the C-ABI declared in include/gcpx.h).

The product path has NO fallback: if the HIP library is missing or a symbol is absent, importing callers fail
loudly.  torch is used only to own device memory and to provide the current HIP stream.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgcpx.so")

ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
HEAD_RAW, HEAD_DLM_MEAN, HEAD_DLM_BOTH, HEAD_TANH_NCHW, HEAD_DLM_NLL, HEAD_DLM_NLL_GRAD = 0, 1, 2, 3, 4, 5
SPLIT_PLAIN, SPLIT_ROWFOLD, SPLIT_HEAD32, SPLIT_ROWFOLD16 = 0, 1, 2, 3
EPI_NONE, EPI_LRELU, EPI_LSTM, EPI_GAUSS_SAMPLE = 0, 1, 2, 3
MLP_PLAIN, MLP_GAUSS, MLP_TANH = 0, 1, 2

vp = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64


class ConvSrc(C.Structure):
    _fields_ = [("ptr", vp), ("scale", vp), ("shift", vp), ("C", i32), ("frame_div", i32), ("act", i32), ("_pad", i32)]


class ConvArgs(C.Structure):
    _fields_ = [("src", ConvSrc * 2), ("nsrc", i32), ("F", i32), ("Hin", i32), ("Win", i32), ("Hout", i32),
                ("Wout", i32), ("Cin", i32), ("Cout", i32), ("out_pitch", i32), ("upsample", i32), ("out_act", i32),
                ("head_mode", i32), ("wpk", vp), ("bias", vp), ("out", vp), ("images", vp), ("stats_partial", vp),
                ("raw_row_map", vp), ("src_row_map", vp), ("src_row_frames", vp), ("n_src_rows", i32), ("w_split_log2", i32), ("wpk_split", vp), ("w_split_log2_dev", vp),
                ("split_layout", i32), ("nll_rows", i32), ("nll_target", vp), ("nll_partial", vp), ("nll_row_weight", vp),
                ("nll_scale", C.c_float), ("_pad3", i32), ("images_rows", vp), ("images_rows_dup", i64), ("bwd_r", vp), ("bwd_scale", vp), ("bwd_shift", vp), ("bwd_mean", vp),
                ("bwd_rstd", vp), ("addend", vp), ("addend_frame_div", i32), ("_pad4", i32)]


class LossArgs(C.Structure):
    _fields_ = [("nll_bt", vp), ("pad_mask", vp), ("kl_b", vp), ("len_logits", vp), ("end_ind", vp), ("existence", vp),
                ("leave", vp), ("regressed_state", vp), ("state_target", vp), ("seq_len", vp), ("out", vp),
                ("B", i32), ("T", i32), ("N", i32), ("state_dim", i32), ("w_rec", C.c_float), ("w_kl", C.c_float),
                ("w_len", C.c_float), ("w_exist", C.c_float), ("w_state", C.c_float), ("total_div", C.c_float),
                ("action_pred", vp), ("action_seq", vp), ("inv_t0", vp), ("cost_pred", vp), ("cost_target", vp),
                ("n_actions", i32), ("w_action", C.c_float), ("w_cost", C.c_float), ("state_mask", vp), ("w_kl_dev", vp)]


class RowSrc(C.Structure):
    _fields_ = [("ptr", vp), ("rowidx", vp), ("scale", vp), ("shiftv", vp), ("sb", i64), ("sr", i64),
                ("width", i32), ("shift", i32), ("act", i32), ("cmod", i32)]


class GemmArgs(C.Structure):
    _fields_ = [("src", RowSrc * 6), ("nsrc", i32), ("M", i32), ("N", i32), ("K", i32), ("rpb", i32),
                ("wpk", vp), ("bias", vp), ("out", vp), ("ob", i64), ("orow", i64), ("epi", i32), ("nbatch", i32),
                ("stats_partial", vp), ("c_prev", vp), ("c_prev_stride", i64), ("h_out", vp), ("c_out", vp),
                ("hb", i64), ("hrow", i64), ("h_copy", vp), ("z_src_off", i64), ("z_w_off", i64),
                ("z_bias_off", i64), ("z_out_off", i64), ("gates_out", vp),
                ("wpk_split", vp), ("w_split_log2_dev", vp), ("w_split_log2", i32), ("_pad_split", i32),
                ("x_planes", vp), ("x_exp", vp), ("x_planes_bytes", i64), ("lstm_bwd", vp)]


class MlpArgs(C.Structure):
    _fields_ = [("src", RowSrc * 6), ("nsrc", i32), ("M", i32), ("rpb", i32), ("in_dim", i32), ("mid", i32),
                ("n_mid", i32), ("out_dim", i32), ("w_in", vp), ("b_in", vp), ("w_mid", vp), ("b_mid", vp),
                ("gn_gamma", vp), ("gn_beta", vp), ("w_out", vp), ("b_out", vp), ("gn_eps", C.c_float),
                ("lrelu_slope", C.c_float), ("epi", i32), ("out_split", i32), ("out", vp), ("ob", i64), ("orow", i64),
                ("oblk", i64), ("eps", vp), ("eb", i64), ("erow", i64), ("z", vp), ("zb", i64), ("zrow", i64), ("save", vp)]


class MlpBwdDx(C.Structure):
    _fields_ = [("wT", vp), ("out", vp), ("ob", i64), ("orow", i64), ("width", i32), ("_pad", i32)]


class MlpBwdArgs(C.Structure):
    _fields_ = [("dout", vp), ("save", vp), ("wT_out", vp), ("wT_mid", vp * 4), ("gn_gamma", vp * 4), ("gn_beta", vp * 4),
                ("du", vp * 5), ("gn_partial", vp * 4), ("dx", MlpBwdDx * 4), ("ldo", i64), ("M", i32), ("rpb", i32),
                ("mid", i32), ("n_mid", i32), ("out_pad", i32), ("ndx", i32), ("gn_eps", C.c_float), ("lrelu_slope", C.c_float)]


class SplitPackDesc(C.Structure):
    _fields_ = [("src", vp), ("idx", vp), ("out", vp), ("log2_out", vp), ("n", i32), ("_pad", i32)]


class WgradArgs(C.Structure):
    _fields_ = [("dy", vp), ("x", vp), ("rowidx", vp), ("frame_map", vp), ("scale", vp), ("shiftv", vp), ("out", vp),
                ("ldy", i64), ("sb", i64), ("sr", i64), ("ldw", i64), ("dy_sb", i64), ("R", i32), ("N", i32), ("n_valid", i32),
                ("K", i32), ("mode", i32), ("Cin", i32), ("H", i32), ("W", i32), ("rpb", i32), ("shift", i32),
                ("act", i32), ("cmod", i32), ("k_off", i32), ("accumulate", i32), ("partial", i32), ("nsplit", i32),
                ("dy_rpb", i32), ("nbatch", i32), ("dbias", vp), ("dbias2", vp), ("z_dy_off", i64), ("z_x_off", i64),
                ("z_out_off", i64), ("z_bias_off", i64), ("split_f16", i32), ("_pad_split", i32)]


class LstmBwdArgs(C.Structure):
    _fields_ = [("gates", vp), ("c_prev", vp), ("c_new", vp), ("dh_dense", vp), ("dh_pos", vp), ("dc_pos", vp),
                ("dgates", vp), ("dc_prev", vp), ("c_prev_stride", i64), ("pb", i64), ("prow", i64), ("dh_stride", i64),
                ("dcp_stride", i64), ("M", i32), ("H", i32), ("rpb", i32), ("_pad", i32)]


class TreeAccumSrc(C.Structure):
    _fields_ = [("ptr", vp), ("ld", i64), ("off_left", i32), ("off_right", i32), ("off_ctx0", i32), ("off_ctxg", i32),
                ("dst_col", i32), ("_pad", i32)]


class TreeAccumArgs(C.Structure):
    _fields_ = [("src", TreeAccumSrc * 6), ("dst", vp), ("dst_sb", i64), ("slot_stride", i64), ("nsrc", i32), ("B", i32),
                ("n", i32), ("width", i32)]


class ActBwdArgs(C.Structure):
    _fields_ = [("da", vp), ("add", vp), ("r", vp), ("scale", vp), ("shift", vp), ("mean", vp), ("rstd", vp), ("dy", vp),
                ("stats_partial", vp), ("ldc", i64), ("c_off", i32), ("up", i32), ("fsum", i32), ("act", i32), ("F", i32),
                ("H", i32), ("W", i32), ("C", i32)]


WG_ROWS, WG_CONV1D, WG_CONV3X3, WG_CONV4X4S2 = 0, 1, 2, 3
WMAP_LINEAR, WMAP_CONV, WMAP_CONVT = 0, 1, 2

# every symbol include/gcpx.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("gcpx_version", C.c_int, []),
    ("gcpx_last_error", C.c_char_p, []),
    ("gcpx_conv_grid", C.c_int, []),
    ("gcpx_conv3x3", C.c_int, [C.POINTER(ConvArgs), vp]),
    ("gcpx_conv3x3_grid", C.c_int, [C.POINTER(ConvArgs)]),
    ("gcpx_conv4x4s2", C.c_int, [C.POINTER(ConvArgs), vp]),
    ("gcpx_conv4x4s2_grid", C.c_int, []),
    ("gcpx_conv4x4s2_image", C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    ("gcpx_bn_finalize", C.c_int, [vp, i32, i32, i32, C.c_double, vp, vp, C.c_float, vp, vp, vp, vp, C.c_float, vp, vp, vp]),
    ("gcpx_bn_fold", C.c_int, [vp, vp, vp, vp, C.c_float, i32, vp, vp, vp]),
    ("gcpx_randn", C.c_int, [vp, i64, vp, vp]),
    ("gcpx_gemm", C.c_int, [C.POINTER(GemmArgs), vp]),
    ("gcpx_gemm_row_blocks", C.c_int, [i32, i32]),
    ("gcpx_gemm_planes_workspace", C.c_int, [i32, i32, i32, vp, vp]),
    ("gcpx_mlp", C.c_int, [C.POINTER(MlpArgs), vp]),
    ("gcpx_balanced_binding", C.c_int, [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    ("gcpx_dlm_nll", C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    ("gcpx_gauss_nll", C.c_int, [vp, vp, vp, vp, i32, i32, vp]),
    ("gcpx_kl_gauss", C.c_int, [vp, vp, i32, i32, i32, i64, i64, C.c_float, vp, i64, vp, vp]),
    ("gcpx_gauss_sample", C.c_int, [vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, i32, i32, vp]),
    ("gcpx_seq_index", C.c_int, [vp, i32, i32, vp, vp, vp]),
    ("gcpx_fill_zero", C.c_int, [vp, i64, vp]),
    ("gcpx_copy_rows", C.c_int, [vp, vp, i32, i32, i64, i64, i64, vp]),
    ("gcpx_loss_combine", C.c_int, [C.POINTER(LossArgs), vp]),
    ("gcpx_loss_pre", C.c_int, [C.POINTER(LossArgs), vp, vp, i32, i32, i64, i64, C.c_float, vp, i64, vp, vp]),
    ("gcpx_loss_final", C.c_int, [C.POINTER(LossArgs), vp]),
    ("gcpx_gather_rows", C.c_int, [vp, vp, vp, i32, i32, i32, i32, i64, vp]),
    ("gcpx_gather_rows_rest", C.c_int, [vp, vp, vp, i32, i32, i32, i32, i64, vp, vp]),
    ("gcpx_compact_index", C.c_int, [vp, i32, i32, i32, vp, vp]),
    ("gcpx_seq_pairs", C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
    ("gcpx_rollout_cost", C.c_int, [vp, i64, vp, vp, i64, vp, i32, i32, i32, i32, i32, C.c_float, vp]),
    ("gcpx_masked_row_sum", C.c_int, [vp, vp, vp, i32, i32, vp]),
    ("gcpx_wgrad", C.c_int, [C.POINTER(WgradArgs), vp]),
    ("gcpx_wgrad_classify", C.c_int, [C.POINTER(WgradArgs), i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("gcpx_wgrad_group", C.c_int, [vp, vp, i32, i32, i32, vp]),
    ("gcpx_wgrad_conv3x3", C.c_int, [vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    ("gcpx_wgrad_conv3x3_split", C.c_int, [vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    ("gcpx_wgrad_conv3x3_split_up", C.c_int, [vp, i32, C.POINTER(ConvArgs), i32, vp, i32, vp]),
    ("gcpx_wgrad_conv3x3_split_src", C.c_int, [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp]),
    ("gcpx_wgrad_image4x4s2", C.c_int, [vp, vp, vp, vp, i32, i32, vp, i32, vp]),
    ("gcpx_split_pack_group", C.c_int, [vp, i32, vp]),
    ("gcpx_split_pack_group2", C.c_int, [vp, i32, vp, vp]),
    ("gcpx_wgrad_reduce", C.c_int, [vp, i32, i32, i32, vp, i32, i32, i32, i32, vp, i64, i32, i32, vp]),
    ("gcpx_colsum", C.c_int, [vp, i64, i32, i32, i32, i64, i32, vp, vp, vp, i32, vp]),
    ("gcpx_reduce_partials", C.c_int, [vp, i32, i64, i32, vp, i32, vp]),
    ("gcpx_lstm_bwd", C.c_int, [C.POINTER(LstmBwdArgs), vp]),
    ("gcpx_mlp_bwd", C.c_int, [C.POINTER(MlpBwdArgs), vp]),
    ("gcpx_mlp_bwd_group", C.c_int, [C.POINTER(MlpBwdArgs), i32, vp]),
    ("gcpx_mlp_bwd_blocks", C.c_int, [i32]),
    ("gcpx_gn_lrelu_bwd", C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, C.c_float, C.c_float, vp]),
    ("gcpx_gn_bwd_blocks", C.c_int, [i32]),
    ("gcpx_tanh_bwd_rows", C.c_int, [vp, vp, vp, i64, i64, i32, i32, i32, vp]),
    ("gcpx_lrelu_bwd", C.c_int, [vp, vp, vp, i64, C.c_float, vp]),
    ("gcpx_kl_bwd", C.c_int, [vp, vp, vp, vp, i32, i32, i32, i64, i64, C.c_float, C.c_float, vp]),
    ("gcpx_kl_bwd_weighted", C.c_int, [vp, vp, vp, vp, i32, i32, i32, i64, i64, C.c_float, C.c_float, vp, i64, vp]),
    ("gcpx_kl_bwd_scheduled", C.c_int, [vp, vp, vp, vp, i32, i32, i32, i64, i64, C.c_float, C.c_float, vp, i64, vp, vp]),
    ("gcpx_rows_strided", C.c_int, [vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, vp]),
    ("gcpx_latent_bwd", C.c_int, [vp, vp, vp, i64, i64, vp, i64, i64, vp, i64, vp, i64, vp, vp, i32, i32, i32, vp]),
    ("gcpx_tree_accum", C.c_int, [C.POINTER(TreeAccumArgs), vp]),
    ("gcpx_timestep_scatter", C.c_int, [vp, i64, i64, vp, vp, i32, i32, i32, i32, vp]),
    ("gcpx_add_rows", C.c_int, [vp, i64, i64, vp, vp, i32, i32, i32, vp]),
    ("gcpx_zero_unmapped_rows", C.c_int, [vp, i64, vp, i32, vp]),
    ("gcpx_index_offset", C.c_int, [vp, vp, i32, i32, i32, vp]),
    ("gcpx_index_inverse", C.c_int, [vp, i32, vp, i32, vp]),
    ("gcpx_act_bwd", C.c_int, [C.POINTER(ActBwdArgs), vp]),
    ("gcpx_act_bwd_blocks", C.c_int, []),
    ("gcpx_act_skip_bwd", C.c_int, [C.POINTER(ActBwdArgs), vp, i32, i32, i32, vp]),
    ("gcpx_bn_bwd_finalize", C.c_int, [vp, i32, i32, C.c_double, vp, vp, vp, vp, vp, i32, vp]),
    ("gcpx_bn_bwd_apply", C.c_int, [vp, vp, vp, vp, vp, i64, i32, vp]),
    ("gcpx_conv_stage", C.c_int, [C.POINTER(ConvArgs), vp]),
    ("gcpx_col2im4x4s2", C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    ("gcpx_im2col_image", C.c_int, [vp, vp, i32, i32, i32, vp]),
    ("gcpx_dlm_nll_bwd", C.c_int, [vp, vp, vp, C.c_float, vp, vp, vp, i32, i32, i32, i32, vp]),
    ("gcpx_loss_heads_bwd", C.c_int, [C.POINTER(LossArgs), vp, vp, vp, vp]),
    ("gcpx_image_metrics", C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    ("gcpx_gemm_group_dims", C.c_int, [vp, i32, vp, vp]),
    ("gcpx_gemm_group", C.c_int, [vp, vp, i32, i32, vp]),
    ("gcpx_mlp_group_dims", C.c_int, [vp, i32, vp, vp]),
    ("gcpx_mlp_group", C.c_int, [vp, vp, i32, i32, i32, vp]),
    ("gcpx_mlp_group_gemm_supported", C.c_int, [C.POINTER(GemmArgs), i32, i32]),
    ("gcpx_mlp_group_gemm", C.c_int, [vp, vp, i32, i32, i32, C.POINTER(GemmArgs), vp]),
    ("gcpx_loss_aux_heads_bwd", C.c_int, [C.POINTER(LossArgs), vp, vp, vp]),
    ("gcpx_aux_sample_indices", C.c_int, [vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    ("gcpx_aux_sample_indices_gauss", C.c_int, [vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    ("gcpx_aux_index_rows", C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp, vp]),
    ("gcpx_path_cost", C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    ("gcpx_sample_length", C.c_int, [vp, vp, i32, i32, i32, vp, vp]),
    ("gcpx_repack", C.c_int, [vp, vp, vp, vp, i64, vp]),
    ("gcpx_split_pack", C.c_int, [vp, vp, i32, vp, vp, vp]),
    ("gcpx_fold_upsample_weights", C.c_int, [vp, i32, i32, vp, vp]),
    ("gcpx_optim_step", C.c_int, [vp, vp, vp, vp, vp, i64, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
    ("gcpx_grad_clip_coef", C.c_int, [vp, i64, C.c_float, C.c_float, vp, i32, vp, vp]),
    ("gcpx_optim_range", C.c_int, [vp, vp, vp, vp, vp, i64, i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32, i32, vp]),
    ("gcpx_repack_blocks", C.c_int, [vp, vp, vp, vp, i64, i32, vp]),
    ("gcpx_radam_step", C.c_int, [vp, vp, vp, vp, vp, i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
    ("gcpx_attention", C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    ("gcpx_cdist_splits", C.c_int, [i64]),
    ("gcpx_cdist", C.c_int, [vp, vp, i32, i32, i32, i64, vp, vp, vp, vp, vp]),
    ("gcpx_soft_dtw", C.c_int, [vp, C.c_float, vp, vp, i32, i32, i32, vp, vp, vp]),
    ("gcpx_soft_dtw_dtemp", C.c_int, [vp, C.c_float, vp, vp, vp, vp, vp, C.c_float, i32, i32, i32, vp, vp, vp, vp]),
    ("gcpx_match_stats", C.c_int, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    ("gcpx_distance_prune", C.c_int, [vp, C.c_float, vp, i32, i32, vp, vp, vp, vp, vp]),
    ("gcpx_averaging_nll", C.c_int, [vp, vp, vp, C.c_float, i32, i32, i32, vp, vp]),
    ("gcpx_soft_average", C.c_int, [vp, vp, vp, i32, i32, i32, i64, vp]),
    ("gcpx_averaging_nll_bwd", C.c_int, [vp, vp, vp, vp, vp, vp, C.c_float, i32, i32, i32, i64, vp, vp, vp]),
    ("gcpx_dlm_mean_bwd", C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    ("gcpx_attention_bwd", C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, i32, vp]),
    ("gcpx_dtw_align", C.c_int, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    ("gcpx_graph_begin", C.c_int, [vp]),
    ("gcpx_graph_end", C.c_int, [vp, C.POINTER(vp)]),
    ("gcpx_graph_launch", C.c_int, [vp, vp]),
    ("gcpx_graph_destroy", C.c_int, [vp]),
    ("gcpx_stream_create", C.c_int, [C.POINTER(vp)]),
    ("gcpx_stream_create_priority", C.c_int, [C.POINTER(vp), C.c_int]),
    ("gcpx_stream_destroy", C.c_int, [vp]),
    ("gcpx_stream_wait_event", C.c_int, [vp, vp]),
    ("gcpx_event_create", C.c_int, [C.POINTER(vp)]),
    ("gcpx_event_create_sync", C.c_int, [C.POINTER(vp)]),
    ("gcpx_event_record", C.c_int, [vp, vp]),
    ("gcpx_event_elapsed_ms", C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    ("gcpx_event_destroy", C.c_int, [vp]),
    ("gcpx_comm_unique_id", C.c_int, [vp]),
    ("gcpx_comm_init", C.c_int, [C.POINTER(vp), i32, i32, vp]),
    ("gcpx_comm_allreduce", C.c_int, [vp, vp, i64, vp]),
    ("gcpx_comm_allgather", C.c_int, [vp, vp, vp, i64, vp]),
    ("gcpx_comm_destroy", C.c_int, [vp]),
]

_lib = None


class GcpxError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen libgcpx.so and bind every symbol; raises if the library or any symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("GCPX_LIB") or LIB_PATH     # GCPX_LIB: another build of the same library (same-box A/B of kernels)
    if not os.path.exists(p):
        raise GcpxError(f"HIP extension not built: {p} is missing (run `python -c 'import __graft_entry__ as g; g.build()'`"
                        " or video-gcp_amd/csrc/build.sh). There is no CPU fallback.")
    lib = C.CDLL(p)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if path is None:
        _lib = lib
    return lib


def lib():
    return load_library()


def check(status, what=""):
    if status != 0:
        msg = lib().gcpx_last_error()
        raise GcpxError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
