"""Host side of the gcp_tree hot path: mirrors the reference's model API
(/root/reference/gcp/prediction/models/base_gcp.py:29-304, models/tree/tree.py:14-77) and drives the HIP kernels of
libgcpx.so through ctypes.  torch owns device memory and streams only — every FLOP of the forward runs in
csrc/*.hip; there is no eager/PyTorch fallback.

Data layout in HBM (fp32):
  * conv activations NHWC; conv outputs are stored RAW (pre-BatchNorm) and the consumer applies the folded
    per-channel affine + LeakyReLU while it stages its tile ("normalise on load");
  * the subgoal tree lives in "position layout": a sequence of P = 2^L + 1 slots per batch element — slot 0 is the
    start frame latent e_0, slot 2^L the goal latent e_g, and tree node (level l, index j) sits at slot
    (2j+1) * 2^(L-1-l), i.e. at its depth-first (= temporal) position + 1.  The parents of a node are the slots
    +-2^(L-1-l) away, so the reference's interleave / bf<->df shuffles (tree_utils.py:37-44, 79-108, 202-232)
    become strides in the kernels' row maps and no data is ever moved between levels;
  * decoded images are [B, N, 3, H, W] in depth-first node order (the reference's `tree.df.images`).
"""
import ctypes as C
import os
from contextlib import contextmanager

import torch

from . import packing as pk
from . import runtime as rt
from .hparams import GCPHParams
from .params import init_params, encoder_layers, encoder_skip_layers, decoder_layers
from .plan_ops import _Plan, Outputs, ModelOutputs, PlanOpsMixin, _addr, N_LANES      # noqa: F401 (re-exported)
from .weights import WeightsMixin
from .forward_plan import ForwardPlanMixin
from .replay import ReplayMixin


class GCPTreeModel(WeightsMixin, PlanOpsMixin, ForwardPlanMixin, ReplayMixin):
    """TreeModel(params, logger) counterpart.  `model(inputs, phase)` -> Outputs."""

    _has_aux_training = True      # sampled inverse-model / cost-model training pairs (base_gcp.py:249-260)
    _has_pred_length = True       # val_mode(pred_length=True) draws the sequence length (base_gcp.py:219-226)
    _rng_in_plan = True           # the latent noise / index draws of a forward without fed noise are an op of the plan (gcpx_randn)

    _n_models = 0
    _stream_pool = {}           # (device index, main-stream priority) -> (torch stream, [lane streams]): shared by all models of the process

    def __init__(self, hp: GCPHParams, params=None, device="cuda", seed=0, materialize_distr=False):
        self._hp = hp
        self.device = torch.device(device)
        self.lib = rt.load_library()          # raises if the HIP extension is missing
        self._check_hp(hp)
        self._flatten_params(params or self._default_params(hp, seed))
        self.training = True                  # BatchNorm uses batch statistics (reference trains and validates so)
        self._decode = True
        self._sample_prior = False            # ProbabilisticModel._sample_prior (switched by val_mode)
        self._use_pred_length = False
        self.materialize_distr = materialize_distr
        self._bufs = {}
        self._plans = {}
        # state other parts of the class create later, declared here (weights.py, replay.py, training.GCPTrainStep)
        self.pk_split = {}                    # split-f16 twins of the conv weights: name -> (pieces, log2 scale)   (weights._pack_split)
        self._split_tab = None                #   ... and the device table their re-split reads
        self._gsplit, self._gsplit_tabs = {}, {}   # split-f16 twins of the row GEMMs' weights: pack address -> (pieces, log2 scales); re-split tables
        self._gsplit_live = False             # the wide tree levels' GEMMs stay on the split kernels in training (backward_weights._live_gemm_split)
        self._merge_side_rows = 1 << 60       # rows from which a level's merge GEMM runs on a side lane / on the planes GEMM (weights._pack_gemm_split)
        self._planes_min_rows = 1 << 60
        self._arena = None                    # one arena for every packed weight (weights.build_arena: the trainer's re-pack)
        self._plan_listeners = []             # called when the plans are dropped (the trainer's backward plans hang off them)
        self._rng_seed = None                 # torch CUDA seed the in-plan Philox stream was started from (replay._sync_rng_state)
        self._rng_stream_id = GCPTreeModel._n_models      # distinct noise streams for the models of one process
        GCPTreeModel._n_models += 1
        self.n_steps = 0                      # BaseGCPModel.step(): KL weight burn-in
        self._kl_w = None
        # how a built plan is replayed: True = hipGraph, False = eager launches over the three lanes, "auto" (default) = whichever
        # replays faster back to back, timed once per plan (GCPX_FORWARD_REPLAY=graph|eager|auto).  At c2 the graph loses 0.19 ms of
        # 2.9 to the eager plan (a ~30 us gap in front of every replay plus its cross-branch edges; the host needs ~0.4 ms to enqueue a
        # 2.8 ms forward); at c1 (a 0.3 ms forward) the eager plan is host-bound and the graph wins
        self.use_graph = {"graph": True, "eager": False}.get(os.environ.get("GCPX_FORWARD_REPLAY", "auto"), "auto")
        # hipGraph capture is not allowed on the legacy default stream: the model launches on its own stream
        # and orders it against the caller's current stream with events (wait_stream), never a host sync
        # Every model of a process on one device shares ONE set of lanes: the runtime deals streams onto its four hardware queues in
        # creation order, and a second model's lanes land on other queues than the first one's — its training step then ran 1.5-2.3 ms
        # slower (13.6 / 15.1 / 13.6 / 15.9 ms for four trainers built in a row, tools/ab_train_inproc.py): lanes that share a queue with
        # each other or with the caller's stream serialise.  Models of one process are not run concurrently (their launches would be
        # ordered by the shared streams); GCPX_PRIVATE_STREAMS=1 gives every model its own.
        pool_key = (self.device.index, os.environ.get("GCPX_MAIN_PRIORITY", "0"))
        pool = None if os.environ.get("GCPX_PRIVATE_STREAMS") else GCPTreeModel._stream_pool.get(pool_key)
        if pool is None:
            main = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("GCPX_MAIN_PRIORITY", "0")))
            lanes = [main.cuda_stream]
            for _ in range(N_LANES - 1):
                sp = C.c_void_p()
                with torch.cuda.device(self.device):
                    rt.check(self.lib.gcpx_stream_create(C.byref(sp)), "stream_create")
                lanes.append(sp)
            pool = (main, lanes)
            if not os.environ.get("GCPX_PRIVATE_STREAMS"):
                GCPTreeModel._stream_pool[pool_key] = pool
        self._stream, self._streams = pool[0], list(pool[1])
        self._set_kl_weight()
        self.save_for_backward = False        # training step: forward plans keep what the backward pass needs
        # split-f16 convs (csrc/split_mfma.h, conv3x3_split.hip, conv3x3_head_split.hip): f32-equivalent results on the f16 matrix pipes.  GCPX_EXACT_F32=1 keeps every
        # conv on the exact f32 MFMA kernels
        self.split_f16 = os.environ.get("GCPX_EXACT_F32") is None
        # forward with losses: likelihood of the matched frames inside the head kernel (GCPX_HEAD_DLM_NLL); GCPX_UNFUSED_NLL=1 keeps
        # the stored-parameters + gcpx_dlm_nll path (what the exact-f32 build and the training forward run)
        self.fused_head_nll = os.environ.get("GCPX_UNFUSED_NLL") is None
        # GCPX_HEAD32=1: the mixture head on 32x32x16 MFMA tiles (csrc/conv3x3_head32.hip) for the modes that store no raw parameters.
        # Parity-green and measured slower than the 16x16x32 kernel (profiles/r06_head32_study.txt): off unless asked for
        self.head32 = os.environ.get("GCPX_HEAD32", "0") == "1"
        self._timed_op = None                 # name of one plan op bracketed by HIP events (bench.py roofline)
        self._timed_events = []
        self._pack_all()
        # sub-module handles under the reference's attribute names (planner_policy.py:225-227, tree_dense_rec.py:13-40)
        from . import handles as Hd
        self.encoder, self.decoder, self.dense_rec = Hd.EncoderHandle(self), Hd.DecoderHandle(self), Hd.DenseRecHandle(self)
        self.tree_module = Hd.TreeModuleHandle(self)
        if hp.attach_inv_mdl:
            self.inv_mdl = Hd.InverseModelHandle(self)
        if hp.attach_cost_mdl:
            self.cost_mdl = Hd.CostModelHandle(self)

    def _check_hp(self, hp):
        assert hp.matching_type in ("balanced", "dtw_image")
        if hp.tree_lstm not in ("split_linear", "linear", "sum", ""):              # tree_lstm.py:52-60; '' = the non-LSTM subgoal
            raise ValueError("don't know this TreeLSTM type")                      # predictor (tree_module.py:45-46,109-110)
        if hp.seq_enc not in ("conv", "none"):
            # base_gcp.py:130-138: 'lstm' / 'bi-lstm' are blox's RecurrentSeqEncodingModule / BidirectionalSeqEncodingModule (absent)
            raise ValueError(f"seq_enc = {hp.seq_enc!r}: 'conv' (ConvSeqEncodingModule) and 'none' (Identity) are built")
        if hp.lstm_init not in ("mlp", "zero"):
            raise ValueError("dont know lstm init type {}!".format(hp.lstm_init))  # tree_lstm.py:74
        if hp.attentive_inference:
            assert hp.n_attention_layers == 1, "one attention layer is built (hyperparameters.py:25 default)"
        assert not (hp.action_conditioned_pred or hp.deterministic or hp.non_goal_conditioned), \
            "the vmpc.py variants belong to the flat predictor (SequentialModel): GCPSequentialModel"
        if hp.adaptive:
            assert hp.top_bias == 1.0 and hp.leaves_bias == 0.0, "WeightsHacker biases are not built (defaults only)"
            assert hp.entropy_weight == 0.0, "the matching entropy is reported, not optimised (hyperparameters.py default)"

    def _n_latents(self):
        return self._hp.n_nodes

    def _zero_row(self, n):
        return self._buf("zero_row", (n,), zero=True)

    def _head_nll_fusable(self):
        hp = self._hp
        return (self.fused_head_nll and self.split_f16 and "dec.head" in self.pk_split and hp.img_sz % 16 == 0 and
                hp.decoder_distribution == "discrete_logistic_mixture" and not self.materialize_distr)

    def _rows_direct(self, key):
        """the output head stores the frames of the balanced tree that belong to a row of the sequence there itself
        (gcpx_conv_args.images_rows): split-f16 mixture head, decoded frames wanted"""
        hp = self._hp
        return bool(key[8] and not hp.adaptive and self.split_f16 and "dec.head" in self.pk_split and hp.img_sz % 16 == 0 and
                    hp.decoder_distribution == "discrete_logistic_mixture" and not self.materialize_distr and
                    os.environ.get("GCPX_HEAD_NO_ROWS") is None and type(self)._build_plan is GCPTreeModel._build_plan)

    def _head_grad_fused(self, key):
        """training forward (posterior path with losses, backward to follow) of the balanced model: the head kernel writes the
        likelihood gradient itself"""
        has_traj, sample_prior, phase, with_loss = key[1], key[3], key[4], key[7]
        return bool(self.save_for_backward and with_loss and has_traj and not sample_prior and phase == "train" and
                    not self._hp.adaptive and self._head_nll_fusable() and type(self)._build_plan is GCPTreeModel._build_plan)

    def _default_params(self, hp, seed):
        return init_params(hp, seed)

    # ------------------------------------------------------------------------------------------------
    # reference API surface
    # ------------------------------------------------------------------------------------------------
    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    @contextmanager
    def val_mode(self, pred_length=True, decode=True):
        """base_gcp.py:44-53: sample from the prior instead of the posterior.  decode=False (planner scoring): the latent tree,
        pruning and the latent-space heads are computed but the image decoder is skipped — the learned planning cost only reads
        latents (cost_fcn.py:84-97), images are decoded for the plan that is returned."""
        self._sample_prior, self._use_pred_length, self._decode = True, pred_length, decode
        try:
            yield
        finally:
            self._sample_prior, self._use_pred_length, self._decode = False, False, True

    def state_dict(self):
        return dict(self.sd)

    def step(self):
        """BaseModel.step (base_model.py:24-25, called once per optimisation step, train.py:163): advances the `Updater` children.
        The only one the reference builds is the KL-weight burn-in (base_gcp.py:121-128, kl_weight_burn_in=None by default):
        `LinearUpdater(kl_weight, n_iter, target)` is blox (absent) — this build's spec: the weight starts at 0 (the reference
        initialises the parameter with zeros) and is target * min(1, steps / n_iter) after `steps` calls.  The current value lives in a
        device scalar that the loss kernels and the KL backward read, so captured graphs need no rebuild."""
        self.n_steps = self.n_steps + 1
        self._set_kl_weight()

    def _set_kl_weight(self):
        hp = self._hp
        if hp.kl_weight_burn_in:
            w = hp.kl_weight * min(1.0, self.n_steps / float(hp.kl_weight_burn_in))
            if self._kl_w is None:
                self._kl_w = torch.zeros(1, device=self.device)
            self._kl_w.fill_(w)
            self.kl_weight_now = w
        else:
            self._kl_w, self.kl_weight_now = None, hp.kl_weight

    def load_state_dict(self, sd, strict=True):
        for k, v in sd.items():
            if k in self.sd:
                self.sd[k].copy_(v)
            elif strict:
                raise KeyError(k)
        self._pack_all()
        self._clear_plans()

    def _clear_plans(self):
        self._plans.clear()
        for cb in self._plan_listeners:
            cb()

    def __call__(self, inputs, phase="train", noise=None):
        return self.forward(inputs, phase, noise)

    def input_buffer(self, name, shape, dtype=None):
        """The persistent device buffer the launch plans read input `name` from.  A data loader that writes its batch straight into
        these buffers (and passes them as the inputs) hands the batch over without the per-call staging copy — 63 MB for traj_seq at
        c2; any other tensor is copied in as before."""
        if dtype is None:
            dtype = torch.int64 if name in ("end_ind", "inv_t0", "inv_t1", "cost_start_idx", "cost_end_idx") else torch.float32
        return self._buf("in." + name, tuple(shape), dtype)

    # ------------------------------------------------------------------------------------------------
    # buffers and plan-building helpers
    # ------------------------------------------------------------------------------------------------
    _buf_prefix = ""          # handle plans (model.encoder / model.decoder) keep their activations apart from the forward's

    def _wrap_outputs(self, o, tin, phase):
        hp = self._hp
        out = ModelOutputs()
        out["_model"] = self
        out.end_ind = tin["end_ind"]
        out.raw = o
        out.images_df = o.get("images_df")            # None after val_mode(decode=False)
        if "seq_len_logits" in o:
            out.seq_len_logits = o["seq_len_logits"]
        if "existence" in o:
            out.existence_predictor = Outputs(existence=o["existence"])
        if "distances" in o:
            out.distance_predictor = Outputs(distances=o["distances"])           # adaptive.py:69
        if "soft_matched_estimates" in o:
            out.soft_matched_estimates = o["soft_matched_estimates"]
        if "match_dist_df" in o:
            out.entropy = o["entropy_df"].index_select(1, TreeView.bf2df_index(hp.hierarchy_levels, self.device))
        out.tree = TreeView(self, o)
        out.dense_rec = Outputs()
        out._lazy = (o, tin)
        return out

    def _plans_rec(self, o):
        return [v[1].rec for v in self._plans.values() if v[1].outs is o][0]

    # ---- eager helpers used by the planner (cost model / inverse model on arbitrary rows) ----
    def predictor_rows(self, name, *inputs):
        """Run one packed Predictor (self.pk[name]) on dense row tensors [R, C_i] -> [R, out_dim]."""
        W = self.pk[name]
        R = inputs[0].shape[0]
        xs = [x.to(device=self.device, dtype=torch.float32).contiguous() for x in inputs]
        out = torch.empty(R, W["out_dim"], device=self.device)
        plan = _Plan(self.lib)
        srcs = [self._rowsrc(x.data_ptr(), 0, x.shape[1], x.shape[1]) for x in xs]
        self._mlp(plan, name, W, srcs, R, R, out=out.data_ptr(), ob=0, orow=W["out_dim"])
        caller = torch.cuda.current_stream(self.device)
        plan.run([caller.cuda_stream] * N_LANES)
        del xs
        return out

    def encode(self, images):
        """encoder(img)[0][:, :, 0, 0] (planner_policy.py:225): NCHW images in [-1, 1] -> latents [F, nz_enc]."""
        return self._encode(images)[0]

    def _encode(self, images, keep_skips=False):
        from .handles import Skips
        x = torch.as_tensor(images).to(device=self.device, dtype=torch.float32).contiguous()
        Fr = x.shape[0]
        out = torch.empty(Fr, self._hp.nz_enc, device=self.device)
        plan = _Plan(self.lib)
        old, self._buf_prefix = self._buf_prefix, f"handle.enc{Fr}."
        try:
            skips = self._plan_encoder(plan, "x", x.data_ptr(), Fr, out.data_ptr(), self._hp.nz_enc, 0, 1)
        finally:
            self._buf_prefix = old
        caller = torch.cuda.current_stream(self.device)
        plan.run([caller.cuda_stream] * N_LANES)
        caller.synchronize()      # the plan's argument structs (and x) must outlive the launches
        if keep_skips:
            # private copies: the encoder's activation buffers are reused by the next call of the same size
            skips = {k: (t.clone(), C_, None if sc is None else sc.clone(), None if sh is None else sh.clone(), act)
                     for k, (t, C_, sc, sh, act) in skips.items()}
        return out, (Skips(skips, Fr) if keep_skips else None)

    def _decode_seq(self, inputs, enc):
        """DecoderModule.decode_seq (tree_dense_rec.py:42): enc [B, N, nz_enc(,1,1)] -> Outputs(images [B, N, 3, H, W])"""
        from .handles import Skips, _rows
        hp = self._hp
        enc = torch.as_tensor(enc).to(device=self.device, dtype=torch.float32)
        while enc.dim() > 3:
            enc = enc[..., 0]
        B, N = enc.shape[:2]
        enc = enc.contiguous()
        sk = inputs.get("skips") if isinstance(inputs, dict) else None
        if not isinstance(sk, Skips):
            _, sk = self._encode(inputs["I_0"], keep_skips=True)
        assert sk.n_frames == B, "one set of skip activations per sequence (base_gcp.py:190: only the start image's)"
        S = hp.img_sz
        images = torch.empty(B, N, hp.input_nc, S, S, device=self.device)
        plan = _Plan(self.lib)
        old, self._buf_prefix = self._buf_prefix, f"handle.dec{B}x{N}."
        train_was = self.training
        try:
            prev = self._plan_decoder_features(plan, self._rowsrc(enc.data_ptr(), N * hp.nz_enc, hp.nz_enc, hp.nz_enc), B * N, N, sk.srcs)
            dlm = hp.decoder_distribution == "discrete_logistic_mixture"
            a = self._conv_args([prev], B * N, S, S, S, S, hp.head_channels, self._head_pitch, self.pk["dec.head.w"], self.pk["dec.head.b"],
                                None, upsample=0, head_mode=(rt.HEAD_DLM_MEAN if dlm else rt.HEAD_TANH_NCHW), images=images)
            self._set_split(a, "dec.head")
            plan.keep.append(a)
            plan.add("dec.head", self.lib.gcpx_conv3x3, C.byref(a))
        finally:
            self._buf_prefix = old
        caller = torch.cuda.current_stream(self.device)
        plan.run([caller.cuda_stream] * N_LANES)
        caller.synchronize()
        return Outputs(images=images)

    # ---- losses: computed inside the forward graph when traj_seq + pad_mask are fed in phase 'train' ----
    LOSS_NAMES = ("dense_img_rec", "kl", "len_pred", "existence_predictor", "state_regression")

    def loss(self, inputs, outputs, log_error_arr=False):
        """BaseGCPModel.loss + TreeModule.loss (base_gcp.py:264-292, tree_module.py:116-157): {name: (value, weight)}."""
        raw = outputs.raw
        if "losses" not in raw:
            raise ValueError("losses need traj_seq and pad_mask in the inputs of a phase='train' forward")
        hp, lv = self._hp, raw["losses"]
        w = dict(dense_img_rec=hp.dense_img_rec_weight, kl=self.kl_weight_now, len_pred=hp.length_pred_weight,
                 existence_predictor=1.0, state_regression=1.0)
        res = Outputs()
        for i, name in enumerate(self.LOSS_NAMES):
            if name == "len_pred" and not hp.regress_length:
                continue
            if name == "existence_predictor" and hp.adaptive:
                name = "distance_predictor"                  # adaptive.py:118-122 takes the slot of the existence BCE
                w[name] = 1.0
            if name == "state_regression" and ("regressed_state_padded" not in raw or "traj_seq_states" not in inputs):
                continue
            res[name] = Outputs(value=lv[i], weight=w[name])
        if hp.adaptive:                                             # tree_module.py:128 (entropy_weight = 0: logged only)
            res["entropy"] = Outputs(value=raw["entropy_sum"][0] / raw["entropy_df"].numel(), weight=hp.entropy_weight)
        if raw.get("actions_sampled") is not None and "actions" in inputs:      # base_gcp.py:275-276, inverse_mdl.py:181-191
            res["action_reconst"] = Outputs(value=lv[7], weight=hp.action_rec_weight)
        if raw.get("cost_pred") is not None:                        # base_gcp.py:279-280, cost_mdl.py:59-62
            res["cost_estimation"] = Outputs(value=lv[8], weight=1.0)
        res["nll"] = Outputs(value=lv[6], weight=0.0)               # base_gcp.py:289-290
        res["_total"] = lv[5]
        return res

    def get_total_loss(self, inputs, losses):
        """base_gcp.py:294-304: sum of weight * value over weights > 0, divided by prod(traj_seq.shape[1:])."""
        return Outputs(value=losses["_total"])

    # ---- ragged views: these synchronise (they read seq_len on the host), keep them out of timed regions ----
    def pruned_prediction(self, out):
        """outputs.pruned_prediction: list of [len_b, 3, H, W] (tree.py:62-65)."""
        o = out.raw
        lens = o["pruned_len" if "pruned_len" in o else "seq_len"].tolist()
        return [o["pruned_padded"][b, :lens[b]] for b in range(len(lens))]

    def soft_matched_estimates(self, out):
        """LossAveragingCriterion.get_soft_estimates (binding_loss.py:44-58): per-frame average of the node images under
        the matching distribution (visualisation only, so computed on demand — one launch on the caller's stream)."""
        o = out.raw
        w, x = o["match_dist_df"], o["images_df"]
        B, N, T = w.shape
        res = torch.empty((B, T) + tuple(x.shape[2:]), device=self.device)
        rt.check(self.lib.gcpx_soft_average(w.data_ptr(), x.data_ptr(), res.data_ptr(), B, N, T, x[0, 0].numel(),
                                            torch.cuda.current_stream(self.device).cuda_stream), "soft_average")
        return res

    def aux_outputs(self, out):
        o = out.raw
        m = int(o["aux_len" if "aux_len" in o else "seq_len"].max().item())
        res = Outputs(model_enc_seq=o["model_enc_seq_padded"][:, :m])
        if "regressed_state_padded" in o:
            res.regressed_state = o["regressed_state_padded"][:, :m]
        if "actions_padded" in o:
            res.actions = o["actions_padded"][:, :m - 1]
        elif "actions_sampled" in o:                                # one sampled frame pair per sequence (inverse_mdl.py:136-178)
            res.actions = o["actions_sampled"]
        if "cost_pred" in o:
            res.cost, res.cost_target = o["cost_pred"], o["cost_target"][:, None]
        return res


class TreeView:
    """`outputs.tree` with `.bf` / `.df` accessors (tree_utils.py:165-199).  Depth-first order is the native
    layout; breadth-first views are index_selects."""

    def __init__(self, model, o):
        self._m, self._o = model, o
        hp = model._hp
        L = hp.hierarchy_levels
        self.depth = L
        self._bf2df = TreeView.bf2df_index(L, model.device)

    _BF2DF = {}

    @staticmethod
    def bf2df_index(L, device):
        """depth-first position of every breadth-first node index.  Cached per (L, device): building it per forward meant a
        pageable host-to-device copy behind the forward on the caller's stream, i.e. a host sync in every model call."""
        key = (L, str(device))
        t = TreeView._BF2DF.get(key)
        if t is None:
            idx = torch.empty(2 ** L - 1, dtype=torch.long)
            for l in range(L):
                for j in range(2 ** l):
                    idx[2 ** l - 1 + j] = (2 * j + 1) * 2 ** (L - 1 - l) - 1
            t = TreeView._BF2DF[key] = idx.to(device)
        return t

    def _df(self, name):
        o, hp = self._o, self._m._hp
        nv = hp.nz_vae
        N = hp.n_nodes
        if name == "images":
            return o["images_df"]
        if name == "e_g_prime":
            return o["E"][:, 1:1 + N]
        if name == "hidden_state":
            return o["Hid"][:, 1:1 + N]
        if name == "z":
            return o["Z"][:, 1:1 + N]
        if name in ("p_z_mu", "p_z_log_sigma", "q_z_mu", "q_z_log_sigma"):
            t = o["PZ" if name.startswith("p_") else "QZ"][:, 1:1 + N]
            return t[..., :nv] if name.endswith("mu") else t[..., nv:]
        if name == "match_timesteps":
            return o["node_t"]
        if name == "match_dist":                      # adaptive binding: tree.bf.match_dist = depthfirst2breadthfirst(w) (adaptive.py:60)
            return o["match_dist_df"]
        if name == "p_n":
            return o["p_n_df"]
        if name in ("gamma", "e_tilde"):              # attentive posterior (attentive_inference.py:31); stored per level
            rec = self._m._plans_rec(o)[name]
            B = o["E"].shape[0]
            bf = torch.cat([rec[l].view(B, 2 ** l, -1) for l in range(self.depth)], 1)
            inv = torch.empty_like(self._bf2df)
            inv[self._bf2df] = torch.arange(len(inv), device=inv.device)
            return bf.index_select(1, inv)
        if name == "distr":
            d = o["distr_df_kernel_order"]
            if d is None:
                raise KeyError("distr not materialised: build the model with materialize_distr=True")
            perm = self._m._dlm_perm
            inv = torch.empty(hp.head_channels, dtype=torch.long, device=d.device)
            slots = torch.nonzero(perm >= 0)[:, 0]
            inv[perm[slots]] = slots
            return d.index_select(-1, inv).permute(0, 1, 4, 2, 3)
        raise KeyError(name)

    class _Acc:
        def __init__(self, tree, bf):
            self._t, self._bf = tree, bf

        def __getattr__(self, name):
            v = self._t._df(name)
            return v.index_select(1, self._t._bf2df) if self._bf else v

        __getitem__ = __getattr__

    @property
    def df(self):
        return TreeView._Acc(self, False)

    @property
    def bf(self):
        return TreeView._Acc(self, True)
