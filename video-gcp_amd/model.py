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


def _addr(t, off_elems=0):
    return t.data_ptr() + 4 * off_elems


N_LANES = 3


class _Plan:
    """A recorded launch sequence (C entry point + argument struct) over up to N_LANES streams: lane 0 is the
    model's main stream, lanes 1.. are side streams for independent branches.  `fork`/`join` order the lanes with
    events; under hipGraph capture they become parallel paths of the graph.  Replayed eagerly or as a graph."""

    def __init__(self, lib):
        self.lib = lib
        self.ops = []        # (name, fn, args, lane) | ("@fork"/"@join", None, (lanes, events), 0)
        self.keep = []       # keeps argument structs / tensors alive
        self.graph = None
        self.eager = False   # replay by eager launches although a graph exists (GCPTreeModel._eager_replays_faster)
        self.lane = 0
        self.rec = {}        # buffers / records the backward plan is built from (training step)
        self.deferred = []   # ops waiting to be issued on a side lane (training.py)

    def add(self, name, fn, *args):
        self.ops.append((name, fn, args, self.lane))

    def _events(self, n):
        evs = []
        for _ in range(n):
            e = C.c_void_p()
            rt.check(self.lib.gcpx_event_create_sync(C.byref(e)), "event_create")
            evs.append(e)
        return evs

    def fork(self, lanes):
        self.ops.append(("@fork", None, (tuple(lanes), self._events(1)), 0))

    def join(self, lanes):
        self.ops.append(("@join", None, (tuple(lanes), self._events(len(lanes))), 0))

    def wait(self, waiter, signaler):
        """lane `waiter` continues only after everything issued so far on lane `signaler` (one directional edge: a chain running
        ahead on a side lane hands over chunk by chunk instead of being joined at every step)"""
        self.ops.append(("@wait", None, (waiter, signaler, self._events(1)[0]), 0))

    def mark(self, tag, payload):
        """a host-side callback point in the launch sequence (eager replay only): `run(..., on_mark=f)` calls f(tag, payload)"""
        self.ops.append(("@mark", None, (tag, payload), 0))

    def run(self, streams, ops=None, on_mark=None):
        lib = self.lib
        for name, fn, args, lane in (self.ops if ops is None else ops):
            if name == "@mark":
                if on_mark is not None:
                    on_mark(*args)
            elif name == "@fork":
                lanes, evs = args
                rt.check(lib.gcpx_event_record(evs[0], streams[0]), "fork")
                for l in lanes:
                    rt.check(lib.gcpx_stream_wait_event(streams[l], evs[0]), "fork")
            elif name == "@wait":
                waiter, signaler, ev = args
                rt.check(lib.gcpx_event_record(ev, streams[signaler]), "wait")
                rt.check(lib.gcpx_stream_wait_event(streams[waiter], ev), "wait")
            elif name == "@join":
                lanes, evs = args
                for l, e in zip(lanes, evs):
                    rt.check(lib.gcpx_event_record(e, streams[l]), "join")
                    rt.check(lib.gcpx_stream_wait_event(streams[0], e), "join")
            else:
                st = fn(*args, streams[lane])
                if st != 0:
                    rt.check(st, name)


class Outputs(dict):
    """AttrDict-like container (the reference returns blox.AttrDict)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class ModelOutputs(Outputs):
    """what `model(inputs)` returns.  The ragged views the reference's callers read as plain attributes — `pruned_prediction`
    (tree.py:62-65), `actions`, `regressed_state`, `model_enc_seq` (base_gcp.py:234-262) — need the sequence lengths on the host,
    so they are built on first access (one device-to-host copy of B integers) instead of inside every forward."""

    _LAZY = ("pruned_prediction", "actions", "regressed_state", "model_enc_seq", "cost", "cost_target")

    def __getattr__(self, name):
        if name in self:
            return self[name]
        if name in ModelOutputs._LAZY and "_model" in self:
            m = self["_model"]
            if name == "pruned_prediction":
                self[name] = m.pruned_prediction(self)
            else:
                aux = m.aux_outputs(self)
                for k in aux:
                    self.setdefault(k, aux[k])
            if name in self:
                return self[name]
        raise AttributeError(name)


class GCPTreeModel:
    """TreeModel(params, logger) counterpart.  `model(inputs, phase)` -> Outputs."""

    _has_aux_training = True      # sampled inverse-model / cost-model training pairs (base_gcp.py:249-260)
    _has_pred_length = True       # val_mode(pred_length=True) draws the sequence length (base_gcp.py:219-226)
    _rng_in_plan = True           # the latent noise / index draws of a forward without fed noise are an op of the plan (gcpx_randn)

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
        # split-f16 convs (csrc/conv3x3_split.hip): f32-equivalent results on the f16 matrix pipes.  GCPX_EXACT_F32=1 keeps every
        # conv on the exact f32 MFMA kernels
        self.split_f16 = os.environ.get("GCPX_EXACT_F32") is None
        # forward with losses: likelihood of the matched frames inside the head kernel (GCPX_HEAD_DLM_NLL); GCPX_UNFUSED_NLL=1 keeps
        # the stored-parameters + gcpx_dlm_nll path (what the exact-f32 build and the training forward run)
        self.fused_head_nll = os.environ.get("GCPX_UNFUSED_NLL") is None
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

    def _flatten_params(self, params):
        """All parameters live in ONE flat fp32 vector `theta` (canonical torch layouts, 16-byte aligned segments);
        `self.sd` holds views.  The optimizer, the gradient all-reduce and the re-pack gather work on the flat vector."""
        off, self._poff = 0, {}
        for k, v in params.items():
            self._poff[k] = (off, tuple(v.shape))
            off += (v.numel() + 3) // 4 * 4
        self.theta = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.sd = {}
        for k, v in params.items():
            o, shp = self._poff[k]
            view = self.theta[o:o + v.numel()].view(shp)
            view.copy_(v)
            self.sd[k] = view

    def _check_hp(self, hp):
        assert hp.matching_type in ("balanced", "dtw_image")
        if hp.tree_lstm not in ("split_linear", "linear", "sum", ""):              # tree_lstm.py:52-60; '' = the non-LSTM subgoal
            raise ValueError("don't know this TreeLSTM type")                      # predictor (tree_module.py:45-46,109-110)
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
        return (self.fused_head_nll and self.split_f16 and "dec.head" in getattr(self, "pk_split", {}) and hp.img_sz % 16 == 0 and
                hp.decoder_distribution == "discrete_logistic_mixture" and not self.materialize_distr)

    def _rows_direct(self, key):
        """the output head stores the frames of the balanced tree that belong to a row of the sequence there itself
        (gcpx_conv_args.images_rows): split-f16 mixture head, decoded frames wanted"""
        hp = self._hp
        return bool(key[8] and not hp.adaptive and self.split_f16 and "dec.head" in getattr(self, "pk_split", {}) and hp.img_sz % 16 == 0 and
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
        self.n_steps = getattr(self, "n_steps", 0) + 1
        self._set_kl_weight()

    def _set_kl_weight(self):
        hp = self._hp
        if hp.kl_weight_burn_in:
            w = hp.kl_weight * min(1.0, getattr(self, "n_steps", 0) / float(hp.kl_weight_burn_in))
            if getattr(self, "_kl_w", None) is None:
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
        for cb in getattr(self, "_plan_listeners", ()):
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
    # weight packing
    # ------------------------------------------------------------------------------------------------
    def _pack_predictor(self, prefix, out_dim):
        sd = self._psd
        mid = sd[f"{prefix}.input.linear.weight"].shape[0]
        n_mid = 0
        while f"{prefix}.pyramid-{n_mid}.linear.weight" in sd:
            n_mid += 1
        out_pad = (out_dim + 15) // 16 * 16
        w_in = sd[f"{prefix}.input.linear.weight"]
        k_raw = w_in.shape[1]
        if k_raw % 16:
            # an input narrower than one MFMA k-group (the action encoder's n_actions columns, sequential.py:108-110): zero columns up
            # to 16 — the caller feeds rows padded the same way; in_dim_raw is the parameter's own width (its gradient's row pitch)
            w_in = torch.cat([w_in, torch.zeros((mid, -k_raw % 16), dtype=w_in.dtype, device=w_in.device)], 1)
        d = dict(mid=mid, n_mid=n_mid, out_dim=out_dim, in_dim=w_in.shape[1], in_dim_raw=k_raw)
        d["w_in"] = pk.pack_gemm(w_in)
        d["b_in"] = sd[f"{prefix}.input.linear.bias"].contiguous()
        if n_mid:
            d["w_mid"] = torch.stack([pk.pack_gemm(sd[f"{prefix}.pyramid-{i}.linear.weight"]) for i in range(n_mid)]).contiguous()
            d["b_mid"] = torch.stack([sd[f"{prefix}.pyramid-{i}.linear.bias"] for i in range(n_mid)]).contiguous()
            d["gn_g"] = torch.stack([sd[f"{prefix}.pyramid-{i}.norm.weight"] for i in range(n_mid)]).contiguous()
            d["gn_b"] = torch.stack([sd[f"{prefix}.pyramid-{i}.norm.bias"] for i in range(n_mid)]).contiguous()
        d["w_out"] = pk.pack_gemm(sd[f"{prefix}.head.linear.weight"])
        d["b_out"] = pk.pad_vec(sd[f"{prefix}.head.linear.bias"], out_pad)
        return d

    def _pack_all(self):
        """(Re)build every fragment-packed weight from the canonical parameters.  Once a parameter arena exists
        (training: `build_arena`), re-packing is ONE gather launch over the flat parameter vector."""
        if getattr(self, "_arena", None) is not None:
            self.repack()
            return
        self.pk = self._pack_tree(self.sd)
        if self._hp.tree_lstm:
            self._pack_fused_embed()
        self._pack_split()
        self._pack_gemm_split()

    def _pack_split(self):
        """The two f16 pieces of the conv weights that have a split-f16 kernel (csrc/conv3x3_split.hip).  They are gathered and split
        on the device from the flat parameter vector (gcpx_split_pack: one small launch per tensor), at weight load and — in
        training — after every optimizer step, right behind the fragment re-pack.  self.pk_split[name] = dict(idx, out, log2)."""
        self.pk_split = {}
        hp = self._hp
        todo = []
        if hp.decoder_distribution == "discrete_logistic_mixture":
            todo.append(("dec.head", "decoder.gen_head.conv.weight", pk.dlm_channel_perm(hp.n_mixtures)))
        for name, c_prev, c_skip, skip_idx, cout in decoder_layers(hp):
            if cout == 16 and c_prev + c_skip == 32:         # bilinear rows folded into the weights (conv3x3_up16_fold_kernel)
                todo.append((f"dec.{name}", f"decoder.net.{name}.conv.weight", "rowfold"))
            elif cout == 16:                                 # the wave-autonomous 16-channel blocks (conv3x3_up16_split_kernel)
                todo.append((f"dec.{name}", f"decoder.net.{name}.conv.weight", None))
            elif cout in (32, 64) and (c_prev + c_skip) % 32 == 0:     # the workgroup-tiled blocks (conv3x3_up32_split_kernel)
                todo.append((f"dec.{name}", f"decoder.net.{name}.conv.weight", "tiled32"))
        for name, cin, cout, norm in self._enc_layers[1:]:   # encoder 4x4 stride-2 blocks (conv4x4s2_split_kernel)
            if cin == 16 or cin % 32 == 0:
                todo.append((f"enc.{name}", f"encoder.net.{name}.conv.weight", "enc4x4"))
        for name, key, perm in todo:
            off, shp = self._poff[key]
            d = {}
            if perm == "enc4x4":
                idx = pk.conv4x4_split_index(shp, off).to(self.device)
            elif perm == "rowfold":                            # gcpx_fold_upsample_weights(theta + off) -> scratch, split from there
                d["fold"] = torch.zeros(24 * shp[0] * shp[1], dtype=torch.float32, device=self.device)
                d["fold_src"] = (off, shp[0], shp[1])
                idx = pk.conv3x3_fold_index().to(self.device)
            elif perm == "tiled32":
                idx = pk.conv3x3_split32_index(shp, off).to(self.device)
            else:
                idx = pk.conv3x3_split_index(shp, off, perm).to(self.device)
            d.update(idx=idx, out=torch.zeros(2 * idx.numel(), dtype=torch.int16, device=self.device),
                     log2=torch.zeros(1, dtype=torch.int32, device=self.device))
            self.pk_split[name] = d
        self.repack_split()

    def repack_split(self, stream=None):
        """re-split every split-f16 weight tensor from the flat parameter vector: the row-folded blocks' weights are folded first, then ONE
        grouped launch splits all tensors side by side (one workgroup each; as separate launches they were 0.45 ms of a training step)"""
        st = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        if not self.pk_split:
            return
        tab = getattr(self, "_split_tab", None)
        key = (self.theta.data_ptr(),) + tuple((d["out"].data_ptr(), d["idx"].data_ptr()) for d in self.pk_split.values())
        if tab is None or tab[2] != key:
            descs = []
            for name, d in self.pk_split.items():
                src = d["fold"] if "fold" in d else self.theta
                e = rt.SplitPackDesc()
                e.src, e.idx, e.out, e.log2_out, e.n = src.data_ptr(), d["idx"].data_ptr(), d["out"].data_ptr(), d["log2"].data_ptr(), d["idx"].numel()
                descs.append(e)
            arr = (rt.SplitPackDesc * len(descs))(*descs)
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            tab = self._split_tab = (dev, len(descs), key, torch.zeros(len(descs), dtype=torch.int32, device=self.device))
        for name, d in self.pk_split.items():
            if "fold" in d:
                off, cout, cin = d["fold_src"]
                rt.check(self.lib.gcpx_fold_upsample_weights(self.theta.data_ptr() + 4 * off, cout, cin, d["fold"].data_ptr(), st), "fold_upsample_weights")
        rt.check(self.lib.gcpx_split_pack_group2(tab[0].data_ptr(), tab[1], tab[3].data_ptr(), st), "split_pack_group2")

    def _pack_gemm_split(self):
        """Inference only (like the fused embedding: a re-split of every GEMM weight after each optimizer step is not worth its
        launches): the two f16 pieces of the tree levels' LSTM and split_linear weights for csrc/gemm_split.hip, keyed by the
        address of the f32 pack they mirror.  gcpx_gemm takes the split kernel from GCPX_GEMM_SPLIT_MIN_ROWS rows on (default 512:
        below that the launch is bound by the per-CU load rate, not by the f32 MFMA rate — NOTEBOOK.md section 6c)."""
        self._gsplit = {}
        # rows from which a split GEMM with >= 1024 columns takes the two-launch planes form (GCPX_GEMM_PLANES_ROWS; 0 = never)
        self._merge_side_rows = int(os.environ.get("GCPX_MERGE_SIDE_ROWS", "512")) or (1 << 60)   # rows from which a level's merge takes a side lane (0: never)
        pr = int(os.environ.get("GCPX_GEMM_PLANES_ROWS", "512"))
        self._planes_min_rows = pr if pr > 0 else 1 << 60
        if not self.split_f16:
            return
        for name, W in self.pk.items():
            if not (isinstance(W, dict) and name.startswith("tree")):
                continue
            for key, wpk in W.items():
                if not (key.endswith(".w") and (key.startswith("lstm") or key == "proj.w")):
                    continue
                stack = wpk if wpk.dim() == 5 else wpk[None]
                if (stack.shape[2] * 16) % 64 or (stack.shape[1] * 16) % 64:
                    continue
                packs = [pk.pack_gemm_split(pk.unpack_gemm(w, w.shape[1] * 16)) for w in stack]
                ws = torch.stack([p_[0] for p_ in packs]).contiguous().to(self.device)
                es = torch.tensor([p_[1] for p_ in packs], dtype=torch.int32, device=self.device)
                self._gsplit[wpk.data_ptr()] = (ws, es)

    def _set_split(self, a, name):
        """Hand conv `name`'s split-f16 weights to the launch if this model runs split-f16 and holds them."""
        d = getattr(self, "pk_split", {}).get(name)
        if self.split_f16 and d is not None:
            a.wpk_split, a.w_split_log2_dev = d["out"].data_ptr(), d["log2"].data_ptr()
            a.split_layout = rt.SPLIT_ROWFOLD if "fold" in d else rt.SPLIT_PLAIN

    def _pack_fused_embed(self):
        """Inference only: the input embedding Linear and LSTM layer 0's input projection are two Linears with nothing in between
        (tree_lstm.py:43-49 -> HiddenStatePredictorModel: embed, then LSTMCell(embed(x), h)), so gates_0 = (W_ih W_e) [e_l, e_r, z, e_0,
        e_g] + W_hh h + (W_ih b_e + b_ih + b_hh): one launch less on every level's dependent chain.  The product is formed in
        float64 once per weight load.  The training step keeps the two layers apart (its backward needs the embedding) and so does
        any model whose packed weights live in the trainer's arena (a gather of theta cannot express a product)."""
        hp = self._hp
        for l in range(hp.hierarchy_levels if hp.untied_layers else 1):
            p = f"tree_module.tree_modules.{l}.subgoal_pred"
            sd = self.sd
            We, be = sd[f"{p}.embed.weight"].double(), sd[f"{p}.embed.bias"].double()
            Wih = sd[f"{p}.lstm.0.weight_ih"].double()
            Wf = (Wih @ We).float()
            bf = (Wih @ be).float() + sd[f"{p}.lstm.0.bias_ih"]
            w, b = pk.lstm_gate_interleave(Wf, sd[f"{p}.lstm.0.weight_hh"], bf, sd[f"{p}.lstm.0.bias_hh"])
            self.pk[f"tree{l}"]["lstm0f.w"], self.pk[f"tree{l}"]["lstm0f.b"] = pk.pack_gemm(w), b

    def _pack_tree(self, sd):
        """Pure index shuffling of `sd` (any dtype) into the kernels' layouts: {name: tensor | nested dict}."""
        hp = self._hp
        self._psd = sd
        P = {}
        layers, c_top = encoder_layers(hp)
        self._enc_layers, self._c_top = layers, c_top
        P["enc.input.w"] = pk.pack_conv4x4_image(sd["encoder.net.input.conv.weight"])
        P["enc.input.b"] = sd["encoder.net.input.conv.bias"].contiguous()
        for name, cin, cout, norm in layers[1:]:
            P[f"enc.{name}.w"] = pk.pack_conv4x4(sd[f"encoder.net.{name}.conv.weight"])
            P[f"enc.{name}.b"] = sd[f"encoder.net.{name}.conv.bias"].contiguous()
        wh = sd["encoder.net.head.weight"]                         # [nz, C, 4, 4] -> K = (y, x, c) of NHWC 4x4xC
        P["enc.head.w"] = pk.pack_gemm(wh.permute(0, 2, 3, 1).reshape(hp.nz_enc, 16 * c_top))
        P["enc.head.b"] = sd["encoder.net.head.bias"].contiguous()
        wt = sd["decoder.net.input.conv.weight"]                   # ConvTranspose2d [nz, Cd, 4, 4] -> n = (y, x, co)
        P["dec.input.w"] = pk.pack_gemm(wt.permute(2, 3, 1, 0).reshape(16 * c_top, hp.nz_enc))
        P["dec.input.b"] = sd["decoder.net.input.conv.bias"].repeat(16).contiguous()
        for name, c_prev, c_skip, skip_idx, cout in decoder_layers(hp):
            # 16-output-channel blocks run the wave-autonomous kernel, which walks the input in 16-channel chunks
            P[f"dec.{name}.w"] = pk.pack_conv3x3(sd[f"decoder.net.{name}.conv.weight"], 16 if cout == 16 else 32)
            P[f"dec.{name}.b"] = pk.pad_vec(sd[f"decoder.net.{name}.conv.bias"], (cout + 15) // 16 * 16)
        hw, hb = sd["decoder.gen_head.conv.weight"], sd["decoder.gen_head.conv.bias"]
        if hp.decoder_distribution == "discrete_logistic_mixture":
            perm = pk.dlm_channel_perm(hp.n_mixtures)
            self._dlm_perm = torch.tensor(perm, device=self.device)
            P["dec.head.w"] = pk.pack_dlm_head(hw, perm)
            bk = torch.zeros(len(perm), device=hb.device, dtype=hb.dtype)
            permd = self._dlm_perm.to(hb.device)
            valid = permd >= 0
            bk[valid] = hb[permd[valid]]
            P["dec.head.b"] = bk
            self._head_pitch = len(perm)
        else:
            P["dec.head.w"] = pk.pack_conv3x3(hw, 16)
            P["dec.head.b"] = pk.pad_vec(hb, 16)
            self._head_pitch = 16
        seq_encs = [("seq", "inf_encoder")] + ([("kseq", "inf_key_encoder.0")] if hp.attentive_inference else [])
        for tag, pre in seq_encs:
            for nm in ["input"] + [f"pyramid-{i}" for i in range(hp.conv_inf_enc_layers)] + ["head"]:
                w = sd[f"{pre}.net.{nm}.conv.weight"]              # [Cout, Cin, k] -> K = (tap, ci)
                P[f"{tag}.{nm}.w"] = pk.pack_gemm(w.permute(0, 2, 1).reshape(w.shape[0], -1))
                P[f"{tag}.{nm}.b"] = sd[f"{pre}.net.{nm}.conv.bias"].contiguous()
        if hp.attentive_inference:
            P["kseq.key.w"] = pk.pack_gemm(sd["inf_key_encoder.1.linear.weight"])
            P["kseq.key.b"] = sd["inf_key_encoder.1.linear.bias"].contiguous()
        if hp.regress_length:
            P["length_pred"] = self._pack_predictor("length_pred.p", hp.max_seq_len)
        if hp.attach_state_regressor:
            P["state_regressor"] = self._pack_predictor("state_regressor", hp.state_dim)
        if hp.attach_inv_mdl:
            P["inv_mdl"] = self._pack_predictor("inv_mdl.action_pred", hp.n_actions)
        if hp.attach_cost_mdl:
            P["cost_mdl"] = self._pack_predictor("cost_mdl.cost_pred", 1)
        self._pack_latent_model(P)
        return P

    # ---- parameter arena: every packed weight is a gather of the flat parameter vector ----
    def build_arena(self, extra_pack=None):
        """Probe the (linear, 0/1) packing map once with index-valued parameters, then keep all packed weights in one
        arena refreshed by gcpx_repack.  `extra_pack(sd) -> dict` adds more packs (the transposed ones of the backward)."""
        dev = self.device
        def probe(only_hh):
            sd = {}
            for k, (o, shp) in self._poff.items():
                n = 1
                for d in shp:
                    n *= d
                hh = k.endswith("bias_hh")
                if hh == only_hh:
                    sd[k] = (torch.arange(n, device=dev, dtype=torch.float64) + (o + 1)).view(shp)
                else:
                    sd[k] = torch.zeros(shp, device=dev, dtype=torch.float64)
            P = self._pack_tree(sd)
            X = extra_pack(sd) if extra_pack is not None else {}
            return P, X
        (P0, X0), (P1, X1) = probe(False), probe(True)
        leaves = []
        def walk(d0, d1, path):
            for k in d0:
                if isinstance(d0[k], dict):
                    walk(d0[k], d1[k], path + (k,))
                elif torch.is_tensor(d0[k]):
                    leaves.append((path + (k,), d0, d0[k], d1[k]))
        walk(P0, P1, ("P",))
        walk(X0, X1, ("X",))
        # leaves that sum two parameters (the fused LSTM biases b_ih + b_hh) go last: the bulk of the arena is then re-packed without
        # reading a second index array
        # ... and inside both halves the leaves are grouped by the slice of the flat vector they gather from (dist.gradient_bucket_ranges:
        # one per untied tree level, "rest" for everything else and for leaves that stack several levels): a trainer that applies the
        # optimizer slice by slice during the backward pass re-packs each slice's leaves as ONE contiguous run per half (repack(bucket=))
        from .dist import gradient_bucket_ranges
        ranges = gradient_bucket_ranges(self._poff, self._hp.hierarchy_levels, self._hp.untied_layers)
        rest = len(ranges) - 1
        def bucket_of(t0, t1):
            ids = torch.cat([t0.reshape(-1), t1.reshape(-1)])
            ids = ids[ids > 0] - 1
            if ids.numel() == 0:
                return rest
            lo, hi = int(ids.min()), int(ids.max())
            for i, (_, a, b) in enumerate(ranges):
                if a <= lo and hi < b:
                    return i
            return rest
        leaves = [lf + (bucket_of(lf[2], lf[3]),) for lf in leaves]
        leaves.sort(key=lambda lf: (bool((lf[3] > 0).any()), lf[4]))
        leaves = [lf[:4] + (lf[4],) for lf in leaves]
        total = sum((lf[2].numel() + 3) // 4 * 4 for lf in leaves)
        self._arena = torch.zeros(total, dtype=torch.float32, device=dev)
        idx0 = torch.full((total,), -1, dtype=torch.int32, device=dev)
        idx1 = torch.full((total,), -1, dtype=torch.int32, device=dev)
        off = 0
        self._arena_split = None
        runs = {}                                  # (bucket, two-index half?) -> [first element, one past the last]
        for path, holder, t0, t1, bkt in leaves:
            n = t0.numel()
            two = bool((t1 > 0).any())
            if self._arena_split is None and two:
                self._arena_split = off
            idx0[off:off + n] = (t0.reshape(-1) - 1).to(torch.int32)
            idx1[off:off + n] = (t1.reshape(-1) - 1).to(torch.int32)
            holder[path[-1]] = self._arena[off:off + n].view(t0.shape)
            r = runs.setdefault((bkt, two), [off, off])
            assert r[1] == off, "leaves of one slice are contiguous inside a half"
            off += (n + 3) // 4 * 4
            r[1] = off
        self._arena_runs = {ranges[b][0]: [] for b in range(len(ranges))}
        for (bkt, two), (a, b) in sorted(runs.items()):
            self._arena_runs[ranges[bkt][0]].append((a, b - a, two))
        self._arena_ranges = ranges
        self._arena_idx0, self._arena_idx1 = idx0, idx1
        self._psd = self.sd
        self.pk = P0
        self._clear_plans()
        self.repack()
        return X0

    def repack(self, stream=None, bucket=None, max_blocks=0):
        """bucket = None: every packed weight.  bucket = a name of dist.gradient_bucket_ranges: only the leaves that gather from that
        slice of the flat vector ("rest" also re-splits the split-f16 tensors, which all gather from it), in launches of at most
        max_blocks workgroups (0: no limit)."""
        st = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        if bucket is not None:
            for off, cnt, two in self._arena_runs[bucket]:
                rt.check(self.lib.gcpx_repack_blocks(self.theta.data_ptr(), self._arena_idx0.data_ptr() + 4 * off,
                                                     (self._arena_idx1.data_ptr() + 4 * off) if two else None, self._arena.data_ptr() + 4 * off,
                                                     cnt, max_blocks, st), "repack")
            self._repack_gsplit(st, bucket)
            if bucket == self._arena_ranges[-1][0]:
                self.repack_split(st)
            return
        n, sp = self._arena.numel(), self._arena_split
        sp = n if sp is None else sp
        if sp > 0:
            rt.check(self.lib.gcpx_repack(self.theta.data_ptr(), self._arena_idx0.data_ptr(), None, self._arena.data_ptr(), sp, st), "repack")
        if sp < n:
            rt.check(self.lib.gcpx_repack(self.theta.data_ptr(), self._arena_idx0.data_ptr() + 4 * sp, self._arena_idx1.data_ptr() + 4 * sp,
                                          self._arena.data_ptr() + 4 * sp, n - sp, st), "repack")
        self._repack_gsplit(st, None)
        self.repack_split(st)

    def _repack_gsplit(self, st, bucket):
        """re-split the GEMM weights a trainer keeps in split-f16 form (training.py: _live_gemm_split) that gather from `bucket`'s slice of
        the flat vector (None: all of them)"""
        for name, (tab, n, scratch) in getattr(self, "_gsplit_tabs", {}).items():
            if bucket is None or bucket == name:
                rt.check(self.lib.gcpx_split_pack_group2(tab.data_ptr(), n, scratch.data_ptr(), st), "split_pack_group2")

    def _pack_hsp(self, prefix, n_layers):
        """embed Linear + n gate-interleaved LSTM layers + out Linear of one recurrent predictor."""
        sd, T = self._psd, {}
        T["embed.w"] = pk.pack_gemm(sd[f"{prefix}.embed.weight"])
        T["embed.b"] = sd[f"{prefix}.embed.bias"].contiguous()
        for i in range(n_layers):
            w, b = pk.lstm_gate_interleave(sd[f"{prefix}.lstm.{i}.weight_ih"], sd[f"{prefix}.lstm.{i}.weight_hh"],
                                           sd[f"{prefix}.lstm.{i}.bias_ih"], sd[f"{prefix}.lstm.{i}.bias_hh"])
            T[f"lstm{i}.w"], T[f"lstm{i}.b"] = pk.pack_gemm(w), b
        T["out.w"] = pk.pack_gemm(sd[f"{prefix}.out.weight"])
        T["out.b"] = sd[f"{prefix}.out.bias"].contiguous()
        return T

    def _pack_latent_model(self, P):
        hp, sd = self._hp, self._psd
        if hp.adaptive:
            P["distance"] = self._pack_predictor("tree_module.tree_modules.0.binding.distance_predictor", 1)
        else:
            P["existence"] = self._pack_predictor("tree_module.tree_modules.0.binding.existence_predictor", 1)
        H = hp.nz_mid_lstm
        n_mod = hp.hierarchy_levels if hp.untied_layers else 1
        if hp.attentive_inference:
            # key / value projections of every level's attention stacked: one batched launch each (blockIdx.z = level)
            att = lambda l, nm: sd[f"tree_module.tree_modules.{l}.inference.attention.attention_layers.0.{nm}"]
            for nm in ("k_proj", "v_proj"):
                P[f"attn.{nm}.w"] = torch.stack([pk.pack_gemm(att(l, f"{nm}.weight")) for l in range(n_mod)]).contiguous()
                P[f"attn.{nm}.b"] = torch.stack([att(l, f"{nm}.bias") for l in range(n_mod)]).contiguous()
        for l in range(hp.hierarchy_levels if hp.untied_layers else 1):
            p = f"tree_module.tree_modules.{l}"
            T = {}
            T["prior"] = self._pack_predictor(f"{p}.prior", 2 * hp.nz_vae)
            T["q"] = self._pack_predictor(f"{p}.inference.q", 2 * hp.nz_vae)
            if not hp.tree_lstm:
                T["sg"] = self._pack_predictor(f"{p}.subgoal_pred.net", hp.nz_enc)
                P[f"tree{l}"] = T
                if hp.attentive_inference:
                    raise ValueError("attentive inference with the non-LSTM subgoal predictor is not built")
                continue
            T["embed.w"] = pk.pack_gemm(sd[f"{p}.subgoal_pred.embed.weight"])
            T["embed.b"] = sd[f"{p}.subgoal_pred.embed.bias"].contiguous()
            for i in range(hp.n_lstm_layers):
                w, b = pk.lstm_gate_interleave(sd[f"{p}.subgoal_pred.lstm.{i}.weight_ih"], sd[f"{p}.subgoal_pred.lstm.{i}.weight_hh"],
                                               sd[f"{p}.subgoal_pred.lstm.{i}.bias_ih"], sd[f"{p}.subgoal_pred.lstm.{i}.bias_hh"])
                T[f"lstm{i}.w"], T[f"lstm{i}.b"] = pk.pack_gemm(w), b
            T["out.w"] = pk.pack_gemm(sd[f"{p}.subgoal_pred.out.weight"])
            T["out.b"] = sd[f"{p}.subgoal_pred.out.bias"].contiguous()
            nproj = 2 * hp.n_lstm_layers
            if hp.tree_lstm == "split_linear":
                T["proj.w"] = torch.stack([pk.pack_gemm(sd[f"{p}.subgoal_pred.projections.{j}.weight"]) for j in range(nproj)]).contiguous()
                T["proj.b"] = torch.stack([sd[f"{p}.subgoal_pred.projections.{j}.bias"] for j in range(nproj)]).contiguous()
            elif hp.tree_lstm == "linear":
                T["proj.w"] = pk.pack_gemm(sd[f"{p}.subgoal_pred.projection.weight"])
                T["proj.b"] = sd[f"{p}.subgoal_pred.projection.bias"].contiguous()
            if l == 0 and hp.lstm_init == "mlp":
                T["init"] = self._pack_predictor(f"{p}.lstm_initializer.net", 2 * hp.lstm_state_dim)
            if hp.attentive_inference:
                a = f"{p}.inference.attention"
                T["attn.query"] = self._pack_predictor(f"{a}.query_net", hp.nz_attn_key)
                for nm, key in (("q_proj", f"{a}.attention_layers.0.q_proj"), ("out_proj", f"{a}.attention_layers.0.out_proj"),
                                ("out", f"{a}.out")):
                    T[f"attn.{nm}.w"] = pk.pack_gemm(sd[f"{key}.weight"])
                    T[f"attn.{nm}.b"] = sd[f"{key}.bias"].contiguous()
            P[f"tree{l}"] = T

    # ------------------------------------------------------------------------------------------------
    # buffers and plan-building helpers
    # ------------------------------------------------------------------------------------------------
    _buf_prefix = ""          # handle plans (model.encoder / model.decoder) keep their activations apart from the forward's

    def _buf(self, name, shape, dtype=torch.float32, zero=False):
        key = (self._buf_prefix + name, tuple(shape), dtype)
        t = self._bufs.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    @staticmethod
    def _rowsrc(ptr, sb, sr, width, shift=0, rowidx=None, scale=None, shiftv=None, act=0, cmod=0):
        s = rt.RowSrc()
        s.ptr, s.rowidx = ptr, (rowidx.data_ptr() if rowidx is not None else None)
        s.scale = scale.data_ptr() if scale is not None else None
        s.shiftv = shiftv.data_ptr() if shiftv is not None else None
        s.sb, s.sr, s.width, s.shift, s.act, s.cmod = sb, sr, width, shift, act, cmod
        return s

    @staticmethod
    def _dense_rows(srcs, rpb, M):
        """One row per batch element (tree level 0, the I_0 / I_g encoder heads): the kernels tile rows inside a batch
        element, so rpb = 1 would mean one-row tiles.  Re-express the same addresses as ONE batch element of M rows
        (row stride = the old batch stride): the MFMA tiles are full again and the launch is one workgroup column."""
        if rpb != 1 or M == 1 or any(s.shift != 0 for s in srcs):
            return srcs, rpb, False
        out = []
        for s in srcs:
            t = rt.RowSrc()
            t.ptr, t.rowidx, t.scale, t.shiftv = s.ptr, s.rowidx, s.scale, s.shiftv
            t.sb, t.sr = 0, (s.sr if s.rowidx else s.sb)
            t.width, t.shift, t.act, t.cmod = s.width, s.shift, s.act, s.cmod
            out.append(t)
        return out, M, True

    def _gemm_group(self, plan, name, group):
        """independent small-M GEMMs as one launch (gcpx_gemm_group); problems outside the split-K regime are launched one by one"""
        if len(group) > 1:
            n = len(group)
            tab = (rt.GemmArgs * n)(*[a for _, a in group])
            dims = (C.c_int32 * (4 * n))()
            total = C.c_int32()
            if self.lib.gcpx_gemm_group_dims(tab, n, dims, C.byref(total)) == 0:
                raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.device)
                dd = torch.tensor(list(dims), dtype=torch.int32, device=self.device)
                plan.keep += [raw, dd, tab]
                plan.add(name, self.lib.gcpx_gemm_group, raw.data_ptr(), dd.data_ptr(), n, total.value)
                return
        for nm, a in group:
            if a.epi == rt.EPI_GAUSS_SAMPLE:                 # the reparametrised draw that rides in grouped launches (sequential.py)
                m, e = a.src[0], a.src[1]
                plan.add(nm, self.lib.gcpx_gauss_sample, m.ptr, m.sb, m.sr, e.ptr, e.sb, e.sr, a.out, a.ob, a.orow, a.M, a.rpb, a.N)
            else:
                plan.add(nm, self.lib.gcpx_gemm, C.byref(a))

    def _gemm(self, plan, name, srcs, M, N, rpb, wpk, bias, out=None, ob=0, orow=0, epi=rt.EPI_NONE,
              stats=None, lstm=None, batch=None, group=None, lstm_bwd=None):
        srcs, rpb, dense = self._dense_rows(srcs, rpb, M)
        if dense:
            ob, orow = 0, ob
            if lstm is not None:
                c_prev, c_prev_stride, h_out, c_out, hb, hrow, h_copy = lstm
                lstm = (c_prev, c_prev_stride, h_out, c_out, 0, hb, h_copy)
        a = rt.GemmArgs()
        for i, s in enumerate(srcs):
            a.src[i] = s
        a.nsrc, a.M, a.N, a.K, a.rpb = len(srcs), M, N, sum(s.width for s in srcs), rpb
        a.wpk, a.bias = wpk.data_ptr(), (bias.data_ptr() if bias is not None else None)
        gs = getattr(self, "_gsplit", {}).get(wpk.data_ptr())
        # (a trainer's model: only packs the trainer re-splits behind every optimizer step — training.py: _live_gemm_split)
        if gs is not None and self.split_f16 and (getattr(self, "_arena", None) is None or getattr(self, "_gsplit_live", False)):
            a.wpk_split, a.w_split_log2_dev = gs[0].data_ptr(), gs[1].data_ptr()
        a.out, a.ob, a.orow, a.epi = out, ob, orow, epi
        a.stats_partial = stats.data_ptr() if stats is not None else None
        if lstm is not None:
            a.c_prev, a.c_prev_stride, a.h_out, a.c_out, a.hb, a.hrow, a.h_copy = lstm
            if self.save_for_backward:
                g = self._buf(f"gates.{name}", (M, N))
                a.gates_out = g.data_ptr()
                plan.rec[f"gates:{name}"] = g
        if batch is not None:
            a.nbatch, a.z_src_off, a.z_w_off, a.z_bias_off, a.z_out_off = batch
        if lstm_bwd is not None:                 # device copy of the LstmBwdArgs of the layer this gradient feeds (gcpx_gemm_args.lstm_bwd)
            a.lstm_bwd = lstm_bwd
        if a.wpk_split and M >= getattr(self, "_planes_min_rows", 1 << 60) and N >= 1024 and group is None and not a.stats_partial:
            # many rows x many columns: conversion pass + LDS-DMA fed GEMM (csrc/gemm_planes.hip).  The workspace is shared by the launches
            # of one lane that need the same size (a lane is a stream: its launches are ordered)
            nbytes, nexp = C.c_int64(), C.c_int64()
            rt.check(self.lib.gcpx_gemm_planes_workspace(M, a.K, a.nbatch, C.byref(nbytes), C.byref(nexp)), "planes workspace")
            wsb = self._buf(f"xplanes.l{plan.lane}", (nbytes.value,), torch.uint8)
            wse = self._buf(f"xexp.l{plan.lane}", (nexp.value,), torch.int32)
            a.x_planes, a.x_exp, a.x_planes_bytes = wsb.data_ptr(), wse.data_ptr(), nbytes.value
        plan.keep.append(a)
        if group is not None:
            group.append((name, a))
            return
        plan.add(name, self.lib.gcpx_gemm, C.byref(a))

    def _mlp(self, plan, name, W, srcs, M, rpb, out=None, ob=0, orow=0, oblk=0, out_split=0, gauss=None, group=None, tanh=False):
        """One Predictor launch — or, with `group` (a list), only its argument struct: `_mlp_group` then issues the whole list as
        ONE launch."""
        hp = self._hp
        rec_srcs, rec_rpb = srcs, rpb            # the backward plan addresses rows the way the caller does
        srcs, rpb, dense = self._dense_rows(srcs, rpb, M)
        if dense:
            ob, orow = 0, ob
            if gauss is not None:
                eps, eb, erow, z, zb, zrow = gauss
                gauss = (eps, 0, eb, z, 0, zb)
        a = rt.MlpArgs()
        for i, s in enumerate(srcs):
            a.src[i] = s
        a.nsrc, a.M, a.rpb = len(srcs), M, rpb
        a.in_dim, a.mid, a.n_mid, a.out_dim = W["in_dim"], W["mid"], W["n_mid"], W["out_dim"]
        assert a.in_dim == sum(s.width for s in srcs), (name, a.in_dim)
        a.w_in, a.b_in = W["w_in"].data_ptr(), W["b_in"].data_ptr()
        if W["n_mid"]:
            a.w_mid, a.b_mid = W["w_mid"].data_ptr(), W["b_mid"].data_ptr()
            a.gn_gamma, a.gn_beta = W["gn_g"].data_ptr(), W["gn_b"].data_ptr()
        a.w_out, a.b_out = W["w_out"].data_ptr(), W["b_out"].data_ptr()
        a.gn_eps, a.lrelu_slope = hp.gn_eps, hp.leaky_slope
        a.out, a.ob, a.orow, a.oblk, a.out_split = out, ob, orow, oblk, out_split
        a.epi = rt.MLP_TANH if tanh else rt.MLP_PLAIN
        if gauss is not None:
            a.epi = rt.MLP_GAUSS
            a.eps, a.eb, a.erow, a.z, a.zb, a.zrow = gauss
        if self.save_for_backward:
            sv = self._buf(f"save.{name}", (1 + 2 * W["n_mid"], M, W["mid"]))
            a.save = sv.data_ptr()
            plan.rec[f"mlp:{name}"] = dict(W=W, srcs=rec_srcs, M=M, rpb=rec_rpb, save=sv)
        plan.keep.append(a)
        if group is not None:
            group.append((name, a))
            return
        plan.add(name, self.lib.gcpx_mlp, C.byref(a))

    def _mlp_group(self, plan, name, group, gemm=None):
        """independent Predictors of one hidden width as one launch (descriptor table uploaded once, when the plan is built).
        gemm: a (name, GemmArgs) that depends on none of them and rides in the same launch when there is a combined kernel for its
        tiling (gcpx_mlp_group_gemm), else it is launched first."""
        if len(group) == 1:
            if gemm is not None:
                plan.add(gemm[0], self.lib.gcpx_gemm, C.byref(gemm[1]))
            plan.add(group[0][0], self.lib.gcpx_mlp, C.byref(group[0][1]))
            return
        n = len(group)
        tab = (rt.MlpArgs * n)(*[a for _, a in group])
        dims = (C.c_int32 * (4 * n))()
        total = C.c_int32()
        rt.check(self.lib.gcpx_mlp_group_dims(tab, n, dims, C.byref(total)), name)
        raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.device)
        dd = torch.tensor(list(dims), dtype=torch.int32, device=self.device)
        plan.keep += [raw, dd, tab]
        mid = group[0][1].mid
        if gemm is not None:
            if os.environ.get("GCPX_NO_LEVEL_PRE") is None and self.lib.gcpx_mlp_group_gemm_supported(C.byref(gemm[1]), total.value, mid):
                plan.add(f"{name}+{gemm[0]}", self.lib.gcpx_mlp_group_gemm, raw.data_ptr(), dd.data_ptr(), n, total.value, mid, C.byref(gemm[1]))
                return
            plan.add(gemm[0], self.lib.gcpx_gemm, C.byref(gemm[1]))
        plan.add(name, self.lib.gcpx_mlp_group, raw.data_ptr(), dd.data_ptr(), n, total.value, mid)

    def _bn(self, plan, tag, prefix, C_, stats, n_partial, pitch, count):
        """(scale, shift) of a BatchNorm: batch statistics when training, running statistics otherwise."""
        sd, hp = self.sd, self._hp
        scale, shift = self._buf(f"{tag}.scale", (C_,)), self._buf(f"{tag}.shift", (C_,))
        g, b = sd[f"{prefix}.weight"], sd[f"{prefix}.bias"]
        if self.training:
            mean = rstd = None
            if self.save_for_backward:
                mean, rstd = self._buf(f"{tag}.mean", (C_,)), self._buf(f"{tag}.rstd", (C_,))
                plan.rec[f"bn:{tag}"] = dict(prefix=prefix, C=C_, count=count, scale=scale, shift=shift, mean=mean, rstd=rstd)
            plan.add(f"bn_finalize:{tag}", self.lib.gcpx_bn_finalize, stats.data_ptr(), n_partial, pitch, C_,
                     C.c_double(float(count)), g.data_ptr(), b.data_ptr(), C.c_float(hp.bn_eps), scale.data_ptr(),
                     shift.data_ptr(), None, None, C.c_float(0.0), rt.ptr(mean), rt.ptr(rstd))
        else:
            plan.add(f"bn_fold:{tag}", self.lib.gcpx_bn_fold, sd[f"{prefix}.running_mean"].data_ptr(),
                     sd[f"{prefix}.running_var"].data_ptr(), g.data_ptr(), b.data_ptr(), C.c_float(hp.bn_eps), C_,
                     scale.data_ptr(), shift.data_ptr())
        return scale, shift

    def _conv_args(self, srcs, F, Hin, Win, Hout, Wout, Cout, out_pitch, wpk, bias, out, upsample=0, out_act=0,
                   head_mode=rt.HEAD_RAW, images=None, stats=None):
        a = rt.ConvArgs()
        cin = 0
        for i, (t_ptr, C_, fdiv, scale, shift, act) in enumerate(srcs):
            s = a.src[i]
            s.ptr, s.C, s.frame_div, s.act = t_ptr, C_, fdiv, act
            s.scale = scale.data_ptr() if scale is not None else None
            s.shift = shift.data_ptr() if shift is not None else None
            cin += C_
        a.nsrc, a.F, a.Hin, a.Win, a.Hout, a.Wout, a.Cin, a.Cout = len(srcs), F, Hin, Win, Hout, Wout, cin, Cout
        a.out_pitch, a.upsample, a.out_act, a.head_mode = out_pitch, upsample, out_act, head_mode
        a.wpk, a.bias = wpk.data_ptr(), bias.data_ptr()
        a.out = out.data_ptr() if out is not None else None
        a.images = images.data_ptr() if images is not None else None
        a.stats_partial = stats.data_ptr() if stats is not None else None
        return a

    # ------------------------------------------------------------------------------------------------
    # plan: encoder
    # ------------------------------------------------------------------------------------------------
    def _plan_encoder(self, plan, tag, x_ptr, F, out_ptr, out_ob, out_orow, out_rpb):
        """Encoder over F NCHW frames at x_ptr; writes the nz_enc latent of frame r=(b,j) to
        out_ptr + b*out_ob + j*out_orow.  Returns the skip sources {module index: (tensor, C, scale, shift, act)}."""
        hp, P, lib = self._hp, self.pk, self.lib
        G = lib.gcpx_conv4x4s2_grid()
        S = hp.img_sz
        skip_idx = encoder_skip_layers(hp)
        skips = {}
        a0 = self._buf(f"{tag}.a0", (F, S // 2, S // 2, hp.ngf))
        enc_rec = dict(F=F, x_ptr=x_ptr, a0=a0, r={}, out=(out_ptr, out_ob, out_orow, out_rpb))
        plan.rec[f"enc:{tag}"] = enc_rec
        plan.add(f"enc.input:{tag}", lib.gcpx_conv4x4s2_image, x_ptr, P["enc.input.w"].data_ptr(),
                 P["enc.input.b"].data_ptr(), a0.data_ptr(), F, S, S, hp.ngf, rt.ACT_LRELU)
        prev = (a0.data_ptr(), hp.ngf, 1, None, None, rt.ACT_NONE)
        if 0 in skip_idx:
            skips[0] = (a0, hp.ngf, None, None, rt.ACT_NONE)
        res = S // 2
        for li, (name, cin, cout, norm) in enumerate(self._enc_layers[1:], start=1):
            r = self._buf(f"{tag}.r{li}", (F, res // 2, res // 2, cout))
            enc_rec["r"][li] = r
            stats = self._buf(f"{tag}.st{li}", (G, 2, cout)) if self.training else None
            a = self._conv_args([prev], F, res, res, res // 2, res // 2, cout, cout, P[f"enc.{name}.w"],
                                P[f"enc.{name}.b"], r, stats=stats)
            self._set_split(a, f"enc.{name}")
            plan.keep.append(a)
            plan.add(f"enc.{name}:{tag}", lib.gcpx_conv4x4s2, C.byref(a))
            res //= 2
            scale, shift = self._bn(plan, f"{tag}.bn{li}", f"encoder.net.{name}.norm", cout, stats, G, cout,
                                    F * res * res)
            prev = (r.data_ptr(), cout, 1, scale, shift, rt.ACT_LRELU)
            if li in skip_idx:
                skips[li] = (r, cout, scale, shift, rt.ACT_LRELU)
        assert res == 4
        ctop = self._c_top
        src = self._rowsrc(prev[0], 16 * ctop, 16 * ctop, 16 * ctop, scale=prev[3], shiftv=prev[4], act=prev[5], cmod=ctop)
        # rows are frames; caller's row map decides where each latent lands
        src.sb, src.sr = out_rpb * 16 * ctop, 16 * ctop
        self._gemm(plan, f"enc.head:{tag}", [src], F, hp.nz_enc, out_rpb, P["enc.head.w"], P["enc.head.b"],
                   out=out_ptr, ob=out_ob, orow=out_orow)
        return skips

    def _plan_seq_encoder(self, plan, tag, prefix, enc_traj, out, B):
        """ConvSeqEncodingModule (base_gcp.py:130-134): three conv1d over time as shifted-row GEMMs."""
        hp, P, lib = self._hp, self.pk, self.lib
        T, nz = hp.max_seq_len, hp.nz_enc
        y1 = self._buf(f"{tag}.y1", (B * T, hp.nz_mid))
        y2 = self._buf(f"{tag}.y2", (B * T, hp.nz_mid))
        taps = lambda t, w, **kw: [self._rowsrc(t.data_ptr(), T * w, w, w, shift=d, **kw) for d in (-1, 0, 1)]
        self._gemm(plan, f"{tag}.input", taps(enc_traj, nz), B * T, hp.nz_mid, T, P[f"{tag}.input.w"], P[f"{tag}.input.b"],
                   out=y1.data_ptr(), ob=T * hp.nz_mid, orow=hp.nz_mid, epi=rt.EPI_LRELU)
        assert hp.conv_inf_enc_layers == 1
        nrb = lib.gcpx_gemm_row_blocks(B * T, hp.nz_mid)
        st = self._buf(f"{tag}.st", (nrb, 2, hp.nz_mid)) if self.training else None
        self._gemm(plan, f"{tag}.pyramid-0", taps(y1, hp.nz_mid), B * T, hp.nz_mid, T, P[f"{tag}.pyramid-0.w"],
                   P[f"{tag}.pyramid-0.b"], out=y2.data_ptr(), ob=T * hp.nz_mid, orow=hp.nz_mid, stats=st)
        sc, sh = self._bn(plan, f"{tag}.bn", f"{prefix}.net.pyramid-0.norm", hp.nz_mid, st, nrb, hp.nz_mid, B * T)
        self._gemm(plan, f"{tag}.head", taps(y2, hp.nz_mid, scale=sc, shiftv=sh, act=rt.ACT_LRELU, cmod=hp.nz_mid),
                   B * T, nz, T, P[f"{tag}.head.w"], P[f"{tag}.head.b"], out=out.data_ptr(), ob=T * nz, orow=nz)

    def _plan_attention(self, plan, l, W, el, er, M, n, B, Kp, Vp, tin):
        """Attention.forward for one tree level (attentive_inference.py:47-86, one layer, mask = the sequence's own
        [start_ind, end_ind]): query MLP -> q_proj -> masked softmax over the T frames -> out_proj -> Attention.out.
        Returns the row source of e_tilde [M, nz_enc]; the attention weights (gamma) stay in plan.rec."""
        hp, lib = self._hp, self.lib
        T, nz, dk = hp.max_seq_len, hp.nz_enc, hp.nz_attn_key
        li = l if hp.untied_layers else 0
        qin = self._buf(f"attn.qin{l}", (M, dk))
        self._mlp(plan, f"attn.query{l}", W["attn.query"], [el, er], M, n, out=qin.data_ptr(), ob=n * dk, orow=dk)
        dense = lambda t, w: self._rowsrc(t.data_ptr(), 0, w, w)
        qp = self._buf(f"attn.q{l}", (M, dk))
        self._gemm(plan, f"attn.q_proj{l}", [dense(qin, dk)], M, dk, M, W["attn.q_proj.w"], W["attn.q_proj.b"],
                   out=qp.data_ptr(), ob=0, orow=dk)
        o = self._buf(f"attn.o{l}", (M, nz))
        gamma = self._buf(f"attn.gamma{l}", (M, T))
        temp = self.sd[f"tree_module.tree_modules.{li}.inference.attention.attention_layers.0.temperature"]
        plan.add(f"attn{l}", lib.gcpx_attention, qp.data_ptr(), _addr(Kp, li * B * T * dk), _addr(Vp, li * B * T * nz), None,
                 tin["end_ind"].data_ptr(), temp.data_ptr(), o.data_ptr(), gamma.data_ptr(), M, n, T, dk, nz, hp.n_attention_heads)
        raw = self._buf(f"attn.raw{l}", (M, nz))
        self._gemm(plan, f"attn.out_proj{l}", [dense(o, nz)], M, nz, M, W["attn.out_proj.w"], W["attn.out_proj.b"],
                   out=raw.data_ptr(), ob=0, orow=nz)
        et = self._buf(f"attn.e_tilde{l}", (M, nz))
        self._gemm(plan, f"attn.out{l}", [dense(raw, nz)], M, nz, M, W["attn.out.w"], W["attn.out.b"], out=et.data_ptr(), ob=0, orow=nz)
        plan.rec.setdefault("gamma", {})[l] = gamma
        plan.rec.setdefault("e_tilde", {})[l] = et
        plan.rec.setdefault("attn", {})[l] = dict(qin=qin, qp=qp, o=o, raw=raw, gamma=gamma, et=et, M=M, n=n, li=li, temp=temp)
        return self._rowsrc(et.data_ptr(), n * nz, nz, nz)          # rows (b, j) of the level, as the posterior MLP walks them

    def _plan_decoder_features(self, plan, e_src, F, rpb, skips):
        """ConvDecoder up to (not including) the output head over F latents given by the row source `e_src` (rows are
        (b, j), j < rpb); the skip activations of I_0 are broadcast over the rpb frames of a sequence.  Returns the head's
        input source tuple (raw 16-channel features + their BatchNorm affine)."""
        hp, P, lib = self._hp, self.pk, self.lib
        ctop = self._c_top
        d0 = self._buf("dec.d0", (F, 4, 4, ctop))
        plan.rec["dec"] = dict(F=F, rpb=rpb, e_src=e_src, d0=d0, blocks=[], skips=skips)
        nrb = lib.gcpx_gemm_row_blocks(F, 16 * ctop)
        st = self._buf("dec.st0", (nrb, 2, 16 * ctop)) if self.training else None
        self._gemm(plan, "dec.input", [e_src], F, 16 * ctop, rpb, P["dec.input.w"], P["dec.input.b"], out=d0.data_ptr(),
                   ob=rpb * 16 * ctop, orow=16 * ctop, stats=st)
        sc, sh = self._bn(plan, "dec.bn0", "decoder.net.input.norm", ctop, st, nrb, 16 * ctop, F * 16)
        prev = (d0.data_ptr(), ctop, 1, sc, sh, rt.ACT_LRELU)
        res = 4
        for name, c_prev, c_skip, skip_idx, cout in decoder_layers(hp):
            srcs = [prev]
            if skip_idx >= 0:
                t, C_, ssc, ssh, sact = skips[skip_idx]
                assert C_ == c_skip and t.shape[1] == res
                srcs.append((t.data_ptr(), C_, rpb, ssc, ssh, sact))    # skips of I_0 broadcast over the sequence's frames
            o = self._buf(f"dec.{name}", (F, 2 * res, 2 * res, cout))
            cpad = (cout + 15) // 16 * 16
            a = self._conv_args(srcs, F, res, res, 2 * res, 2 * res, cout, cout, P[f"dec.{name}.w"], P[f"dec.{name}.b"],
                                o, upsample=1, stats=(o if self.training else None))     # placeholder pointer for the query
            Gl = lib.gcpx_conv3x3_grid(C.byref(a))
            assert Gl > 0, rt.lib().gcpx_last_error()
            st = self._buf(f"dec.st.{name}", (Gl, 2, cpad)) if self.training else None
            a.stats_partial = st.data_ptr() if st is not None else None
            plan.keep.append(a)
            self._set_split(a, f"dec.{name}")
            plan.add(f"dec.{name}", lib.gcpx_conv3x3, C.byref(a))
            plan.rec["dec"]["blocks"].append(dict(name=name, srcs=srcs, out=o, res_in=res, cout=cout, c_prev=c_prev, c_skip=c_skip,
                                                  skip_idx=skip_idx))
            res *= 2
            sc, sh = self._bn(plan, f"dec.bn.{name}", f"decoder.net.{name}.norm", cout, st, Gl, cpad, F * res * res)
            prev = (o.data_ptr(), cout, 1, sc, sh, rt.ACT_LRELU)
        assert res == hp.img_sz
        return prev

    # ------------------------------------------------------------------------------------------------
    # plan: whole forward
    # ------------------------------------------------------------------------------------------------
    def _build_plan(self, key, tin):
        hp, P, lib = self._hp, self.pk, self.lib
        B, has_traj, has_z, sample_prior, phase = key[0], key[1], key[2], key[3], key[4]
        pred_len = key[9]
        train_aux = has_traj and phase == "train" and not sample_prior    # the posterior path of a training / validation-loss forward
        L, T, N = hp.hierarchy_levels, hp.max_seq_len, hp.n_nodes
        nz, nv, H, SD = hp.nz_enc, hp.nz_vae, hp.nz_mid_lstm, hp.lstm_state_dim
        PS = 2 ** L + 1                                     # slots per batch element
        plan = _Plan(lib)
        G = lib.gcpx_conv_grid()
        def plan_aux_indices():
            AUXK = ("inv_t0", "inv_t1", "cost_start_idx", "cost_end_idx")
            plan.add("aux_sample_indices", lib.gcpx_aux_sample_indices_gauss, tin["end_ind"].data_ptr(), tin["aux_n"].data_ptr(), B,
                     hp.inv_mdl_temp_dist, *[tin[k].data_ptr() for k in AUXK])
        if "aux_n" in tin and pred_len:
            plan_aux_indices()

        E = self._buf("E", (B, PS, nz))
        Hid = self._buf("Hid", (B, PS, SD))
        Z = self._buf("Z", (B, PS, nv))
        PZ = self._buf("PZ", (B, PS, 2 * nv))
        QZ = self._buf("QZ", (B, PS, 2 * nv))
        node_t = self._buf("node_t", (B, N), torch.int32)
        leave = self._buf("leave", (B, N), torch.int32)
        f2n = self._buf("frame2node", (B, T), torch.int32)
        etrow = self._buf("etilde_row", (B * N,), torch.int32)
        seq_len = self._buf("seq_len", (B,), torch.int32)
        kept_idx = self._buf("kept_idx", (B, T), torch.int32)
        node2row = self._buf("node2row", (B, N), torch.int32)

        adaptive, attentive = hp.adaptive, hp.attentive_inference
        if adaptive:
            # learned pruning keeps up to N nodes (adaptive.py:62-77): the kept-position table is N wide
            kept_idx = self._buf("kept_idx", (B, N), torch.int32)
        def plan_bookkeeping():
            # ---- integer bookkeeping (frame_binding.py:42-65, evaluation_matching.py:192-206): first needed by the tree's
            # posterior gather, so it rides on a side lane next to an encoder pass instead of in front of the trajectory encoder
            if not adaptive:
                plan.add("balanced_binding", lib.gcpx_balanced_binding, tin["end_ind"].data_ptr(), B, L, T, node_t.data_ptr(),
                         leave.data_ptr(), f2n.data_ptr(), etrow.data_ptr(), seq_len.data_ptr(), node2row.data_ptr())
                plan.add("compact_index", lib.gcpx_compact_index, leave.data_ptr(), B, N, T, kept_idx.data_ptr())
                r2f = self._buf("row2frame", (B * T,), torch.int32)
                if self._head_grad_fused(key) or self._rows_direct(key):
                    plan.add("row2frame", lib.gcpx_index_inverse, node2row.data_ptr(), B * N, r2f.data_ptr(), B * T)
                if self._head_grad_fused(key):
                    # training forward with the likelihood gradient written by the head kernel (GCPX_HEAD_DLM_NLL_GRAD): the rows of
                    # the matched-frame gradient that no node maps to (padded frames) are zeroed here, early and on this side lane
                    S_ = hp.img_sz
                    dMD = self._buf("bw.dMD", (B * T, S_, S_, self._head_pitch))
                    plan.add("zero_unmapped", lib.gcpx_zero_unmapped_rows, dMD.data_ptr(), S_ * S_ * self._head_pitch, r2f.data_ptr(), B * T)

        # ---- run_encoder (base_gcp.py:184-213) ----
        enc_traj = inf_enc = None
        # three independent encoder passes (separate BatchNorm statistics, base_gcp.py:188,208,209) on three lanes
        plan.fork([1, 2])
        plan.lane = 1
        if "rng_all" in tin:
            # Gaussian.sample()'s numbers for this forward (+ the index draws' four per sequence): first needed by level 0's posterior
            plan.add("randn", lib.gcpx_randn, tin["rng_all"].data_ptr(), tin["rng_all"].numel(), self._buf("rng_state", (2,), torch.int64).data_ptr())
            if "aux_n" in tin and not pred_len:
                plan.wait(2, 1)             # lane 2's index draw reads them
        skips = self._plan_encoder(plan, "I0", tin["I_0"].data_ptr(), B, _addr(E), PS * nz, 0, 1)
        plan.lane = 2
        if not pred_len:
            if "aux_n" in tin:
                plan_aux_indices()          # read by the ground-truth cost below and by the heads behind the tree: not in front of the encoders
            plan_bookkeeping()
        if hp.attach_cost_mdl and hp.run_cost_mdl and train_aux and self._has_aux_training:
            # ground-truth cost of the cost model's sampled segment (cost_mdl.py:101-117, EuclideanPathLength): reads traj_seq and
            # two index vectors only, so it rides on this side lane instead of sitting in front of the decoder
            gt = self._buf("cost_target", (B,))
            rows = hp.input_nc * hp.img_sz
            plan.add("path_cost", lib.gcpx_path_cost, tin["traj_seq"].data_ptr(), tin["cost_start_idx"].data_ptr(),
                     tin["cost_end_idx"].data_ptr(), B, T, rows, hp.img_sz, self._buf("cost_partial", (B, rows)).data_ptr(), gt.data_ptr())
        self._plan_encoder(plan, "Ig", tin["I_g"].data_ptr(), B, _addr(E, 2 ** L * nz), PS * nz, 0, 1)
        outs = {}
        e0 = lambda: self._rowsrc(_addr(E), PS * nz, 0, nz)
        eg = lambda: self._rowsrc(_addr(E, 2 ** L * nz), PS * nz, 0, nz)
        if hp.regress_length and not pred_len:
            # get_end_ind's length predictor (misc.py:45-51) feeds only the loss and the outputs: it runs here, behind the two image
            # encoders and beside the trajectory encoder, instead of in front of the tree (20 us of the serial chain)
            plan.wait(2, 1)
            logits = self._buf("seq_len_logits", (B, T))
            self._mlp(plan, "length_pred", P["length_pred"], [e0(), eg()], B, 1, out=logits.data_ptr(), ob=T, orow=0)
            outs["seq_len_logits"] = logits
        plan.lane = 0
        if has_traj:
            enc_traj = self._buf("enc_traj", (B * T, nz))
            self._plan_encoder(plan, "traj", tin["traj_seq"].data_ptr(), B * T, enc_traj.data_ptr(), T * nz, nz, T)
            inf_enc = self._buf("inf_enc_seq", (B * T, nz))
            self._plan_seq_encoder(plan, "seq", "inf_encoder", enc_traj, inf_enc, B)
            if attentive:
                # attention keys: second temporal encoder + per-frame Linear (base_gcp.py:122-123, :200); then the key /
                # value projections of every level's attention in one batched launch each
                dk = hp.nz_attn_key
                n_mod = L if hp.untied_layers else 1
                kenc = self._buf("inf_key_enc", (B * T, nz))
                self._plan_seq_encoder(plan, "kseq", "inf_key_encoder.0", enc_traj, kenc, B)
                keys = self._buf("inf_enc_key_seq", (B * T, dk))
                dense = lambda t, w: self._rowsrc(t.data_ptr(), 0, w, w)
                self._gemm(plan, "kseq.key", [dense(kenc, nz)], B * T, dk, B * T, P["kseq.key.w"], P["kseq.key.b"],
                           out=keys.data_ptr(), ob=0, orow=dk)
                Kp = self._buf("attn.K", (n_mod, B * T, dk))
                Vp = self._buf("attn.V", (n_mod, B * T, nz))
                self._gemm(plan, "attn.k_proj", [dense(keys, dk)], B * T, dk, B * T, P["attn.k_proj.w"], P["attn.k_proj.b"],
                           out=Kp.data_ptr(), ob=0, orow=dk, batch=(n_mod, 0, P["attn.k_proj.w"][0].numel(), dk, B * T * dk))
                self._gemm(plan, "attn.v_proj", [dense(inf_enc, nz)], B * T, nz, B * T, P["attn.v_proj.w"], P["attn.v_proj.b"],
                           out=Vp.data_ptr(), ob=0, orow=nz, batch=(n_mod, 0, P["attn.v_proj.w"][0].numel(), nz, B * T * nz))
                plan.rec["attn_kv"] = dict(Kp=Kp, Vp=Vp, keys=keys, kenc=kenc, n_mod=n_mod)
        plan.join([1, 2])

        # ---- get_end_ind: length predictor (misc.py:45-51) ----
        if hp.regress_length and pred_len:
            logits = self._buf("seq_len_logits", (B, T))
            self._mlp(plan, "length_pred", P["length_pred"], [e0(), eg()], B, 1, out=logits.data_ptr(), ob=T, orow=0)
            outs["seq_len_logits"] = logits
            # get_end_ind under val_mode(pred_length=True) (base_gcp.py:219-226): the fed end_ind is REPLACED by a draw from the
            # length predictor, clamped to >= 2; the integer bookkeeping therefore follows the draw instead of riding on a side lane
            plan.add("sample_length", lib.gcpx_sample_length, logits.data_ptr(), tin["len_u"].data_ptr(), B, T, 2, tin["end_ind"].data_ptr())
            plan_bookkeeping()

        # ---- predict_sequence: level-serial tree (tree_utils.py:21-44, tree_module.py:67-114) ----
        side_merge = False                                  # the merge of the level being planned is already running on lane 1
        for l in range(L):
            W = P[f"tree{l if hp.untied_layers else 0}"]
            s = 2 ** (L - 1 - l)
            n = 2 ** l
            M = B * n
            nodeoff = lambda w: s * w                       # first node of this level inside a batch element
            el = lambda: self._rowsrc(_addr(E), PS * nz, 2 * s * nz, nz)
            er = lambda: self._rowsrc(_addr(E, 2 * s * nz), PS * nz, 2 * s * nz, nz)
            pz_out = (_addr(PZ, nodeoff(2 * nv)), PS * 2 * nv, 2 * s * 2 * nv)
            z_map = (_addr(Z, nodeoff(nv)), PS * nv, 2 * s * nv)
            nl = hp.n_lstm_layers
            merged = self._buf(f"merged{l}", (M, 2 * nl * H))

            def plan_merge(lv=l, group=None):
                # split_linear merge of the parents' hidden states of level lv (tree_lstm.py:43-48): all 2*n_lstm_layers
                # projections in one launch, blockIdx.z = projection index
                s_, n_ = 2 ** (L - 1 - lv), 2 ** lv
                Wl = P[f"tree{lv if hp.untied_layers else 0}"]
                mg = self._buf(f"merged{lv}", (B * n_, 2 * nl * H))
                if hp.tree_lstm == "sum":
                    # SumTree (tree_lstm.py:14-16): the parents' states added, no parameters
                    for side, mode in ((0, 0), (2 * s_ * SD, 1)):
                        plan.add(f"merge{lv}.{mode}", lib.gcpx_rows_strided, _addr(mg), n_ * SD, SD, _addr(Hid, side), PS * SD, 2 * s_ * SD,
                                 B, n_, SD, mode)
                    return
                if hp.tree_lstm == "linear":
                    # LinTree (tree_lstm.py:25-27): ONE Linear(2 SD -> SD) over both parents' whole states
                    h1 = self._rowsrc(_addr(Hid), PS * SD, 2 * s_ * SD, SD)
                    h2 = self._rowsrc(_addr(Hid, 2 * s_ * SD), PS * SD, 2 * s_ * SD, SD)
                    self._gemm(plan, f"merge{lv}", [h1, h2], B * n_, SD, n_, Wl["proj.w"], Wl["proj.b"], out=_addr(mg), ob=n_ * SD, orow=SD,
                               group=group)
                    return
                h1 = self._rowsrc(_addr(Hid), PS * SD, 2 * s_ * SD, H)
                h2 = self._rowsrc(_addr(Hid, 2 * s_ * SD), PS * SD, 2 * s_ * SD, H)
                self._gemm(plan, f"merge{lv}", [h1, h2], B * n_, H, n_, Wl["proj.w"], Wl["proj.b"], out=_addr(mg),
                           ob=n_ * 2 * nl * H, orow=2 * nl * H, batch=(2 * nl, H, Wl["proj.w"][0].numel(), H, H), group=group)

            # One lane for the whole level: the parent-state merge on a side lane (a parallel graph branch) bought nothing — a
            # cross-queue join costs ~10 us and the big levels are throughput-bound anyway (tools/fwd_tree_phase.py: level 6 323 us
            # with the side lane, 329 us in line).  Instead the merge of level l + 1, which needs nothing but the hidden states of
            # level l, shares the launch of level l's `out` Linear while both are in the small-M regime (gcpx_gemm_group).
            merge_with_predictors = not has_z and not sample_prior and hp.tree_lstm not in ("sum", "")
            if has_z:
                # given latents in depth-first order (tree.py:38); reparametrised with the learned prior (:79-82)
                g = (_addr(tin["z"], (s - 1) * nv), N * nv, 2 * s * nv) + z_map
                self._mlp(plan, f"prior{l}", W["prior"], [el(), er()], M, n, out=pz_out[0], ob=pz_out[1], orow=pz_out[2], gauss=g)
            elif sample_prior:
                g = (_addr(tin["eps"], (n - 1) * nv), N * nv, nv) + z_map
                self._mlp(plan, f"prior{l}", W["prior"], [el(), er()], M, n, out=pz_out[0], ob=pz_out[1], orow=pz_out[2], gauss=g)
            else:
                # the prior only feeds the KL term here: it shares the posterior's launch instead of a side lane of its own
                pq = []
                self._mlp(plan, f"prior{l}", W["prior"], [el(), er()], M, n, out=pz_out[0], ob=pz_out[1], orow=pz_out[2], group=pq)
                if attentive:
                    # AttentiveInference (attentive_inference.py:16-32): e_tilde = attention over the encoded sequence
                    et = self._plan_attention(plan, l, W, el(), er(), M, n, B, Kp, Vp, tin)
                else:
                    # posterior: gather inf_enc_seq at the node's matched timestep (inference.py:27-33)
                    et = self._rowsrc(inf_enc.data_ptr(), 0, nz, nz, rowidx=etrow[B * (n - 1):])
                g = (_addr(tin["eps"], (n - 1) * nv), N * nv, nv) + z_map
                self._mlp(plan, f"posterior{l}", W["q"], [el(), er(), et], M, n, out=_addr(QZ, nodeoff(2 * nv)),
                          ob=PS * 2 * nv, orow=2 * s * 2 * nv, gauss=g, group=pq)
                # the merge of this level's parent states needs level l - 1 only, like the two Predictors: same launch — or, at the wide
                # levels (a split-f16 GEMM of its own), a side lane started behind level l - 1's last LSTM layer (below)
                mg = []
                if l > 0 and merge_with_predictors and not side_merge:
                    plan_merge(l, group=mg)
                self._mlp_group(plan, f"prior+posterior{l}", pq, gemm=(mg[0] if mg else None))
                if side_merge:
                    plan.join([1])
                    side_merge = False
            zs = lambda: self._rowsrc(z_map[0], z_map[1], z_map[2], nv)
            if not hp.tree_lstm:
                # non-LSTM subgoal predictor (tree_module.py:109-110): e = tanh(Predictor([e_l, e_r, z (, e_0, e_g)])), no hidden state
                srcs = [el(), er(), zs()] + ([e0(), eg()] if hp.context_every_step else [])
                self._mlp(plan, f"subgoal{l}", W["sg"], srcs, M, n, out=_addr(E, nodeoff(nz)), ob=PS * nz, orow=2 * s * nz, tanh=True)
                continue
            if l == 0:
                if hp.lstm_init == "zero":
                    # ZeroLSTMCellInitializer (tree_lstm.py:68-70): both root parents start from zero states
                    for slot in (0, 2 ** L):
                        plan.add(f"lstm_init.zero{slot}", lib.gcpx_rows_strided, _addr(Hid, slot * SD), PS * SD, 0, self._zero_row(SD).data_ptr(),
                                 0, 0, B, 1, SD, 0)
                else:
                    # MLPLSTMCellInitializer (tree_module.py:104-105): (h_left, h_right) -> slots 0 and 2^L
                    self._mlp(plan, "lstm_init", W["init"], [el(), er(), zs()], M, n, out=_addr(Hid), ob=PS * SD, orow=0,
                              oblk=2 ** L * SD, out_split=SD)
                plan_merge()
            # input embedding of [e_l, e_r, z, e_0, e_g] (tree_module.py:97-101); inference plans fold it into LSTM layer 0
            x = self._buf(f"x{l}.0", (M, H))
            srcs = [el(), er(), zs()] + ([e0(), eg()] if hp.context_every_step else [])
            fused = "lstm0f.w" in W and not self.save_for_backward
            if not fused:
                self._gemm(plan, f"embed{l}", srcs, M, H, n, W["embed.w"], W["embed.b"], out=x.data_ptr(), ob=n * H, orow=H)
            for i in range(nl):
                xn = self._buf(f"x{l}.{i + 1}", (M, H))
                xs = self._rowsrc(x.data_ptr(), n * H, H, H)
                hs = self._rowsrc(_addr(merged, 2 * i * H), n * 2 * nl * H, 2 * nl * H, H)
                lstm = (_addr(merged, (2 * i + 1) * H), 2 * nl * H, _addr(Hid, nodeoff(SD) + 2 * i * H),
                        _addr(Hid, nodeoff(SD) + (2 * i + 1) * H), PS * SD, 2 * s * SD, xn.data_ptr())
                if i == 0 and fused:
                    self._gemm(plan, f"lstm{l}.0", srcs + [hs], M, 4 * H, n, W["lstm0f.w"], W["lstm0f.b"], epi=rt.EPI_LSTM, lstm=lstm)
                else:
                    self._gemm(plan, f"lstm{l}.{i}", [xs, hs], M, 4 * H, n, W[f"lstm{i}.w"], W[f"lstm{i}.b"],
                               epi=rt.EPI_LSTM, lstm=lstm)
                x = xn
            g = []
            self._gemm(plan, f"out{l}", [self._rowsrc(x.data_ptr(), n * H, H, H)], M, nz, n, W["out.w"], W["out.b"],
                       out=_addr(E, nodeoff(nz)), ob=PS * nz, orow=2 * s * nz, group=g)
            if l + 1 < L and not merge_with_predictors:
                plan_merge(l + 1, group=g)
            self._gemm_group(plan, f"out{l}+merge{l + 1}" if len(g) > 1 else f"out{l}", g)
            if (l + 1 < L and merge_with_predictors and hp.tree_lstm == "split_linear" and B * 2 ** (l + 1) >= getattr(self, "_merge_side_rows", 1 << 60)):
                # wide level ahead: its merge (a 33-43 us split GEMM of its own at 512 / 1024 rows) needs the hidden states just written and
                # nothing else — it runs on lane 1 beside the next level's prior + posterior instead of in front of them.  Forked BEHIND
                # `out`: started together, the merge's 768 workgroups starved the 5 us `out` GEMM for 36 us (profiles/r04f_fwd_trace.txt)
                plan.fork([1])
                plan.lane = 1
                plan_merge(l + 1)
                plan.lane = 0
                side_merge = True

        # ---- latent-space heads: independent of the decoder, run next to it on lane 1 ----
        F = B * N
        matching = adaptive and has_traj and phase == "train"        # soft-DTW binding is computed (tree.py:54-56)

        heads = []            # the latent-space heads are independent Predictors of one width: ONE grouped launch

        def plan_aux(idx, Wd):
            """run_auxilliary_models (base_gcp.py:234-262) on the pruned / matched latent sequence given by idx [B, Wd]"""
            mes = self._buf("model_enc_seq", (B, Wd, nz))
            plan.add("gather.model_enc_seq", lib.gcpx_gather_rows, E.data_ptr(), idx.data_ptr(), mes.data_ptr(), B, Wd, PS, 1, nz)
            outs["model_enc_seq_padded"] = mes
            if hp.attach_state_regressor:
                rs = self._buf("regressed_state", (B, Wd, hp.state_dim))
                self._mlp(plan, "state_regressor", P["state_regressor"], [self._rowsrc(mes.data_ptr(), Wd * nz, nz, nz)],
                          B * Wd, Wd, out=rs.data_ptr(), ob=Wd * hp.state_dim, orow=hp.state_dim, group=heads)
                outs["regressed_state_padded"] = rs
            if hp.attach_inv_mdl and phase == "train" and (sample_prior or hp.train_inv_mdl_full_seq or not has_traj):
                # InverseModel.full_seq_forward (inverse_mdl.py:110-134): val_mode sets _inv_mdl_full_seq (base_gcp.py:44-53,250)
                act = self._buf("actions", (B, Wd - 1, hp.n_actions))
                first = enc_traj if has_traj else mes
                s0 = self._rowsrc(first.data_ptr(), (T if has_traj else Wd) * nz, nz, nz)
                s1 = self._rowsrc(_addr(mes, nz), Wd * nz, nz, nz)
                self._mlp(plan, "inv_mdl", P["inv_mdl"], [s0, s1], B * (Wd - 1), Wd - 1, out=act.data_ptr(),
                          ob=(Wd - 1) * hp.n_actions, orow=hp.n_actions, group=heads)
                outs["actions_padded"] = act
            aux_rows = None
            if train_aux and ((hp.attach_inv_mdl and not hp.train_inv_mdl_full_seq) or (hp.attach_cost_mdl and hp.run_cost_mdl)):
                aux_rows = self._buf("aux_rows", (4, B), torch.int32)
                plan.add("aux_index_rows", lib.gcpx_aux_index_rows, tin["inv_t0"].data_ptr(), tin["inv_t1"].data_ptr(),
                         tin["cost_start_idx"].data_ptr(), tin["cost_end_idx"].data_ptr(), B, T, Wd, aux_rows.data_ptr())
                gather = lambda t, i: self._rowsrc(t.data_ptr(), 0, nz, nz, rowidx=aux_rows[i])
            if hp.attach_inv_mdl and train_aux and not hp.train_inv_mdl_full_seq:
                # InverseModel.forward on ONE sampled frame pair per sequence (inverse_mdl.py:136-178): first frame from the encoder
                # (train_im0_enc), second from the model's matched latents; both detached, so only action_pred is trained
                act = self._buf("actions_sampled", (B, hp.n_actions))
                self._mlp(plan, "inv_mdl", P["inv_mdl"], [gather(enc_traj, 0), gather(mes, 1)], B, B, out=act.data_ptr(), ob=0,
                          orow=hp.n_actions, group=heads)
                outs["actions_sampled"] = act
            if hp.attach_cost_mdl and hp.run_cost_mdl and train_aux:
                # CostModel.forward (cost_mdl.py:42-57): cost_pred on a sampled (start, end) pair of the matched latents against the
                # ground-truth path cost of the same segment of traj_seq (_general_cost with EuclideanPathLength, conf.py:35-37)
                cost = self._buf("cost_pred", (B, 1))
                self._mlp(plan, "cost_mdl", P["cost_mdl"], [gather(mes, 2), gather(mes, 3)], B, B, out=cost.data_ptr(), ob=0, orow=1,
                          group=heads)
                outs["cost_pred"], outs["cost_target"] = cost, self._buf("cost_target", (B,))     # filled on lane 2 (see above)

        # The latent-space heads are ~60 us of small launches.  Beside the decoder blocks (persistent grids, two workgroups per
        # CU) they cost more than that in interference (pyramid-2: 317 us beside them, 200 us alone), so they run in front.
        heads_lane = 0
        if heads_lane:
            plan.fork([1])
        plan.lane = heads_lane
        if adaptive:
            # learned pruning (adaptive.py:62-77): distance predictor on consecutive depth-first latents
            dist = self._buf("distances", (B, N - 1))
            self._mlp(plan, "distance", P["distance"], [self._rowsrc(_addr(E, nz), PS * nz, nz, nz),
                                                        self._rowsrc(_addr(E, 2 * nz), PS * nz, nz, nz)],
                      B * (N - 1), N - 1, out=dist.data_ptr(), ob=N - 1, orow=1)
            pruned_len = self._buf("pruned_len", (B,), torch.int32)
            plan.add("distance_prune", lib.gcpx_distance_prune, dist.data_ptr(), C.c_float(hp.learned_pruning_threshold), None, B, N,
                     leave.data_ptr(), kept_idx.data_ptr(), pruned_len.data_ptr(), None)
            outs["distances"], outs["pruned_len"] = dist, pruned_len
            if not matching:
                plan_aux(kept_idx, N)                    # get_predicted_pruned_seqs (tree.py:69-70)
                outs["aux_len"] = pruned_len
        else:
            plan_aux(kept_idx, T)
            # existence predictor over depth-first latents (frame_binding.py:67-78)
            exist = self._buf("existence", (B, N))
            self._mlp(plan, "existence", P["existence"], [self._rowsrc(_addr(E, nz), PS * nz, nz, nz)], F, N,
                      out=exist.data_ptr(), ob=N, orow=1, group=heads)
            outs["existence"] = exist
        if heads:
            self._mlp_group(plan, "heads", heads)
            heads.clear()
        plan.lane = 0

        def kl_args(kl_b, batch=()):
            return (_addr(QZ, 2 * nv), _addr(PZ, 2 * nv)) + batch + (N, nv, PS * 2 * nv, 2 * nv, C.c_float(hp.free_nats), None, 0, kl_b.data_ptr())

        def loss_args():
            """gcpx_loss_args of this forward (base_gcp.py:264-304, tree_module.py:116-157)"""
            kl_b = self._buf("kl_b", (B,))
            la = rt.LossArgs()
            la.nll_bt, la.pad_mask, la.kl_b = self._buf("nll_bt", (B, T)).data_ptr(), tin["pad_mask"].data_ptr(), kl_b.data_ptr()
            la.len_logits = outs["seq_len_logits"].data_ptr() if "seq_len_logits" in outs else None
            la.end_ind = tin["end_ind"].data_ptr()
            if adaptive:     # BCE of the learned-pruning logits against "same best frame" (adaptive.py:118-122), N - 1 pairs
                la.existence, la.leave = outs["distances"].data_ptr(), outs["distance_target"].data_ptr()
            else:
                la.existence, la.leave = outs["existence"].data_ptr(), leave.data_ptr()
            if "regressed_state_padded" in outs and "traj_seq_states" in tin:
                la.regressed_state, la.state_target = outs["regressed_state_padded"].data_ptr(), tin["traj_seq_states"].data_ptr()
            la.seq_len = seq_len.data_ptr()
            if "actions_sampled" in outs and "actions" in tin:          # inverse_mdl.py:181-191
                la.action_pred, la.action_seq, la.inv_t0 = outs["actions_sampled"].data_ptr(), tin["actions"].data_ptr(), tin["inv_t0"].data_ptr()
                la.n_actions, la.w_action = hp.n_actions, hp.action_rec_weight
            if "cost_pred" in outs:                                     # cost_mdl.py:59-62
                la.cost_pred, la.cost_target, la.w_cost = outs["cost_pred"].data_ptr(), outs["cost_target"].data_ptr(), 1.0
            loss_out = self._buf("losses", (16,), zero=True)
            la.out, la.B, la.T, la.N, la.state_dim = loss_out.data_ptr(), B, T, (N - 1 if adaptive else N), hp.state_dim
            la.w_rec, la.w_kl, la.w_len, la.w_exist, la.w_state = hp.dense_img_rec_weight, hp.kl_weight, hp.length_pred_weight, 1.0, 1.0
            if self._kl_w is not None:                   # burn-in schedule: the current weight is read from device memory
                la.w_kl_dev = self._kl_w.data_ptr()
            la.total_div = float(T * hp.input_nc * hp.img_sz * hp.img_sz)
            plan.keep.append(la)
            return la, kl_b

        decode, with_loss = key[8], key[7]
        # Everything of the loss that needs no decoded frame — the KL and the latent-side terms — goes in front of the decoder in one
        # launch (gcpx_loss_pre); behind the head only the reconstruction sum and the total remain (gcpx_loss_final).  (The adaptive
        # model's pruning target comes out of the soft-DTW matching of decoded frames: it keeps the single combine at the end.)
        loss_pre = None
        if with_loss and not adaptive:
            loss_pre = loss_args()
            plan.add("loss.pre", lib.gcpx_loss_pre, C.byref(loss_pre[0]), *kl_args(loss_pre[1]))
        if decode:
            # ---- dense_rec: decode every node (tree_dense_rec.py:41-44) ----
            F = B * N
            S = hp.img_sz
            prev = self._plan_decoder_features(plan, self._rowsrc(_addr(E, nz), PS * nz, nz, nz), F, N, skips)
            images = self._buf("images_df", (B, N, hp.input_nc, S, S))
            distr = matched_distr = None
            with_loss = key[7]
            dlm = hp.decoder_distribution == "discrete_logistic_mixture"
            head_out, row_map = None, None
            fused_nll = None
            if dlm:
                mode = rt.HEAD_DLM_MEAN
                if self.materialize_distr or (adaptive and self.save_for_backward and with_loss):
                    # (adaptive training: the backward of the mixture mean needs the raw parameters of every node)
                    mode, distr = rt.HEAD_DLM_BOTH, self._buf("distr_df", (B, N, S, S, self._head_pitch))
                    head_out = distr
                elif self._head_grad_fused(key):
                    # training forward: likelihood AND its gradient w.r.t. the parameters in the head's epilogue
                    # (GCPX_HEAD_DLM_NLL_GRAD): the parameters themselves are never stored, gcpx_dlm_nll_bwd's pass over them is gone
                    mode, row_map = rt.HEAD_DLM_NLL_GRAD, node2row
                    head_out = self._buf("bw.dMD", (B * T, S, S, self._head_pitch))
                    fused_nll = self._buf("nll_partial", ((S // 4) * (S // 16), B * T), zero=True)
                    plan.rec["nll_bwd_fused"] = plan.rec["head_grad_fused"] = True
                elif with_loss and not adaptive and not self.save_for_backward and self._head_nll_fusable():
                    # forward with losses, no backward to follow: the likelihood of the matched frames is evaluated in the head's
                    # epilogue (GCPX_HEAD_DLM_NLL) — their 2.35 GB of raw parameters (c2) are neither written nor read back
                    mode, row_map = rt.HEAD_DLM_NLL, node2row
                    fused_nll = self._buf("nll_partial", ((S // 4) * (S // 16), B * T), zero=True)
                elif with_loss and not adaptive:
                    # only the nodes matched to a ground-truth frame keep their distribution parameters
                    # (frame_binding.py:91-92): row b*T+t of matched_distr <- node matched to frame t
                    mode, matched_distr = rt.HEAD_DLM_BOTH, self._buf("matched_distr", (B, T, S, S, self._head_pitch))
                    head_out, row_map = matched_distr, node2row
            else:
                mode = rt.HEAD_TANH_NCHW
            a = self._conv_args([prev], F, S, S, S, S, hp.head_channels, self._head_pitch, P["dec.head.w"], P["dec.head.b"],
                                head_out, upsample=0, head_mode=mode, images=images)
            # the split-f16 mixture head stores the frames node2row maps to a row a second time, in sequence order: the matched /
            # kept frames (tree_dense_rec.py:56-60, tree.py:62-65) need no gather pass over the decoded frames afterwards
            rows_direct = self._rows_direct(key) and (row_map is not None or mode == rt.HEAD_DLM_MEAN)
            rows_images = None
            if rows_direct:
                row_map = node2row
                want_matched = has_traj and phase == "train"
                rows_images = self._buf("rows_images", (2 if want_matched else 1, B, T, hp.input_nc, S, S))
                a.images_rows = rows_images.data_ptr()
                a.images_rows_dup = rows_images[0].numel() if want_matched else 0
            a.raw_row_map = row_map.data_ptr() if row_map is not None else None
            if fused_nll is not None:
                a.nll_target, a.nll_partial, a.nll_rows = tin["traj_seq"].data_ptr(), fused_nll.data_ptr(), B * T
                if mode == rt.HEAD_DLM_NLL_GRAD:
                    # d total / d nll_bt = w_rec * pad_mask / (B * prod(traj_seq.shape[1:])) (base_gcp.py:299-301)
                    a.nll_row_weight = tin["pad_mask"].data_ptr()
                    a.nll_scale = hp.dense_img_rec_weight / (B * float(T * hp.input_nc * S * S))
            self._set_split(a, "dec.head")
            plan.keep.append(a)
            if heads_lane:
                plan.join([1])       # the latent-space heads overlapped the decoder blocks; the head runs alone
            plan.add("dec.head", lib.gcpx_conv3x3, C.byref(a))
            outs["images_df"], outs["distr_df_kernel_order"] = images, distr

            # ---- pruning / matching gathers of decoded frames ----
            row = hp.input_nc * S * S
            if matching:
                # AdaptiveBinding.get_w (adaptive.py:32-60): image cost matrix -> soft-DTW posterior over alignments -> w
                ns = lib.gcpx_cdist_splits(row)
                dsum = self._buf("cdist.dsum", (B, N, T))
                plan.add("cdist", lib.gcpx_cdist, images.data_ptr(), tin["traj_seq"].data_ptr(), B, N, T, row,
                         self._buf("cdist.part", (ns, B, N, T)).data_ptr(), self._buf("cdist.xn", (B * N,)).data_ptr(),
                         self._buf("cdist.yn", (B * T,)).data_ptr(), dsum.data_ptr())
                wdf = self._buf("match_dist_df", (B, N, T))
                temp = self.sd["tree_module.tree_modules.0.binding.temp"]
                plan.add("soft_dtw", lib.gcpx_soft_dtw, dsum.data_ptr(), C.c_float(float(row)), temp.data_ptr(), tin["end_ind"].data_ptr(),
                         B, N, T, self._buf("dtw.acc", (2 * B, N, T), torch.float64).data_ptr(), wdf.data_ptr())
                matched_idx = self._buf("matched_idx", (B, T), torch.int32)
                best_t = self._buf("best_t", (B, N), torch.int32)
                entropy, p_n = self._buf("entropy", (B, N)), self._buf("p_n", (B, N))
                plan.add("match_stats", lib.gcpx_match_stats, wdf.data_ptr(), tin["end_ind"].data_ptr(), B, L, T, f2n.data_ptr(),
                         matched_idx.data_ptr(), best_t.data_ptr(), entropy.data_ptr(), p_n.data_ptr())
                dist_tgt = self._buf("distance_target", (B, N - 1), torch.int32)
                plan.add("distance_target", lib.gcpx_distance_prune, outs["distances"].data_ptr(),
                         C.c_float(hp.learned_pruning_threshold), best_t.data_ptr(), B, N, leave.data_ptr(), kept_idx.data_ptr(),
                         outs["pruned_len"].data_ptr(), dist_tgt.data_ptr())
                plan.add("seq_len", lib.gcpx_seq_index, tin["end_ind"].data_ptr(), B, T, self._buf("seq_idx", (B, T), torch.int32).data_ptr(),
                         seq_len.data_ptr())
                plan_aux(matched_idx, T)                     # get_matched_pruned_seqs for 'dtw' (base_gcp.py:358-366)
                self._mlp_group(plan, "heads", heads)
                heads.clear()
                ent_sum = self._buf("entropy_sum", (1,))
                plan.add("entropy_sum", lib.gcpx_reduce_partials, entropy.data_ptr(), B * N, 1, 1, ent_sum.data_ptr(), 0)
                outs["entropy_sum"] = ent_sum
                outs.update(cdist_sum=dsum, match_dist_df=wdf, matched_idx=matched_idx, best_t=best_t, entropy_df=entropy, p_n_df=p_n,
                            distance_target=dist_tgt, aux_len=seq_len)
            elif has_traj and phase == "train":
                if rows_images is not None:
                    # (what the head has not written: the padded frames, which argmax over an all-zero column matches to the root)
                    matched = rows_images[1]
                    plan.add("gather.matched.rest", lib.gcpx_gather_rows_rest, images.data_ptr(), f2n.data_ptr(), matched.data_ptr(), B, T, N,
                             0, row, self._buf("row2frame", (B * T,), torch.int32).data_ptr())
                else:
                    matched = self._buf("matched_images", (B, T, hp.input_nc, S, S))
                    plan.add("gather.matched", lib.gcpx_gather_rows, images.data_ptr(), f2n.data_ptr(), matched.data_ptr(), B, T, N,
                             0, row)
                outs["soft_matched_estimates"] = matched
            Wp = N if adaptive else T
            if rows_images is not None:
                pruned = rows_images[0]                      # (rows beyond the sequence: zeros)
                plan.add("gather.pruned.rest", lib.gcpx_gather_rows_rest, images.data_ptr(), kept_idx.data_ptr(), pruned.data_ptr(), B, Wp, N,
                         0, row, self._buf("row2frame", (B * T,), torch.int32).data_ptr())
            else:
                pruned = self._buf("pruned_images", (B, Wp, hp.input_nc, S, S))
                plan.add("gather.pruned", lib.gcpx_gather_rows, images.data_ptr(), kept_idx.data_ptr(), pruned.data_ptr(), B, Wp, N, 0,
                         row)
            outs["pruned_padded"] = pruned

        # ---- losses (base_gcp.py:264-304, tree_module.py:116-157) ----
        if with_loss:
            nll_bt = self._buf("nll_bt", (B, T))
            if adaptive:
                # LossAveragingCriterion.loss (binding_loss.py:19-42)
                plan.add("loss.averaging_nll", lib.gcpx_averaging_nll, dsum.data_ptr(), wdf.data_ptr(),
                         self.sd["decoder.log_sigma"].data_ptr(), C.c_float(float(row)), B, N, T, nll_bt.data_ptr())
            elif dlm and fused_nll is not None:
                # rows of padded frames (t > end_ind) are written by no node: they keep whatever an earlier call left (finite) and
                # carry pad_mask 0 in the combination below
                plan.add("loss.nll_reduce", lib.gcpx_reduce_partials, fused_nll.data_ptr(), fused_nll.shape[0], B * T, B * T,
                         nll_bt.data_ptr(), 0)
            elif dlm:
                if matched_distr is None:       # materialize_distr: gather the matched rows out of the full tensor
                    matched_distr = self._buf("matched_distr", (B, T, S, S, self._head_pitch))
                    plan.add("gather.matched_distr", lib.gcpx_gather_rows, distr.data_ptr(), f2n.data_ptr(),
                             matched_distr.data_ptr(), B, T, N, 0, S * S * self._head_pitch)
                if self.save_for_backward:
                    # training step: loss and its gradient w.r.t. the matched parameters in one pass (the backward plan reuses
                    # dMD); d total / d nll_bt = w_rec * pad_mask / (B * prod(traj_seq.shape[1:])) (base_gcp.py:299-301)
                    dMD = self._buf("bw.dMD", (B * T, S, S, self._head_pitch))
                    div = float(T * hp.input_nc * S * S)
                    plan.add("loss.dlm_nll+bwd", lib.gcpx_dlm_nll_bwd, matched_distr.data_ptr(), tin["traj_seq"].data_ptr(),
                             tin["pad_mask"].data_ptr(), C.c_float(hp.dense_img_rec_weight / (B * div)), dMD.data_ptr(),
                             self._buf("bw.dMD.colsum", (B * T, self._head_pitch)).data_ptr(), nll_bt.data_ptr(), B * T, S * S,
                             self._head_pitch, hp.n_mixtures)
                    plan.rec["nll_bwd_fused"] = True
                else:
                    plan.add("loss.dlm_nll", lib.gcpx_dlm_nll, matched_distr.data_ptr(), tin["traj_seq"].data_ptr(),
                             tin["pad_mask"].data_ptr(), nll_bt.data_ptr(), B * T, S * S, self._head_pitch, hp.n_mixtures)
            else:
                plan.add("loss.gauss_nll", lib.gcpx_gauss_nll, outs["soft_matched_estimates"].data_ptr(),
                         tin["traj_seq"].data_ptr(), self.sd["decoder.log_sigma"].data_ptr(), nll_bt.data_ptr(), B * T,
                         hp.input_nc * S * S)
            if loss_pre is None:
                la, kl_b = loss_args()
                plan.add("loss.kl", lib.gcpx_kl_gauss, *kl_args(kl_b, (B,)))
                plan.add("loss.combine", lib.gcpx_loss_combine, C.byref(la))
            else:
                la, kl_b = loss_pre
                plan.add("loss.final", lib.gcpx_loss_final, C.byref(la))
            loss_out = self._buf("losses", (16,), zero=True)
            outs["losses"], outs["nll_bt"], outs["kl_b"] = loss_out, nll_bt, kl_b
            outs["matched_distr_kernel_order"] = matched_distr

        outs.update(E=E, Hid=Hid, Z=Z, PZ=PZ, QZ=QZ, node_t=node_t, leave=leave, frame2node=f2n, seq_len=seq_len,
                    kept_idx=kept_idx, enc_traj_seq=enc_traj, inf_enc_seq=inf_enc, node2row=node2row, etilde_row=etrow)
        plan.rec.update(head_src=(prev if decode else None), tin=tin, key=key)
        if with_loss:
            plan.rec["loss_args"] = la
        plan.outs = outs
        return plan

    # ------------------------------------------------------------------------------------------------
    # forward
    def _sync_rng_state(self):
        """{seed, offset} of the plan's generator on the device.  The seed follows torch's CUDA generator (what the torch draw used:
        `torch.cuda.manual_seed` — per rank in train.py — keeps its meaning); re-seeding torch restarts the stream."""
        seed = int(torch.cuda.initial_seed()) & ((1 << 63) - 1)
        if getattr(self, "_rng_seed", None) != seed:
            self._rng_seed = seed
            st = self._buf("rng_state", (2,), torch.int64)
            st.copy_(torch.tensor([seed, 0], dtype=torch.int64), non_blocking=True)

    # ------------------------------------------------------------------------------------------------
    def forward(self, inputs, phase="train", noise=None):
        """BaseGCPModel.forward (base_gcp.py:140-161).

        inputs: dict with I_0, I_g [B,3,H,W], end_ind int64 [B]; optional traj_seq [B,T,3,H,W], z [B,N,nz_vae]
        (depth-first node order).  `noise` [B,N,nz_vae] (breadth-first node order) replaces the RNG draws of
        Gaussian.sample(); when None it is drawn with torch.randn on the device.
        """
        hp = self._hp
        B = inputs["I_0"].shape[0]
        has_traj = "traj_seq" in inputs and not self._sample_prior
        has_z = "z" in inputs
        if not has_traj and not has_z and not self._sample_prior and not hp.deterministic:
            raise ValueError("posterior path needs traj_seq (or use val_mode() / feed z)")
        # get_end_ind (base_gcp.py:215-229): under val_mode(pred_length=True) the length is drawn from the length predictor whenever
        # its loss is trained (or no end_ind is fed); otherwise the fed end_ind is used
        pred_len = bool(self._has_pred_length and self._use_pred_length and hp.regress_length and
                        (hp.length_pred_weight > 0 or "end_ind" not in inputs))
        if "end_ind" not in inputs and not pred_len:
            raise ValueError("end_ind must be fed unless val_mode(pred_length=True) draws it from the length predictor")
        with_loss = has_traj and phase == "train" and "pad_mask" in inputs
        train_aux = has_traj and phase == "train"
        need_idx = self._has_aux_training and train_aux and ((hp.attach_inv_mdl and not hp.train_inv_mdl_full_seq) or
                                                             (hp.attach_cost_mdl and hp.run_cost_mdl))
        AUX = ("inv_t0", "inv_t1", "cost_start_idx", "cost_end_idx")
        fed_idx = need_idx and all(k in inputs for k in AUX)
        opt = tuple(k for k in ("pad_mask", "traj_seq_states", "w0", "actions") if with_loss and k in inputs)
        if hp.action_conditioned_pred and "actions" not in opt:
            opt += ("actions",)                      # the action-conditioned predictor reads them on every path (sequential.py:45-47)
        if not self._decode and (with_loss or has_traj):
            raise ValueError("decode=False is the planner's prior / given-z path: no ground-truth sequence, no losses")
        # inputs are copied into persistent buffers (one D2D copy; 63 MB for traj_seq at c2 = ~25 us) so that the
        # captured graph — which bakes in device pointers — stays valid whatever tensors the caller passes
        # The copies are enqueued on the MODEL's stream (ordered behind the caller's stream by one event), so the launch that
        # follows needs no second cross-stream hand-over before its first kernel.
        caller = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(caller)
        tin = {}
        # under pred_len the fed end_ind is replaced by the draw (base_gcp.py:219-226): it is not read, and the draw goes to a buffer
        # of its own so that a caller who filled input_buffer('end_ind') in place keeps its ground-truth lengths
        names = ("I_0", "I_g") + (("end_ind",) if ("end_ind" in inputs and not pred_len) else ()) + (("traj_seq",) if has_traj else ()) + \
            (("z",) if has_z else ()) + opt + (AUX if fed_idx else ())
        with torch.cuda.stream(self._stream):
            for k in names:
                t = inputs[k]
                want = torch.int64 if (k == "end_ind" or k in AUX) else torch.float32
                buf = self._buf("in." + k, tuple(t.shape), want)
                if not (t.is_cuda and t.data_ptr() == buf.data_ptr() and t.dtype == want):
                    # (a caller that fills `input_buffer(k, shape)` directly — a loader writing its batch in place — skips the copy)
                    buf.copy_(t, non_blocking=True)
                    if t.is_cuda:
                        t.record_stream(self._stream)
                tin[k] = buf
            if "end_ind" not in tin:
                tin["end_ind"] = self._buf("out.end_ind", (B,), torch.int64)      # written by the length draw inside the plan
            if pred_len:
                # the OneHotCategorical draw of the sequence length (misc.py:49) as one uniform number per sequence
                lu = self._buf("in.len_u", (B,))
                if "len_u" in inputs:
                    lu.copy_(inputs["len_u"], non_blocking=True)
                else:
                    lu.uniform_()
                tin["len_u"] = lu
            # One generator launch per call: the latent noise of Gaussian.sample() and the four numbers per sequence behind the
            # inverse / cost model index draws (InverseModel.sample_offsets / CostModel._general_cost draw with np.random on the host,
            # inverse_mdl.py:84-104, cost_mdl.py:105-107) share one buffer [noise | 4 B numbers]; the index kernel reads the
            # latter as standard-normal draws (u = Phi(n)) and is an op of the plan, i.e. inside the graph.
            n_eps = 0 if has_z else B * self._n_latents() * hp.nz_vae
            draw_idx = need_idx and not fed_idx
            rng = self._buf("rng", (n_eps + (4 * B if draw_idx else 0),)) if (n_eps or draw_idx) else None
            # drawn by the plan itself (a side-lane op of the graph, off the encoder chain) when nothing is fed
            in_plan = bool(self._rng_in_plan and rng is not None and noise is None and os.environ.get("GCPX_TORCH_RNG") is None and
                           not (pred_len and draw_idx))       # (there the index draw sits in front of the fork: keep the torch draw)
            if draw_idx:
                tin["aux_n"] = rng[n_eps:].view(4, B)
                for k in AUX:
                    tin[k] = self._buf("in." + k, (B,), torch.int64)
            if not has_z and n_eps:
                # the draws of Gaussian.sample() live in a persistent buffer as well
                eps = rng[:n_eps].view(B, self._n_latents(), hp.nz_vae)
                if noise is None:
                    if not in_plan:
                        rng.normal_()
                else:
                    if not (noise.is_cuda and noise.data_ptr() == eps.data_ptr()):
                        eps.copy_(noise)
                        if noise.is_cuda:
                            noise.record_stream(self._stream)
                    if draw_idx:
                        tin["aux_n"].normal_()
                tin["eps"] = eps
            elif draw_idx and not in_plan:
                rng.normal_()                        # (z is fed, or the predictor is deterministic: only the index draws)
            if in_plan:
                tin["rng_all"] = rng
                self._sync_rng_state()
        # the plan (and its captured graph) bakes in buffer addresses and sizes: everything that selects buffers is part of the key
        shapes = tuple((k, tuple(tin[k].shape)) for k in sorted(tin))
        key = (B, has_traj, has_z, self._sample_prior, phase, self.training, self.materialize_distr, with_loss, self._decode, pred_len,
               shapes)             # (a plan that draws its own noise has "rng_all" among its inputs: part of `shapes`)
        if key not in self._plans:
            plan = self._build_plan(key, tin)
            plan.keep.append(tin)
            self._plans[key] = (None, plan)
        plan = self._plans[key][1]
        stream = self._stream.cuda_stream
        if self._timed_op is not None:
            self._run_timed(plan, stream)
        elif self.use_graph:
            if plan.graph is None:
                plan.run(self._streams)               # warm-up (sets kernel attributes) outside capture
                plan.graph = self._capture(plan, plan.ops, stream)
                plan.eager = self.use_graph == "auto" and self._eager_replays_faster(plan, stream)
            if plan.eager:
                plan.run(self._streams)
            else:
                rt.check(self.lib.gcpx_graph_launch(plan.graph, stream), "graph_launch")
        else:
            plan.run(self._streams)
        caller.wait_stream(self._stream)
        return self._wrap_outputs(plan.outs, tin, phase)

    def _eager_replays_faster(self, plan, stream, reps=4, trials=3):
        """time `reps` consecutive replays of the plan as a hipGraph and as eager launches — each replay between the same two stream
        hand-overs a forward() call makes (caller -> model stream -> caller), host enqueue included: what a caller's loop pays —,
        `trials` times in turn, and say whether the best eager time beats the best graph time by more than 2 %.  (Timed WITHOUT the
        hand-overs, back-to-back graph launches pipeline into each other and look 0.2 ms faster per forward than they are inside a
        loop of forward() calls.)  One-time cost per plan: 2 x trials x (reps + 1) forwards."""
        import time
        caller = torch.cuda.current_stream(self.device)

        def one(eager):
            self._stream.wait_stream(caller)
            if eager:
                plan.run(self._streams)
            else:
                rt.check(self.lib.gcpx_graph_launch(plan.graph, stream), "graph_launch")
            caller.wait_stream(self._stream)

        def timed(eager):
            one(eager)
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(reps):
                one(eager)
            torch.cuda.synchronize(self.device)
            return time.perf_counter() - t0
        tg = te = float("inf")
        for _ in range(trials):
            tg = min(tg, timed(False))
            te = min(te, timed(True))
        plan.tuned = (tg / reps, te / reps)
        return te < 0.98 * tg

    def replay_info(self):
        """How the plan of the latest forward() is replayed and what the one-time comparison measured (bench.py reports it):
        {"mode": "graph" | "eager", "policy": the GCPX_FORWARD_REPLAY setting, "tuned_ms": {"graph", "eager"} or None}."""
        plan = [v[1] for v in self._plans.values()][-1]
        tuned = getattr(plan, "tuned", None)
        return {"mode": "eager" if (plan.eager or not self.use_graph) else "graph",
                "policy": {True: "graph", False: "eager"}.get(self.use_graph, "auto"),
                "tuned_ms": None if tuned is None else {"graph": round(1e3 * tuned[0], 4), "eager": round(1e3 * tuned[1], 4)}}

    def force_replay(self, mode):
        """Replay every existing plan as a hipGraph ("graph"), by eager launches ("eager") or as the one-time comparison chose ("auto",
        re-measured on the next call).  Measurement aid (bench.py's also.forward_graph / forward_eager legs)."""
        self.use_graph = {"graph": True, "eager": False, "auto": "auto"}[mode]
        for _, plan in self._plans.values():
            if mode == "auto":
                plan.graph = None
            plan.eager = mode == "eager"

    def _capture(self, plan, ops, stream):
        rt.check(self.lib.gcpx_graph_begin(stream), "graph_begin")
        plan.run(self._streams, ops)
        g = C.c_void_p()
        rt.check(self.lib.gcpx_graph_end(stream, C.byref(g)), "graph_end")
        return g

    # ---- one op of the plan bracketed by HIP events on the launch stream (roofline measurement) ----
    def set_timed_op(self, name):
        self._timed_op = name
        self._timed_events = []

    def _run_timed(self, plan, stream):
        names = [op[0] for op in plan.ops]
        i = names.index(self._timed_op)
        e0, e1 = C.c_void_p(), C.c_void_p()
        rt.check(self.lib.gcpx_event_create(C.byref(e0)), "event_create")
        rt.check(self.lib.gcpx_event_create(C.byref(e1)), "event_create")
        if self.use_graph == "auto" and plan.graph is None:
            plan.run(self._streams)
            plan.graph = self._capture(plan, plan.ops, stream)
            plan.eager = self._eager_replays_faster(plan, stream)
        if plan.eager or not self.use_graph:
            # the plan as it is replayed (eager launches over the lanes), with an event on the main lane in front of and behind the op
            if getattr(plan, "timed_ops", None) is None or plan.timed_ops[0] != self._timed_op:
                assert plan.ops[i][3] == 0, "the timed op must be on the main lane"
                plan.timed_ops = (self._timed_op, plan.ops[:i] + [("@mark", None, ("timed", 0), 0), plan.ops[i], ("@mark", None, ("timed", 1), 0)] +
                                  plan.ops[i + 1:])
            plan.run(self._streams, ops=plan.timed_ops[1],
                     on_mark=lambda tag, k: rt.check(self.lib.gcpx_event_record(e1 if k else e0, stream), "event_record") if tag == "timed" else None)
            self._timed_events.append((e0, e1))
            return
        if getattr(plan, "split", None) is None:
            plan.run(self._streams)
            assert plan.ops[i][3] == 0, "the timed op must be on the main lane"
            plan.split = (self._capture(plan, plan.ops[:i], stream), self._capture(plan, plan.ops[i + 1:], stream))
        rt.check(self.lib.gcpx_graph_launch(plan.split[0], stream), "graph_launch")
        rt.check(self.lib.gcpx_event_record(e0, stream), "event_record")
        name, fn, args, _ = plan.ops[i]
        rt.check(fn(*args, stream), name)
        rt.check(self.lib.gcpx_event_record(e1, stream), "event_record")
        rt.check(self.lib.gcpx_graph_launch(plan.split[1], stream), "graph_launch")
        self._timed_events.append((e0, e1))

    def profile_ops(self, inputs, phase="train", noise=None, repeats=5):
        """Per-op device time of the current plan (eager launches bracketed by events): [(name, microseconds)].
        Tuning aid; not used on the hot path."""
        self.forward(inputs, phase, noise)
        torch.cuda.synchronize()
        plan = [v[1] for v in self._plans.values()][-1]
        res = []
        with torch.cuda.stream(self._stream):
            for name, fn, args, _ in plan.ops:
                if name.startswith("@"):
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                rt.check(fn(*args, self._stream.cuda_stream), name)
                e0.record(self._stream)
                for _ in range(repeats):
                    rt.check(fn(*args, self._stream.cuda_stream), name)
                e1.record(self._stream)
                self._stream.synchronize()
                res.append((name, 1e3 * e0.elapsed_time(e1) / repeats))
        return res

    def timed_op_ms(self):
        """Durations (ms) of the timed op for every forward since set_timed_op(); synchronises."""
        out = []
        for e0, e1 in self._timed_events:
            ms = C.c_float()
            rt.check(self.lib.gcpx_event_elapsed_ms(e0, e1, C.byref(ms)), "event_elapsed")
            out.append(ms.value)
            self.lib.gcpx_event_destroy(e0)
            self.lib.gcpx_event_destroy(e1)
        self._timed_events = []
        return out

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
