"""Training step of the gcp_tree model on MI355X: forward + explicit backward + RAdam, all in HIP.

Mirrors the reference's inner loop (/root/reference/gcp/prediction/train.py:155-163):
    optimizer.zero_grad(); output = model(inputs); losses = model.loss(...); total = model.get_total_loss(...)
    total.value.backward(); optimizer.step()
with optimizer = RAdam(lr, betas=(adam_beta, 0.999)) (gcp_builder.py:88-89,178-179; gradient_clip defaults to None).

The reference leans on torch autograd; here the backward pass is a second recorded launch plan, generated from the
records the forward plan leaves behind (`plan.rec`), and replayed as a hipGraph:
  * data gradients of Linear / Conv1d / ConvTranspose / strided-conv layers: gcpx_gemm with transposed weight packs;
    of the decoder's 3x3 convs: gcpx_conv3x3 with flipped + transposed packs;
  * weight gradients: gcpx_wgrad (f32 MFMA "TN" GEMM; convs as implicit im2col);
  * everything in between (LSTM cell, GroupNorm, BatchNorm, bilinear upsample, skips broadcast, interleave, sampling,
    losses): csrc/backward.hip.
Gradients land in ONE flat fp32 vector laid out like `model.theta` (so the data-parallel all-reduce is a single RCCL
call over one buffer) and the optimizer + the re-pack of the MFMA-ordered weights are one launch each.
"""
import ctypes as C
import os
import re

import torch

from . import packing as pk
from . import runtime as rt
from .plan_ops import _Plan, _addr, N_LANES
from .params import decoder_layers
from .backward_weights import BackwardWeightsMixin, _c16      # noqa: F401 (re-exported)
from .backward_ops import BackwardOpsMixin, _PtrHolder      # noqa: F401
from .backward_plan import BackwardPlanMixin
from .backward_stages import BackwardStagesMixin


class GCPTrainStep(BackwardWeightsMixin, BackwardOpsMixin, BackwardPlanMixin, BackwardStagesMixin):
    """`step(inputs)` = one optimisation step of `model` (GCPTreeModel) on one minibatch; `backward(inputs)` stops
    after the gradients (tests)."""

    OPTIMIZERS = {"radam": 0, "adam": 1, "rmsprop": 2, "sgd": 3}
    _early_on, _caller, _slice_stream, last_bplan_caller_lane = False, None, None, False          # (see __init__: early_optimizer)
    _pending_slices = ()

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None, optimizer="radam", momentum=0.0,
                 gradient_clip=None):
        """optimizer / momentum / gradient_clip: the trainer's `optimizer` ('radam' default, 'adam', 'rmsprop', 'sgd'), `momentum`
        (RMSprop / SGD) and `gradient_clip` settings (gcp_builder.py:174-186,255-263)."""
        if optimizer not in self.OPTIMIZERS:
            raise ValueError("Optimizer '{}' not supported!".format(optimizer))          # gcp_builder.py:185
        self.optimizer, self.momentum, self.gradient_clip = optimizer, float(momentum), gradient_clip
        self._clip_state_dirty, self._clip_part = True, None      # the clip coefficient slot of the optimizer state / its scratch
        hp = model._hp
        assert hp.decoder_distribution == "discrete_logistic_mixture", "training path implements the DLM head"
        if hp.attentive_inference:
            assert hp.n_attention_heads == 1, "attention backward is built for one head (hyperparameters.py:24 default)"
        self.m = model
        self.lr, self.betas, self.eps = lr, betas, eps
        self.pg = process_group
        model.train()
        model.save_for_backward = True
        self.grad = torch.zeros_like(model.theta)
        self.exp_avg = torch.zeros_like(model.theta)
        self.exp_avg_sq = torch.zeros_like(model.theta)
        self.opt_state = torch.zeros(4, device=model.device)
        from .dist import GradBuckets, gradient_bucket_ranges
        ranges = gradient_bucket_ranges(model._poff, hp.hierarchy_levels, hp.untied_layers)
        self._ranges = ranges
        self._bucket_index = {name: i for i, (name, _, _) in enumerate(ranges)}
        self.buckets = GradBuckets(self.grad, ranges, process_group) if process_group is not None else None
        # step(): the optimizer update of a tree level's slice (+ the re-pack of its weights) is issued on the CALLER's stream — idle
        # while the backward plan runs on the model's lanes — as soon as the backward reports the slice final (and, data-parallel, its
        # all-reduce is done): the 7 levels' 10.3 M parameters each (c2) are updated under the remaining levels and the encoder
        # backward, instead of 2.9 GB of optimizer + re-pack traffic behind the last gradient.  Not with gradient_clip (the global norm
        # needs every slice) and not with a graph-captured backward (no host callbacks).  backward() alone never touches parameters.
        # Issued at a level's mark, a full grid of the optimizer kernel stretched the tree chain's GEMMs beside it from 41 to 101 us
        # (profiles/r04n_train_timeline.txt) and took back 0.4 of the 0.58 ms; with the levels' merge chains on the caller's stream
        # (merge_on_caller_lane) the slices are held until the last chain is out and run under the encoder backward (_on_mark): -0.24 ms
        # per step (tools/ab_train_attr.py early_optimizer 1 0).  `early_blocks` > 0 holds their launches to that many workgroups
        # (one workgroup per CU, 256, is the default since round 6: -0.1 ms against full grids; 128: +0.3 ms, 64: +1.6).
        self.early_optimizer = os.environ.get("GCPX_NO_EARLY_OPTIMIZER") is None
        self.early_blocks = int(os.environ.get("GCPX_EARLY_BLOCKS", "256"))     # (round 6 sweep: 0 = full grids 12.00 ms, 64: 13.6, 128: 12.3, 256: 11.90, 512: 11.96)
        self.early_on_caller = os.environ.get("GCPX_EARLY_STREAM", "caller") == "caller"     # else: communication stream / last side lane
        self._early_on, self._applied, self._caller = False, set(), None
        self.split_dgrad_wide = os.environ.get("GCPX_NO_SPLIT_DGRAD_WIDE") is None     # data gradients of the 32- / 64-channel decoder blocks on the split-f16 kernel
        self.bk = model.build_arena(self._pack_backward)
        self._pack_backward_split()
        self._live_gemm_split()
        self._bplans = {}
        # backward plans are built from forward plans: drop them whenever the model drops those (load_state_dict, build_arena)
        model._plan_listeners = model._plan_listeners + [self._bplans.clear]
        self.wgroup_min_blocks = int(os.environ.get("GCPX_WGROUP_MIN", "256"))
        self.fused_image_wgrad = os.environ.get("GCPX_IMAGE_WGRAD_UNFUSED") is None
        self.split_wgrad = os.environ.get("GCPX_WGRAD_NOSPLIT") is None     # decoder conv weight gradients on the split-f16 kernel
        # the I_0 / I_g encoder backward chains on the side lanes beside the trajectory pass's: the backward's tail gets 0.15 ms shorter in
        # tools/train_phase_times.py, the step does not (8 same-box pairs, c2: 14.94 ms with, 14.64 without) — off
        self.parallel_encoder_passes = os.environ.get("GCPX_PARALLEL_ENCODER_BWD") is not None
        self.wgrad_per_cu = int(os.environ.get("GCPX_WGRAD_PER_CU", "1"))         # conv weight gradients: workgroups per CU (0: the kernel's own occupancy)
        self.lean_mean_grad = os.environ.get("GCPX_NO_LEAN_MEAN_GRAD") is None   # adaptive: head backward over the 80 slots of the mixture mean
        self.fuse_stage = os.environ.get("GCPX_NO_STAGE_FUSION") is None     # 16-channel upsampling blocks: weight gradient without gcpx_conv_stage
        self.fuse_head_act = os.environ.get("GCPX_NO_HEAD_ACT_FUSION") is None     # activation backward of the last decoder block in the head's data gradient
        self.fuse_skip = os.environ.get("GCPX_NO_SKIP_FUSION") is None             # skip-connection sum of a 16 + 16 channel block in the activation pass in front of it
        self.split_wgrad_rows = os.environ.get("GCPX_WGRAD_ROWS_NOSPLIT") is None   # the tree's Linear / LSTM weight gradients (>= 256 rows) likewise
        # The parent-state merge chain of a level (batched d h_prev GEMM -> merge data gradient -> accumulation into the parents' states:
        # three dependent launches that only the NEXT level's LSTM backward waits for) runs on the CALLER's stream as a fourth lane — idle
        # during the backward, with a hardware queue of its own (a fifth stream would share one of the four) — beside the level's
        # embedding / latent / Predictor chain on lane 0
        self.merge_on_caller_lane = os.environ.get("GCPX_NO_MERGE_LANE") is None
        # LSTM cell backward in the epilogue of the GEMM that feeds it (gcpx_gemm_args.lstm_bwd): 3 launches per level off the chain, but
        # the cell's dependent loads then sit behind the K loop of a launch with 32 .. 64 workgroups: c2 step 13.07 ms with, 12.91 without
        # (tools/ab_train_attr.py fuse_lstm_bwd 1 0); gcp_sequential 24.3 either way (host issue time 10 -> 7.6 ms).  Off.
        self.fuse_lstm_bwd = os.environ.get("GCPX_LSTM_BWD_FUSION") is not None
        self.zero_on_side_lane = os.environ.get("GCPX_ZERO_ON_LANE0") is None
        self.heads_on_side_lane = os.environ.get("GCPX_NO_HEADS_ASIDE") is None

        self.batch_dh = os.environ.get("GCPX_NO_BATCH_DH") is None                 # a level's d h_prev GEMMs as one batched launch
        self.group_mlp_bwd = os.environ.get("GCPX_NO_MLP_BWD_GROUP") is None       # a level's posterior + prior backward as one launch
        self.early_fork = os.environ.get("GCPX_EARLY_FORK") is not None   # measured: forking the head's weight gradient before its data gradient costs 0.25 ms (contention on the critical lane)
        self.fused_mlp_bwd = os.environ.get("GCPX_NO_FUSED_MLP_BWD") is None
        self._pad_fixups = []                 # (see _mlp_in_dst)
        # the decoder's weight gradients (5 ms of throughput-bound kernels) are forked after the decoder's data-gradient chain: they then
        # fill the chip during the latency-bound tree phase instead of competing with the data gradients (23.2 -> 22.8 ms / step)
        self.defer_decoder_side = os.environ.get("GCPX_NO_DEFER_DEC_SIDE") is None
        # ... and held back further, until the tree backward has passed its large levels: issued right behind the decoder's data
        # gradients they ran beside the level L-1 / L-2 GEMMs of the tree (1024 / 512 rows, throughput-bound) and stretched two of them
        # from ~0.1 to 1.5 ms each on the critical lane (profiles/r02f_train_lanes.txt); below those levels the tree backward is a
        # chain of small launches that leaves the chip to the weight gradients.  GCPX_DEC_SIDE_LEVEL = level after which they go out
        # (>= L: right after the decoder, the old behaviour).  Default L - 1: with the split-f16 weight gradients (3.4 instead of 7 ms of
        # side-lane kernels) holding them past the largest level only is best — same-box, c2: 16.8-17.2 ms / step against 17.3-17.4 (L - 2)
        # and 17.2-17.5 (L); c5: 22.0 against 22.4 / 22.3.
        # Round 3 (the tree's own weight gradients on the split-f16 kernel, 1.6 -> 1.2 ms of side-lane kernels per step): three runs each
        # on one box, c2: 15.36 ms (L - 1), 15.14 (L - 3), 15.12 (L - 4) — L - 3.
        self.dec_side_level = int(os.environ.get("GCPX_DEC_SIDE_LEVEL", str(max(0, hp.hierarchy_levels - 3))))
        self.group_wgrads = os.environ.get("GCPX_NO_WGROUP") is None   # one grouped launch per level and kernel variant
        self.side_lanes = bool(hp.untied_layers)   # tied levels accumulate into the same weights: keep them on one lane
        self.wgrad_waves = 8192               # wavefronts a split weight-gradient launch aims for (latency hiding)
        # The backward plan is replayed EAGERLY over real streams by default: as parallel branches of one hipGraph the runtime
        # maps nodes of the critical chain onto the same hardware queue as multi-millisecond weight-gradient kernels and
        # serialises them (measured: 32.0 ms / step as a graph, 28.0 ms eager, c2).
        self.backward_graph = False
        # GCPX_BWD_SEGMENTS=n > 0: EVERY run of >= n consecutive launches on one lane is replayed as a small linear hipGraph
        # (plan_ops.compact); 0: only where the plan asks for it (rec["segment_ranges"]); -1: nowhere
        self.segment_graphs = int(os.environ.get("GCPX_BWD_SEGMENTS", "0"))
        # posterior / prior / merge chains of a level on three lanes: measured SLOWER (30.0 vs 28.2 ms / step) — the side lanes are
        # busy with the previous level's weight gradients, so the forked chains queue behind them.  Kept for experiments.
        self.parallel_level_chains = False
        self.n_side = int(__import__("os").environ.get("GCPX_NSIDE", N_LANES - 1))   # side lanes of the backward plan
        self.side_priority = 0                # middle priority; lowest (> 0) starves the side lanes: 40.9 ms / step
        self._lanes = None
        self._lane_streams = None
        self._zeros = torch.zeros(256, device=model.device)

    # ------------------------------------------------------------------------------------------------
    # running
    # ------------------------------------------------------------------------------------------------
    def backward(self, inputs, noise=None):
        """forward (phase 'train', losses on device) + backward; gradients in self.grad.  Returns the forward outputs."""
        m = self.m
        if self.buckets is not None:
            self.buckets.begin()          # a backward without an optimizer step in between must exchange its buckets again
        out = m.forward(inputs, "train", noise)
        if "losses" not in out.raw:
            raise ValueError("the training step needs traj_seq and pad_mask")
        key = [k for k, v in m._plans.items() if v[1].outs is out.raw][0]
        if key not in self._bplans:
            self._bplans[key] = self._build_backward(m._plans[key][1])
        bplan = self._bplans[key]
        caller = torch.cuda.current_stream(m.device)
        self._caller = caller
        self._applied = set()
        self._pending_slices = []
        self.last_bplan_caller_lane = bool(bplan.rec.get("caller_lane"))
        m._stream.wait_stream(caller)
        stream = m._stream.cuda_stream
        if m.use_graph and self.backward_graph:
            if bplan.graph is None:
                bplan.run(m._streams)
                bplan.graph = m._capture(bplan, bplan.ops, stream)
            rt.check(m.lib.gcpx_graph_launch(bplan.graph, stream), "graph_launch")
        else:
            # (the caller's stream is the plan's last lane: the levels' merge chains, and step()'s early optimizer slices behind them)
            lanes = self._backward_streams() + [caller.cuda_stream]
            ops = None
            seg_ranges = None if self.segment_graphs > 0 else bplan.rec.get("segment_ranges")
            if (self.segment_graphs > 0 or (seg_ranges and self.segment_graphs == 0)) and not bplan.rec.get("caller_lane"):
                # runs of launches on one lane as small linear graphs (fewer host calls; lanes, events and marks unchanged): everywhere
                # (GCPX_BWD_SEGMENTS=n) or where the plan asks for it (`segment_ranges`: chains with slack, GCPX_BWD_SEGMENTS=-1: nowhere)
                if bplan.rec.get("_segments") is None:
                    bplan.run(lanes, on_mark=self._on_mark)
                    torch.cuda.synchronize(m.device)
                    bplan.rec["_segments"] = bplan.compact(self._backward_streams(), max(self.segment_graphs, 3), seg_ranges)
                    self.last_bplan = bplan
                    caller.wait_stream(m._stream)
                    return out
                ops = bplan.rec["_segments"]
            bplan.run(lanes, ops=ops, on_mark=self._on_mark)
        caller.wait_stream(m._stream)
        self.last_bplan = bplan
        return out

    def _on_mark(self, tag, payload):
        if tag == "slices":
            self._issue_pending_slices()
            return
        if tag != "bucket" or (self.buckets is None and not self._early_on):
            return
        if self._lane_streams is None:
            self._lane_streams = [torch.cuda.ExternalStream(int(s.value if hasattr(s, "value") else s), device=self.m.device)
                                  for s in self._backward_streams()]
        if self.buckets is not None:
            self.buckets.reduce_async(payload, after_streams=self._lane_streams)
        if self._early_on:
            if self.early_on_caller and self.last_bplan_caller_lane:
                # The caller's stream also carries the levels' merge chains (merge_on_caller_lane): a slice issued here would wait for
                # the level's weight gradients on the side lanes and hold up the next level's merge chain behind it (measured: 15.0
                # instead of 13.2 ms / step).  Remember where the lanes stand (events) and issue the slices behind the last merge chain
                # (the plan's "slices" mark): they then run under the encoder backward.
                evs = []
                for s_ in self._lane_streams:
                    e = torch.cuda.Event()
                    e.record(s_)
                    evs.append(e)
                self._pending_slices.append((payload, evs))
                return
            # elsewhere: at once, on the caller's stream or — early_on_caller = False — on the communication stream (data-parallel: the
            # slice follows its all-reduce there) / the last side lane
            if self.early_on_caller:
                on = self._caller
            elif self.buckets is not None and self.buckets.comm_stream is not None:
                on = self.buckets.comm_stream
            else:
                on = self._lane_streams[-1]
            self._slice_stream = on
            self._apply_slice(payload, tick=False, on=on, after=[s_ for s_ in self._lane_streams if s_ is not on],
                              max_blocks=self.early_blocks)
            self._applied.add(payload)

    def _issue_pending_slices(self):
        for i, evs in self._pending_slices:
            self._slice_stream = self._caller
            self._apply_slice(i, tick=False, on=self._caller, after=evs, max_blocks=self.early_blocks)
            self._applied.add(i)
        self._pending_slices = []

    def _apply_slice(self, i, tick, on, after=(), max_blocks=0):
        """optimizer update of slice i of the flat vectors + re-pack of the weights that gather from it, on stream `on` (which first
        waits for `after` — streams or events: the lanes that produce the slice's gradient — and for the slice's all-reduce)"""
        m = self.m
        name, lo, hi = self._ranges[i]
        caller = on
        for s in after:
            if isinstance(s, torch.cuda.Event):
                caller.wait_event(s)
            else:
                caller.wait_stream(s)
        scale = 1.0
        if self.buckets is not None and torch.distributed.is_initialized():
            w = self.buckets.works.get(i)
            if w is not None:
                with torch.cuda.stream(caller):
                    w.wait()                                   # the stream waits for the collective (no host block)
            scale = 1.0 / torch.distributed.get_world_size(self.buckets.group)
        kind = self.OPTIMIZERS[self.optimizer]
        p1, p2 = (self.betas[0], self.betas[1]) if kind <= 1 else ((self.momentum, 0.99) if kind == 2 else (self.momentum, 0.0))
        st = caller.cuda_stream
        o = 4 * lo
        rt.check(m.lib.gcpx_optim_range(m.theta.data_ptr() + o, self.grad.data_ptr() + o, self.exp_avg.data_ptr() + o,
                                        self.exp_avg_sq.data_ptr() + o, self.opt_state.data_ptr(), hi - lo, kind, self.lr, p1, p2, self.eps,
                                        scale, 1 if tick else 0, max_blocks, st), "optim_range")
        m.repack(st, bucket=name, max_blocks=max_blocks)

    def _backward_streams(self):
        """lane 0 = the model's stream, side lanes = the model's own side streams: the process then uses four streams in all
        (torch's, the model's three), one per hardware queue of the default runtime configuration — extra streams get
        multiplexed onto the same queues and serialise the lanes again (measured: +4 ms / step)."""
        if self._lanes is None:
            m = self.m
            self._lanes = list(m._streams[:1 + self.n_side])
            for _ in range(1 + self.n_side - len(self._lanes)):
                sp = C.c_void_p()
                with torch.cuda.device(m.device):
                    rt.check(m.lib.gcpx_stream_create_priority(C.byref(sp), int(self.side_priority)), "stream_create")
                self._lanes.append(sp)
        return self._lanes

    def optimizer_step(self):
        """The configured optimizer (RAdam / Adam / RMSprop / SGD, optional clip_grad_norm_) on the flat vectors + one re-pack gather
        (gcp_builder.py:88-89,178-179)."""
        m = self.m
        st = torch.cuda.current_stream(m.device).cuda_stream
        scale = 1.0
        if self.buckets is not None:
            # the tree-level buckets were started during the backward; the last one (conv stacks, heads, level 0) goes now
            scale = self.buckets.finish()
        applied, self._applied = self._applied, set()
        if applied:
            # step(): those slices were updated during the backward; the others (at least "rest") follow here, the last one ticks
            assert self._caller is not None and self._caller.cuda_stream == st, "step() runs on one stream"
            if self._clip_state_dirty:
                self.opt_state[1:2].zero_()      # (ordered behind the early slices: they already read it — see step())
                self._clip_state_dirty = False
            if self._slice_stream is not None and self._slice_stream is not self._caller:
                self._caller.wait_stream(self._slice_stream)
            todo = [i for i in range(len(self._ranges)) if i not in applied]
            for j, i in enumerate(todo):
                self._apply_slice(i, tick=(j == len(todo) - 1), on=self._caller)
            return
        n = m.theta.numel()
        if self.gradient_clip:
            # clip_grad_norm_ over all parameters of the (averaged) gradient: its coefficient lands in opt_state[1], which the step reads
            if self._clip_part is None:
                self._clip_part = torch.empty(1024, device=m.device)
            rt.check(m.lib.gcpx_grad_clip_coef(self.grad.data_ptr(), n, scale, float(self.gradient_clip), self._clip_part.data_ptr(), 1024,
                                               self.opt_state.data_ptr(), st), "grad_clip")
        elif self._clip_state_dirty:
            # opt_state[1] is the clip coefficient the step kernels apply whenever it is > 0: without clipping it must be 0 — also after
            # resuming a checkpoint that was trained WITH clipping (the whole opt_state is restored)
            self.opt_state[1:2].zero_()
            self._clip_state_dirty = False
        kind = self.OPTIMIZERS[self.optimizer]
        if kind == 0:
            rt.check(m.lib.gcpx_radam_step(m.theta.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                           self.opt_state.data_ptr(), n, self.lr, self.betas[0], self.betas[1], self.eps, scale, st), "radam")
        else:
            # torch.optim defaults of the reference's get_optimizer_class: Adam betas (adam_beta, 0.999); RMSprop alpha 0.99; eps 1e-8
            p1, p2 = (self.betas[0], self.betas[1]) if kind == 1 else ((self.momentum, 0.99) if kind == 2 else (self.momentum, 0.0))
            rt.check(m.lib.gcpx_optim_step(m.theta.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                           self.opt_state.data_ptr(), n, kind, self.lr, p1, p2, self.eps, scale, st), "optim_step")
        m.repack(st)

    def step(self, inputs, noise=None):
        self._early_on = (self.early_optimizer and not self.gradient_clip and len(self._ranges) > 1 and
                          not (self.m.use_graph and self.backward_graph))
        if self._early_on and self._clip_state_dirty:
            self.opt_state[1:2].zero_()          # the early slices read the clip coefficient: it must be 0 (unset) before they run
            self._clip_state_dirty = False
        try:
            out = self.backward(inputs, noise)
        finally:
            self._early_on = False
        self.optimizer_step()
        return out

    def optimizer_state(self):
        """optimizer.state_dict() counterpart (train.py:111): flat first / second moments + step counter"""
        return {"exp_avg": self.exp_avg.detach().cpu(), "exp_avg_sq": self.exp_avg_sq.detach().cpu(),
                "state": self.opt_state.detach().cpu(), "lr": self.lr, "betas": tuple(self.betas), "eps": self.eps,
                "optimizer": self.optimizer, "momentum": self.momentum, "gradient_clip": self.gradient_clip}

    def load_optimizer_state(self, st):
        self.exp_avg.copy_(st["exp_avg"])
        self.exp_avg_sq.copy_(st["exp_avg_sq"])
        self.opt_state.copy_(st["state"])
        self.lr, self.betas, self.eps = st["lr"], tuple(st["betas"]), st["eps"]
        if st.get("optimizer", self.optimizer) != self.optimizer:
            raise ValueError(f"checkpoint holds the state of optimizer '{st['optimizer']}', this trainer runs '{self.optimizer}'")
        # momentum / gradient_clip are the TRAINER's settings (the conf's), as torch.optim's load_state_dict keeps the param-group
        # values it is given; a checkpoint trained with other values resumes, but says so
        for k in ("momentum", "gradient_clip"):
            if k in st and st[k] != getattr(self, k):
                import warnings
                warnings.warn(f"checkpoint was trained with {k}={st[k]!r}, this trainer runs {k}={getattr(self, k)!r}")
        self._clip_state_dirty = True         # a restored clip coefficient must not outlive a trainer without clipping

    def named_grads(self):
        return {k: self.grad[o:o + int(torch.tensor(shp).prod())].view(shp) for k, (o, shp) in self.m._poff.items()}
