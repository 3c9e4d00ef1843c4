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
from .model import _Plan, _addr, N_LANES
from .params import decoder_layers


def _c16(n):
    return (n + 15) // 16 * 16


class GCPTrainStep:
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
        # (128: +0.07 ms, 32: +3.5 ms; 0 = full grids).
        self.early_optimizer = os.environ.get("GCPX_NO_EARLY_OPTIMIZER") is None
        self.early_blocks = int(os.environ.get("GCPX_EARLY_BLOCKS", "0"))
        self.early_on_caller = os.environ.get("GCPX_EARLY_STREAM", "caller") == "caller"     # else: communication stream / last side lane
        self._early_on, self._applied, self._caller = False, set(), None
        self.split_dgrad_wide = os.environ.get("GCPX_NO_SPLIT_DGRAD_WIDE") is None     # data gradients of the 32- / 64-channel decoder blocks on the split-f16 kernel
        self.bk = model.build_arena(self._pack_backward)
        self._pack_backward_split()
        self._live_gemm_split()
        self._bplans = {}
        # backward plans are built from forward plans: drop them whenever the model drops those (load_state_dict, build_arena)
        model._plan_listeners = getattr(model, "_plan_listeners", []) + [self._bplans.clear]
        self.wgroup_min_blocks = int(os.environ.get("GCPX_WGROUP_MIN", "256"))
        self.fused_image_wgrad = os.environ.get("GCPX_IMAGE_WGRAD_UNFUSED") is None
        self.split_wgrad = os.environ.get("GCPX_WGRAD_NOSPLIT") is None     # decoder conv weight gradients on the split-f16 kernel
        # the I_0 / I_g encoder backward chains on the side lanes beside the trajectory pass's: the backward's tail gets 0.15 ms shorter in
        # tools/train_phase_times.py, the step does not (8 same-box pairs, c2: 14.94 ms with, 14.64 without) — off
        self.parallel_encoder_passes = os.environ.get("GCPX_PARALLEL_ENCODER_BWD") is not None
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
        # posterior / prior / merge chains of a level on three lanes: measured SLOWER (30.0 vs 28.2 ms / step) — the side lanes are
        # busy with the previous level's weight gradients, so the forked chains queue behind them.  Kept for experiments.
        self.parallel_level_chains = False
        self.n_side = int(__import__("os").environ.get("GCPX_NSIDE", N_LANES - 1))   # side lanes of the backward plan
        self.side_priority = 0                # middle priority; lowest (> 0) starves the side lanes: 40.9 ms / step
        self._lanes = None
        self._lane_streams = None
        self._zeros = torch.zeros(256, device=model.device)

    # ------------------------------------------------------------------------------------------------
    # transposed weight packs (data-gradient GEMMs / convs)
    # ------------------------------------------------------------------------------------------------
    def _pack_predictor_T(self, sd, prefix, splits):
        """splits: list of (col0, width) groups of the input layer whose gradients go to different places."""
        T = {}
        w_out = sd[f"{prefix}.head.linear.weight"]
        od = w_out.shape[0]
        w_out = pk._pad_rows(w_out, _c16(od))
        T["wT_out"] = pk.pack_gemm(w_out.t().contiguous())                 # [N = mid][K = out_pad]
        l = 0
        while f"{prefix}.pyramid-{l}.linear.weight" in sd:
            T[f"wT_mid{l}"] = pk.pack_gemm(sd[f"{prefix}.pyramid-{l}.linear.weight"].t().contiguous())
            l += 1
        w_in = sd[f"{prefix}.input.linear.weight"]
        for i, (c0, w) in enumerate(splits):
            T[f"wT_in{i}"] = pk.pack_gemm(w_in[:, c0:c0 + w].t().contiguous())   # [N = w][K = mid]
        return T

    def _pack_backward_split(self):
        """Split-f16 pieces of the transposed, flipped weights of the data-gradient convs that run the wave-autonomous kernel
        (conv3x3_wave_split_kernel): the output head's (112 kernel slots -> 16) and the 16-channel decoder blocks' (16 -> 32).  Like
        the forward's split weights they are index gathers of the flat parameter vector (model.pk_split), re-split by
        gcpx_split_pack behind every optimizer step."""
        m, hp = self.m, self.m._hp
        if not m.split_f16:
            return

        def ids_of(key):
            off, shp = m._poff[key]
            n = 1
            for d in shp:
                n *= d
            return (torch.arange(n, dtype=torch.float64) + (off + 1)).view(shp)

        todo = {}
        if hp.decoder_distribution == "discrete_logistic_mixture":
            hw = ids_of("decoder.gen_head.conv.weight")                      # [100, 16, 3, 3]; inputs of the dgrad = kernel slots
            perm = torch.as_tensor(pk.dlm_channel_perm(hp.n_mixtures))
            wk = torch.zeros((len(perm),) + tuple(hw.shape[1:]), dtype=hw.dtype)
            wk[perm >= 0] = hw[perm[perm >= 0]]
            todo["bw.dec.head"] = wk.flip(2, 3).transpose(0, 1).contiguous()
        for name, c_prev, c_skip, skip_idx, cout in decoder_layers(hp):
            if cout == 16 and c_prev + c_skip == 32:
                todo[f"bw.dec.{name}"] = ids_of(f"decoder.net.{name}.conv.weight").flip(2, 3).transpose(0, 1).contiguous()
            elif self.split_dgrad_wide and cout % 16 == 0 and (c_prev + c_skip) % 32 == 0:
                # wider blocks: 32 of the block's input channels per launch on the same kernel (conv3x3_wave_split_kernel<2>: the exact
                # f32 tiles they ran on take 3x the MFMA time), where the frame is >= 16 wide (_decoder_backward)
                wT = ids_of(f"decoder.net.{name}.conv.weight").flip(2, 3).transpose(0, 1).contiguous()          # [cin, cout, 3, 3]
                for h in range((c_prev + c_skip) // 32):
                    todo[f"bw.dec.{name}.q{h}"] = wT[32 * h:32 * (h + 1)].contiguous()
        for name, wT in todo.items():
            idx = (pk.conv3x3_split_gather(wT).reshape(-1) - 1).to(torch.int32).to(m.device)
            m.pk_split[name] = dict(idx=idx, out=torch.zeros(2 * idx.numel(), dtype=torch.int16, device=m.device),
                                    log2=torch.zeros(1, dtype=torch.int32, device=m.device))
        m.repack_split()

    def _live_gemm_split(self):
        """The GEMM weights of the tree levels that run with >= 512 rows at the configured batch size, kept in split-f16 form ALSO in
        training (the inference model splits them once at weight load, model._pack_gemm_split): their forward merge / output GEMMs and every
        data-gradient GEMM of those levels then run on the split-f16 kernels (3x the f32 MFMA rate) instead of the exact f32 tiles.  Each
        pack is an index gather of the flat parameter vector — the arena's index map pushed through packing.unpack_gemm / gemm_split_gather —
        re-split by gcpx_split_pack_group2 with the slice of the optimizer step it belongs to (model.repack(bucket=): under the encoder
        backward, like the slice itself)."""
        m, hp = self.m, self.m._hp
        m._gsplit, m._gsplit_tabs, m._gsplit_live = {}, {}, False
        if not (m.split_f16 and hp.tree_lstm and os.environ.get("GCPX_NO_LIVE_GEMM_SPLIT") is None):
            return
        min_rows = int(os.environ.get("GCPX_GEMM_SPLIT_MIN_ROWS", "512"))
        L = hp.hierarchy_levels
        levels = [l for l in range(L) if hp.batch_size * 2 ** l >= min_rows and f"tree{l}" in m.pk and f"tree{l}" in self.bk]
        if not hp.untied_layers:
            levels = [0] if levels else []
        base = m._arena.data_ptr()
        descs = {}
        keep = []
        for l in levels:
            bucket = f"tree{l}" if f"tree{l}" in [n_ for n_, _, _ in m._arena_ranges] else m._arena_ranges[-1][0]

            fwd = lambda k: k in ("proj.w", "out.w", "embed.w") or re.fullmatch(r"lstm\d+\.w", k)
            bwd = lambda k: k in ("proj.wT", "out.wT", "embed.wT", "lstm.whT") or re.fullmatch(r"lstm\d+\.wxT", k)
            leaves = [(k, v) for k, v in m.pk[f"tree{l}"].items() if torch.is_tensor(v) and fwd(k)]
            leaves += [(k, v) for k, v in self.bk[f"tree{l}"].items() if torch.is_tensor(v) and bwd(k)]
            for k, leaf in leaves:
                stack = leaf if leaf.dim() == 5 else leaf[None]
                KG, NT = stack.shape[1], stack.shape[2]
                N, K = NT * 16, KG * 16
                if K % 64 or N % 64:
                    continue
                off = (leaf.data_ptr() - base) // 4
                idx0 = m._arena_idx0[off:off + leaf.numel()].view(stack.shape).to(torch.int64)
                n_el = N * K
                ws = torch.zeros(stack.shape[0], 2 * n_el, dtype=torch.int16, device=m.device)
                es = torch.zeros(stack.shape[0], dtype=torch.int32, device=m.device)
                for b in range(stack.shape[0]):
                    ids = pk.gemm_split_gather(pk.unpack_gemm(idx0[b] + 1, N)).reshape(-1) - 1          # (-1: a zero-padded slot)
                    ids = ids.to(torch.int32).contiguous()
                    e = rt.SplitPackDesc()
                    e.src, e.idx, e.out, e.log2_out, e.n = m.theta.data_ptr(), ids.data_ptr(), ws[b].data_ptr(), es[b:b + 1].data_ptr(), n_el
                    descs.setdefault(bucket, []).append(e)
                    keep.append(ids)
                m._gsplit[leaf.data_ptr()] = (ws, es)
        self._gsplit_keep = keep
        for bucket, ds in descs.items():
            arr = (rt.SplitPackDesc * len(ds))(*ds)
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(m.device)
            m._gsplit_tabs[bucket] = (dev, len(ds), torch.zeros(len(ds), dtype=torch.int32, device=m.device))
        m._gsplit_live = bool(descs)
        m._repack_gsplit(torch.cuda.current_stream(m.device).cuda_stream, None)

    def _pack_backward(self, sd):
        m, hp = self.m, self.m._hp
        nz, nv, H = hp.nz_enc, hp.nz_vae, hp.nz_mid_lstm
        X = {}
        layers, ctop = m._enc_layers, m._c_top
        for name, cin, cout, norm in layers[1:]:
            w = sd[f"encoder.net.{name}.conv.weight"]                       # [co, ci, 4, 4] -> [n = (tap, ci)][k = co]
            X[f"enc.{name}.wT"] = pk.pack_gemm(w.permute(2, 3, 1, 0).reshape(16 * cin, cout))
        wh = sd["encoder.net.head.weight"]                                   # [nz, c, 4, 4] -> [n = (tap, c)][k = nz]
        X["enc.head.wT"] = pk.pack_gemm(wh.permute(2, 3, 1, 0).reshape(16 * ctop, nz))
        wt = sd["decoder.net.input.conv.weight"]                             # [nz, co, 4, 4] -> [n = nz][k = (tap, co)]
        X["dec.input.wT"] = pk.pack_gemm(wt.permute(0, 2, 3, 1).reshape(nz, 16 * ctop))
        for name, c_prev, c_skip, skip_idx, cout in decoder_layers(hp):
            w = sd[f"decoder.net.{name}.conv.weight"]                       # dgrad = 3x3 conv with w^T flipped
            wT = w.flip(2, 3).transpose(0, 1).contiguous()                  # [cin, cout, 3, 3]
            cin = wT.shape[0]
            for h in range((cin + 63) // 64):
                X[f"dec.{name}.wT{h}"] = pk.pack_conv3x3(wT[64 * h:64 * (h + 1)], 16)
            if self.split_dgrad_wide and m.split_f16 and cin % 32 == 0 and not (cout == 16 and cin == 32):
                # (the f32 twin of the 32-channel launches of _decoder_backward: what the launch falls back to if its split form does not fit)
                for h in range(cin // 32):
                    X[f"dec.{name}.wTq{h}"] = pk.pack_conv3x3(wT[32 * h:32 * (h + 1)], 16)
        hw = sd["decoder.gen_head.conv.weight"]                              # [100, 16, 3, 3]; inputs of the dgrad = kernel slots
        perm = torch.as_tensor(pk.dlm_channel_perm(hp.n_mixtures), device=hw.device)
        wk = torch.zeros((len(perm),) + tuple(hw.shape[1:]), dtype=hw.dtype, device=hw.device)
        wk[perm >= 0] = hw[perm[perm >= 0]]
        X["dec.head.wT"] = pk.pack_conv3x3(wk.flip(2, 3).transpose(0, 1).contiguous(), 16)
        for nm in ["input"] + [f"pyramid-{i}" for i in range(hp.conv_inf_enc_layers)] + ["head"]:
            w = sd[f"inf_encoder.net.{nm}.conv.weight"]                     # [co, ci, 3] -> [n = ci][k = (tap, co)]
            X[f"seq.{nm}.wT"] = pk.pack_gemm(w.permute(1, 2, 0).reshape(w.shape[1], -1))
        if hp.regress_length:
            X["length_pred"] = self._pack_predictor_T(sd, "length_pred.p", [(0, 2 * nz)])
        if hp.attach_state_regressor:
            X["state_regressor"] = self._pack_predictor_T(sd, "state_regressor", [])
        if hp.attach_inv_mdl:
            X["inv_mdl"] = self._pack_predictor_T(sd, "inv_mdl.action_pred", [])
        if hp.attach_cost_mdl:
            X["cost_mdl"] = self._pack_predictor_T(sd, "cost_mdl.cost_pred", [])
        if hp.adaptive:
            X["distance"] = self._pack_predictor_T(sd, "tree_module.tree_modules.0.binding.distance_predictor", [(0, nz), (nz, nz)])
        else:
            X["existence"] = self._pack_predictor_T(sd, "tree_module.tree_modules.0.binding.existence_predictor", [(0, nz)])
        if hp.attentive_inference:
            for nm in ["input"] + [f"pyramid-{i}" for i in range(hp.conv_inf_enc_layers)] + ["head"]:
                w = sd[f"inf_key_encoder.0.net.{nm}.conv.weight"]
                X[f"kseq.{nm}.wT"] = pk.pack_gemm(w.permute(1, 2, 0).reshape(w.shape[1], -1))
            X["kseq.key.wT"] = pk.pack_gemm(sd["inf_key_encoder.1.linear.weight"].t().contiguous())                # [nz][dk]
            n_mod = hp.hierarchy_levels if hp.untied_layers else 1
            att = lambda l, nm: sd[f"tree_module.tree_modules.{l}.inference.attention.attention_layers.0.{nm}.weight"]
            # d keys = [dK'_0 | dK'_1 | ...] @ [Wk_0; Wk_1; ...]: one GEMM over the level blocks laid side by side
            X["attn.k_proj.wT"] = pk.pack_gemm(torch.cat([att(l, "k_proj") for l in range(n_mod)], 0).t().contiguous())   # [dk][n_mod*dk]
            X["attn.v_proj.wT"] = pk.pack_gemm(torch.cat([att(l, "v_proj") for l in range(n_mod)], 0).t().contiguous())   # [nz][n_mod*nz]
        for l in range(hp.hierarchy_levels if hp.untied_layers else 1):
            p = f"tree_module.tree_modules.{l}"
            T = {}
            T["prior"] = self._pack_predictor_T(sd, f"{p}.prior", [(0, 2 * nz)])
            T["q"] = self._pack_predictor_T(sd, f"{p}.inference.q", [(0, 2 * nz), (2 * nz, nz)])
            if not hp.tree_lstm:                                   # non-LSTM subgoal predictor (tree_module.py:109-110)
                T["sg"] = self._pack_predictor_T(sd, f"{p}.subgoal_pred.net", [(0, hp.pred_inp_dim)])
                X[f"tree{l}"] = T
                continue
            T["embed.wT"] = pk.pack_gemm(sd[f"{p}.subgoal_pred.embed.weight"].t().contiguous())
            for i in range(hp.n_lstm_layers):
                T[f"lstm{i}.wxT"] = pk.pack_gemm(sd[f"{p}.subgoal_pred.lstm.{i}.weight_ih"].t().contiguous())   # [H][4H]
            # (stacked: the layers' d h_prev GEMMs of a level are ONE batched launch behind the level's d x chain)
            T["lstm.whT"] = torch.stack([pk.pack_gemm(sd[f"{p}.subgoal_pred.lstm.{i}.weight_hh"].t().contiguous())
                                         for i in range(hp.n_lstm_layers)]).contiguous()
            T["out.wT"] = pk.pack_gemm(sd[f"{p}.subgoal_pred.out.weight"].t().contiguous())                     # [H][nz]
            if hp.tree_lstm == "split_linear":
                T["proj.wT"] = torch.stack([pk.pack_gemm(sd[f"{p}.subgoal_pred.projections.{j}.weight"].t().contiguous())
                                            for j in range(2 * hp.n_lstm_layers)]).contiguous()                 # [2H][H] each
            elif hp.tree_lstm == "linear":
                T["proj.wT"] = pk.pack_gemm(sd[f"{p}.subgoal_pred.projection.weight"].t().contiguous())         # [n = 2 SD][k = SD]
            if l == 0 and hp.lstm_init == "mlp":
                T["init"] = self._pack_predictor_T(sd, f"{p}.lstm_initializer.net", [(0, 2 * nz + nv)])
            if hp.attentive_inference:
                a = f"{p}.inference.attention"
                T["attn.query"] = self._pack_predictor_T(sd, f"{a}.query_net", [(0, 2 * nz)])
                T["attn.q_proj.wT"] = pk.pack_gemm(sd[f"{a}.attention_layers.0.q_proj.weight"].t().contiguous())
                T["attn.out_proj.wT"] = pk.pack_gemm(sd[f"{a}.attention_layers.0.out_proj.weight"].t().contiguous())
                T["attn.out.wT"] = pk.pack_gemm(sd[f"{a}.out.weight"].t().contiguous())
            X[f"tree{l}"] = T
        return X

    # ------------------------------------------------------------------------------------------------
    # plan-building helpers
    # ------------------------------------------------------------------------------------------------
    # Weight / bias gradients are off the critical path (only data gradients chain): they are queued and issued on the
    # side lanes after the producing stage, so under hipGraph capture they become parallel branches of the graph.
    def _side(self, plan, name, fn, *args):
        plan.deferred.append((name, fn, args))

    def _flush(self, plan, one_lane=False, only_lane=None):
        """one_lane: everything of this flush goes to ONE side lane, behind all work issued so far on the others — for gradients that
        ACCUMULATE into parameters an earlier flush (or another op of this one) also accumulates into: the three encoder passes (trajectory
        frames, I_0, I_g) share their weights, and two lanes adding to one address at the same time lose an update."""
        if not plan.deferred:
            return
        if not self.side_lanes:
            for name, fn, args in plan.deferred:
                plan.add(name, fn, *args)
            plan.deferred = []
            return
        lanes = list(range(1, 1 + self.n_side))
        if only_lane is not None:                              # everything of this flush on ONE given side lane (the others stay free)
            lanes = [only_lane]
        plan.fork(lanes)
        if one_lane:
            for other in lanes[1:]:
                plan.wait(lanes[0], other)
            lanes = lanes[:1]
        if self.group_wgrads:
            plan.deferred = self._group_wgrads(plan, plan.deferred)
        # ops of one tag (wgrad + its reduce) stay on one lane, in order
        lane_of = plan.rec.setdefault("_lane_of", {})
        alias = plan.rec.get("_lane_alias", {})
        for name, fn, args in plan.deferred:
            tag = name.split(":", 1)[1] if ":" in name else name
            tag = alias.get(tag, tag)
            if one_lane:
                lane_of[tag] = lanes[0]
            elif tag not in lane_of:
                lane_of[tag] = lanes[len(lane_of) % len(lanes)]
            plan.lane = lane_of[tag]
            plan.add(name, fn, *args)
        plan.lane = 0
        plan.deferred = []

    def _group_wgrads(self, plan, deferred):
        """The direct-mode gcpx_wgrad launches of one flush (the ~40 small weight gradients of a tree level) become ONE grouped
        launch per kernel variant: descriptors and block table are uploaded once, when the plan is built."""
        lib, m = self.m.lib, self.m
        groups, rest, cand = {}, [], []
        v, nb = C.c_int32(), C.c_int32()
        produced = set()          # tags that already have a non-wgrad op queued: a weight gradient of that tag reads its output
        tag_of = lambda nm: nm.split(":", 1)[1] if ":" in nm else nm
        for op in deferred:
            name, fn, args = op
            a = args[0]._obj if fn is lib.gcpx_wgrad else None
            if a is None or tag_of(name) in produced:
                rest.append(op)
                if name.startswith(("bw.act:", "bw.im2col:", "bw.stage:")):
                    produced.add(tag_of(name))
            else:
                cand.append((name, a))
        # the in-workgroup row split exists to fill the chip from ONE small problem; a group that already brings >= 1 workgroup
        # per CU without it runs one wavefront per 64 x 64 tile instead (4x fewer, lighter workgroups)
        total = 0
        for name, a in cand:
            rt.check(lib.gcpx_wgrad_classify(C.byref(a), 0, C.byref(v), C.byref(nb)), name)
            total += nb.value
        split = 0 if total >= self.wgroup_min_blocks else -1
        for name, a in cand:
            rt.check(lib.gcpx_wgrad_classify(C.byref(a), split, C.byref(v), C.byref(nb)), name)
            groups.setdefault(v.value, []).append((name, a, nb.value))
        out = []
        for v, items in sorted(groups.items()):
            for c0 in range(0, len(items), 64):
                chunk = items[c0:c0 + 64]
                if len(chunk) == 1:
                    out.append((chunk[0][0], lib.gcpx_wgrad, (C.byref(chunk[0][1]),)))
                    continue
                tab = (rt.WgradArgs * len(chunk))(*[it[1] for it in chunk])
                raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(m.device)
                starts, tot = [], 0
                for it in chunk:
                    starts.append(tot)
                    tot += it[2]
                bst = torch.tensor(starts, dtype=torch.int32, device=m.device)
                plan.keep += [raw, bst]
                gid = plan.rec["_ngroups"] = plan.rec.get("_ngroups", 0) + 1
                gtag = f"g{gid}.v{v}x{len(chunk)}"
                plan.rec.setdefault("_groups", {})[gtag] = [(it[0], it[1].R, it[1].N, it[1].K, it[2]) for it in chunk]   # (tools: what a group holds)
                out.append((f"bw.wgroup:{gtag}", lib.gcpx_wgrad_group, (raw.data_ptr(), bst.data_ptr(), len(chunk), tot, v)))
                # the reduction of a split problem's partials must follow the group on the same lane
                alias = plan.rec.setdefault("_lane_alias", {})
                for it in chunk:
                    if it[1].partial:
                        alias[it[0].split(":", 1)[1]] = gtag
        return out + rest

    def g(self, name, off=0):
        """device address of the gradient of parameter `name`"""
        return self.grad.data_ptr() + 4 * (self.m._poff[name][0] + off)

    def _wgrad(self, plan, tag, dy, ldy, R, N, x, K, dst, ldw=0, k_off=0, n_valid=None, mode=rt.WG_ROWS, rpb=None, sb=0, sr=0,
               shift=0, rowidx=None, frame_map=None, scale=None, shiftv=None, act=0, cmod=0, Cin=0, H=0, W=0, dy_rpb=0,
               dy_sb=0, wmap=rt.WMAP_LINEAR, ntap=1, Cout=0, n_map=None, dbias=None, dbias2=None, batch=None):
        """dbias / dbias2: gradient addresses of the layer's bias(es) (column sums of dy), fused into the launch when it
        writes directly, a separate gcpx_colsum otherwise.  batch = (nbatch, z_dy_off, z_x_off, z_out_off, z_bias_off)."""
        lib, m = self.m.lib, self.m
        a = rt.WgradArgs()
        n_valid = N if n_valid is None else n_valid
        a.dy, a.x, a.ldy, a.R, a.N, a.n_valid, a.K, a.mode = dy, x, ldy, R, N, n_valid, K, mode
        a.rowidx = rowidx.data_ptr() if rowidx is not None else None
        a.frame_map = frame_map.data_ptr() if frame_map is not None else None
        a.scale = scale.data_ptr() if scale is not None else None
        a.shiftv = shiftv.data_ptr() if shiftv is not None else None
        a.sb, a.sr, a.rpb, a.shift, a.act, a.cmod = sb, sr, (rpb if rpb is not None else R), shift, act, cmod
        a.Cin, a.H, a.W, a.dy_rpb, a.dy_sb = Cin, H, W, dy_rpb, dy_sb
        waves = ((K + 63) // 64) * ((N + 63) // 64 if N > 16 else 1)
        nsplit = max(1, min(self.wgrad_waves // waves, R // 256, 512))
        # plain row problems of whole 128 x 128 blocks with enough rows: the split-f16 kernel (csrc/wgrad_rows_split.hip; one workgroup per
        # block walks all rows, no row split, direct output) unless the model runs on the exact f32 kernels
        if (m.split_f16 and self.split_wgrad_rows and mode == rt.WG_ROWS and rowidx is None and scale is None and not act and shift == 0 and
                R >= 256 and N % 128 == 0 and K % 128 == 0 and n_valid == N and wmap == rt.WMAP_LINEAR and ldw % 4 == 0 and k_off % 4 == 0):
            a.split_f16, nsplit = 1, 1
        if batch is not None:
            nsplit = 1
            a.nbatch, a.z_dy_off, a.z_x_off, a.z_out_off, a.z_bias_off = batch
        if wmap == rt.WMAP_LINEAR and nsplit == 1 and ldw % 4 == 0 and k_off % 4 == 0:
            a.out, a.ldw, a.k_off, a.accumulate, a.partial, a.nsplit = dst, ldw, k_off, 1, 0, 1
            a.dbias, a.dbias2 = dbias, dbias2
            plan.keep.append(a)
            self._side(plan, f"bw.wgrad:{tag}", lib.gcpx_wgrad, C.byref(a))
            return
        assert batch is None
        if dbias is not None:
            self._colsum(plan, tag, dy, ldy, R, n_valid, dbias, dst2=dbias2, dy_rpb=dy_rpb, dy_sb=dy_sb)
        part = m._buf(f"bw.part:{tag}", (nsplit, n_valid, K))
        a.out, a.partial, a.nsplit = part.data_ptr(), 1, nsplit
        plan.keep.append(a)
        self._side(plan, f"bw.wgrad:{tag}", lib.gcpx_wgrad, C.byref(a))
        self._side(plan, f"bw.wreduce:{tag}", lib.gcpx_wgrad_reduce, part.data_ptr(), nsplit, n_valid, K, dst, wmap, Cin, ntap, Cout,
                   (n_map.data_ptr() if n_map is not None else None), ldw, k_off, 1)

    def _wgrad_conv3(self, plan, tag, dy, ldy, u, F, Hh, Ww, Cin, Cout, dst, n_map=None, up_args=None, src=None, dbias=None):
        """LDS-tiled 3x3 conv weight gradient (decoder blocks / output head) + its deterministic reduction.
        up_args: the block's forward descriptor — the split-f16 kernel then interpolates its operand from the block's own sources
        (gcpx_wgrad_conv3x3_split_up) and `u` is not read"""
        lib, m = self.m.lib, self.m
        N16 = _c16(Cout)
        ych = Cin // 32 if (Cin % 32 == 0 and N16 != 112) else Cin // 16
        ntiles = F * max(1, (Hh * Ww) // 64)
        # workgroups per CU that are resident at once (registers): 2 x 4 wavefronts for the 112-column head and the 64-column block (about 200 registers), 3 otherwise
        per_cu = 2 if N16 in (112, 64) else 3
        grid = max(1, min((lib.gcpx_conv_grid() // 2) * per_cu // ych, ntiles))
        part = m._buf(f"bw.part:{tag}", (grid, N16, 9 * Cin))
        # split-f16 kernel (f32-equivalent, csrc/wgrad_conv_split.hip) unless the model runs on the exact f32 kernels (GCPX_EXACT_F32)
        fn = lib.gcpx_wgrad_conv3x3_split if (m.split_f16 and self.split_wgrad) else lib.gcpx_wgrad_conv3x3
        if up_args is not None:
            self._side(plan, f"bw.wgrad:{tag}", lib.gcpx_wgrad_conv3x3_split_up, dy, ldy, C.byref(up_args), Cout, part.data_ptr(), grid)
        elif src is not None:      # (raw tensor, frame map, scale, shift): operand = LeakyReLU(scale * x + shift) at the mapped frames
            bpart = None
            if dbias is not None:  # the bias gradient (column sums of dy) out of the same launch
                bpart = m._buf(f"bw.bpart:{tag}", (grid, N16))
            self._side(plan, f"bw.wgrad:{tag}", lib.gcpx_wgrad_conv3x3_split_src, dy, ldy, *src, F, Hh, Ww, Cin, Cout, part.data_ptr(),
                       rt.ptr(bpart), grid)
            if bpart is not None:
                self._side(plan, f"bw.creduce:{tag}", lib.gcpx_wgrad_reduce, bpart.data_ptr(), grid, N16, 1, dbias, rt.WMAP_CONV, 1, 1, 0,
                           (n_map.data_ptr() if n_map is not None else None), 0, 0, 1)
        else:
            self._side(plan, f"bw.wgrad:{tag}", fn, dy, ldy, u, F, Hh, Ww, Cin, Cout, part.data_ptr(), grid)
        self._side(plan, f"bw.wreduce:{tag}", lib.gcpx_wgrad_reduce, part.data_ptr(), grid, N16, 9 * Cin, dst, rt.WMAP_CONV, Cin, 9, 0,
                   (n_map.data_ptr() if n_map is not None else None), 0, 0, 1)

    def _colsum(self, plan, tag, dy, ldy, R, N, dst, dst2=None, dy_rpb=0, dy_sb=0, n_map=None):
        lib, m = self.m.lib, self.m
        # a workgroup covers 256 / max(1, N / 4 rounded up to a power of two) rows per iteration: give every chunk ~16 iterations
        tpr = 1
        while tpr < N // 4 and tpr < 256:
            tpr *= 2
        nsplit = max(1, min(1024, R // (16 * (256 // tpr))))
        if nsplit == 1 and n_map is None:
            self._side(plan, f"bw.colsum:{tag}", lib.gcpx_colsum, dy, ldy, R, N, dy_rpb, dy_sb, 1, None, dst, dst2, 1)
            return
        nsplit = max(nsplit, 2)
        part = m._buf(f"bw.cpart:{tag}", (nsplit, N))
        self._side(plan, f"bw.colsum:{tag}", lib.gcpx_colsum, dy, ldy, R, N, dy_rpb, dy_sb, nsplit, part.data_ptr(), None, None, 0)
        if n_map is None:
            self._side(plan, f"bw.creduce:{tag}", lib.gcpx_reduce_partials, part.data_ptr(), nsplit, N, N, dst, 1)
            if dst2 is not None:
                self._side(plan, f"bw.creduce2:{tag}", lib.gcpx_reduce_partials, part.data_ptr(), nsplit, N, N, dst2, 1)
        else:   # bias of the output head: kernel slot -> canonical channel
            assert dst2 is None
            self._side(plan, f"bw.creduce:{tag}", lib.gcpx_wgrad_reduce, part.data_ptr(), nsplit, N, 1, dst, rt.WMAP_CONV, 1, 1, 0,
                       n_map.data_ptr(), 0, 0, 1)

    def _dgemm(self, plan, tag, srcs, M, N, rpb, wpk, out, ob, orow, batch=None, lstm_bwd=None):
        """data-gradient GEMM: out = concat(srcs) @ packed(W^T).  lstm_bwd: LstmBwdArgs of the LSTM layer this gradient is the d h of —
        its cell backward then runs in the GEMM's epilogue (gcpx_gemm_args.lstm_bwd) instead of a launch of its own."""
        dev = None
        if lstm_bwd is not None:
            t = torch.frombuffer(bytearray(bytes(lstm_bwd)), dtype=torch.uint8).to(self.m.device)
            plan.keep += [t, lstm_bwd]
            dev = t.data_ptr()
        self.m._gemm(plan, f"bw.dgrad:{tag}", srcs, M, N, rpb, wpk, None, out=out, ob=ob, orow=orow, batch=batch, lstm_bwd=dev)

    def _dense(self, ptr, ld, width, M):
        return self.m._rowsrc(ptr, M * ld, ld, width)

    def _bn_bwd(self, plan, tag, bn, da, ldc, c_off, up, r, F, Hh, Ww, add=None, fused=None, defer_affine=False, skip=None):
        """activation + BatchNorm backward of one conv block: returns the buffer holding d(raw conv output).
        fused = (dy, partial sums [nb][2][C], nb): the data-gradient conv that produced `da` already applied the activation's derivative
        and left the statistics (gcpx_conv_args.bwd_r): only the BatchNorm half remains, in place.
        skip = (ds, channel offset, channels, frames per sequence): `da` also holds the gradient of a skip connection's channels, whose sum
        over a sequence's frames comes out of the same pass (gcpx_act_skip_bwd)."""
        m, lib, hp = self.m, self.m.lib, self.m._hp
        Cc = bn["C"]
        if fused is not None:
            dy, st, nb = fused
        else:
            dy = m._buf(f"bw.dy:{tag}", (F, Hh, Ww, Cc))
            nb = lib.gcpx_act_bwd_blocks()
            st = m._buf(f"bw.st:{tag}", (nb, 2, Cc))
            a = rt.ActBwdArgs()
            a.da, a.add, a.r = da, (add.data_ptr() if add is not None else None), r.data_ptr()
            a.scale, a.shift, a.mean, a.rstd = bn["scale"].data_ptr(), bn["shift"].data_ptr(), bn["mean"].data_ptr(), bn["rstd"].data_ptr()
            a.dy, a.stats_partial, a.ldc, a.c_off, a.up, a.fsum, a.act = dy.data_ptr(), st.data_ptr(), ldc, c_off, up, 1, rt.ACT_LRELU
            a.F, a.H, a.W, a.C = F, Hh, Ww, Cc
            plan.keep.append(a)
            if skip is not None:
                ds, c_off_s, Cs, rpb_s = skip
                plan.add(f"bw.act+skip:{tag}", lib.gcpx_act_skip_bwd, C.byref(a), ds.data_ptr(), c_off_s, Cs, rpb_s)
            else:
                plan.add(f"bw.act:{tag}", lib.gcpx_act_bwd, C.byref(a))
        coef = m._buf(f"bw.coef:{tag}", (3, Cc))
        pre = bn["prefix"]
        if defer_affine:
            # this chain runs beside other chains that accumulate into the same d gamma / d beta (the three encoder passes): the sums go
            # to a scratch pair and are added with the pass's weight gradients, in order on one lane
            dgb = m._buf(f"bw.dgb:{tag}", (2, Cc))
            plan.add(f"bw.bnfin:{tag}", lib.gcpx_bn_bwd_finalize, st.data_ptr(), nb, Cc, C.c_double(float(F * Hh * Ww)),
                     m.sd[f"{pre}.weight"].data_ptr(), bn["rstd"].data_ptr(), coef.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(), 0)
            self._side(plan, f"bw.bnacc:{tag}.g", lib.gcpx_reduce_partials, dgb[0].data_ptr(), 1, Cc, Cc, self.g(f"{pre}.weight"), 1)
            self._side(plan, f"bw.bnacc:{tag}.b", lib.gcpx_reduce_partials, dgb[1].data_ptr(), 1, Cc, Cc, self.g(f"{pre}.bias"), 1)
        else:
            plan.add(f"bw.bnfin:{tag}", lib.gcpx_bn_bwd_finalize, st.data_ptr(), nb, Cc, C.c_double(float(F * Hh * Ww)),
                     m.sd[f"{pre}.weight"].data_ptr(), bn["rstd"].data_ptr(), coef.data_ptr(), self.g(f"{pre}.weight"),
                     self.g(f"{pre}.bias"), 1)
        plan.add(f"bw.bnapply:{tag}", lib.gcpx_bn_bwd_apply, dy.data_ptr(), r.data_ptr(), bn["mean"].data_ptr(),
                 bn["rstd"].data_ptr(), coef.data_ptr(), F * Hh * Ww * Cc, Cc)
        return dy

    def _mlp_bwd_group(self, plan, tag, group):
        """the launches `_mlp_bwd(..., group=group)` held back: one grouped launch when they share a hidden width, else one each"""
        if not group:
            return
        lib = self.m.lib
        if len(group) > 1 and len(group) <= 4 and len({a.mid for _, a in group}) == 1:
            tab = (rt.MlpBwdArgs * len(group))(*[a for _, a in group])
            plan.keep.append(tab)
            plan.add(f"bw.mlp:{tag}", lib.gcpx_mlp_bwd_group, tab, len(group))
        else:
            for t, a in group:
                plan.add(f"bw.mlp:{t}", lib.gcpx_mlp_bwd, C.byref(a))

    def _mlp_bwd(self, plan, tag, prefix, rec, T, dout, ldo, dx_outs, group=None):
        """Backward of one Predictor MLP.  dout: dense [M][ldo] gradient of the head output (pad columns zero).
        dx_outs: one (out_ptr, ob, orow) per input split packed in T (wT_in{i}); rows (b, j) with the forward's rpb."""
        m, lib, hp = self.m, self.m.lib, self.m._hp
        W, srcs, M, rpb, save = rec["W"], rec["srcs"], rec["M"], rec["rpb"], rec["save"]
        mid, n_mid, out_dim, in_dim = W["mid"], W["n_mid"], W["out_dim"], W["in_dim"]
        out_pad = _c16(out_dim)
        assert ldo == out_pad
        sv = lambda i: save.data_ptr() + 4 * i * M * mid
        a_ptr = [sv(0)] + [sv(2 + 2 * l) for l in range(n_mid)]
        u_ptr = [sv(1 + 2 * l) for l in range(n_mid)]
        # head
        self._wgrad(plan, f"{tag}.out", dout, ldo, M, out_pad, a_ptr[n_mid], mid, self.g(f"{prefix}.head.linear.weight"),
                    ldw=mid, n_valid=out_dim, sr=mid, sb=M * mid, rpb=M, dbias=self.g(f"{prefix}.head.linear.bias"))
        if self.fused_mlp_bwd and mid in (128, 32) and n_mid <= 4 and len(dx_outs) <= 4 and out_pad <= 1024 and \
                all(ob % 4 == 0 and orow % 4 == 0 for _, ob, orow in dx_outs):
            self._mlp_bwd_fused(plan, tag, prefix, rec, T, dout, ldo, dx_outs, a_ptr, group=group)
            return
        da = m._buf(f"bw.{tag}.da{n_mid}", (M, mid))
        self._dgemm(plan, f"{tag}.out", [self._dense(dout, ldo, out_pad, M)], M, mid, M, T["wT_out"], da.data_ptr(), 0, mid)
        for l in reversed(range(n_mid)):
            nb = lib.gcpx_gn_bwd_blocks(M)
            part = m._buf(f"bw.{tag}.gnpart{l}", (nb, 2, mid))
            du = m._buf(f"bw.{tag}.du{l + 1}", (M, mid))
            pre = f"{prefix}.pyramid-{l}"
            plan.add(f"bw.gn:{tag}.{l}", lib.gcpx_gn_lrelu_bwd, u_ptr[l], da.data_ptr(), m.sd[f"{pre}.norm.weight"].data_ptr(),
                     m.sd[f"{pre}.norm.bias"].data_ptr(), du.data_ptr(), part.data_ptr(), M, mid, hp.gn_groups,
                     C.c_float(hp.gn_eps), C.c_float(hp.leaky_slope))
            self._gn_param_grads(plan, f"{tag}.{l}", pre, part, nb, mid)
            self._wgrad(plan, f"{tag}.mid{l}", du.data_ptr(), mid, M, mid, a_ptr[l], mid, self.g(f"{pre}.linear.weight"), ldw=mid,
                        sr=mid, sb=M * mid, rpb=M, dbias=self.g(f"{pre}.linear.bias"))
            da = m._buf(f"bw.{tag}.da{l}", (M, mid))
            self._dgemm(plan, f"{tag}.mid{l}", [self._dense(du.data_ptr(), mid, mid, M)], M, mid, M, T[f"wT_mid{l}"], da.data_ptr(), 0, mid)
        du0 = m._buf(f"bw.{tag}.du0", (M, mid))
        plan.add(f"bw.lrelu:{tag}", lib.gcpx_lrelu_bwd, a_ptr[0], da.data_ptr(), du0.data_ptr(), M * mid, C.c_float(hp.leaky_slope))
        koff = 0
        w_in_dst = self._mlp_in_dst(plan, tag, prefix, W)
        for i, s in enumerate(srcs):
            self._wgrad(plan, f"{tag}.in{i}", du0.data_ptr(), mid, M, mid, s.ptr, s.width, w_in_dst,
                        ldw=in_dim, k_off=koff, rpb=rpb, sb=s.sb, sr=s.sr, shift=s.shift,
                        rowidx=_PtrHolder(s.rowidx) if s.rowidx else None,
                        dbias=(self.g(f"{prefix}.input.linear.bias") if i == 0 else None))
            koff += s.width
        for i, (optr, ob, orow) in enumerate(dx_outs):
            wT = T[f"wT_in{i}"]
            width = wT.shape[1] * 16
            self._dgemm(plan, f"{tag}.in{i}", [self.m._rowsrc(du0.data_ptr(), rpb * mid, mid, mid)], M, width, rpb, wT, optr, ob, orow)

    def _gn_param_grads(self, plan, tag, pre, part, nb, mid):
        """GroupNorm gamma / beta gradients of one Predictor layer from the per-workgroup partials [nb][2][mid].  The two parameters are
        neighbours in the flat vector (params._predictor lists weight, then bias), so ONE reduction over 2 * mid columns writes both —
        60 launches of ~4 us less on the side lanes of a c2 step than one reduction each."""
        lib, off = self.m.lib, self.m._poff
        if off[f"{pre}.norm.bias"][0] == off[f"{pre}.norm.weight"][0] + mid:
            self._side(plan, f"bw.gnred:{tag}", lib.gcpx_reduce_partials, part.data_ptr(), nb, 2 * mid, 2 * mid, self.g(f"{pre}.norm.weight"), 1)
            return
        self._side(plan, f"bw.gnred:{tag}", lib.gcpx_reduce_partials, part.data_ptr(), nb, 2 * mid, mid, self.g(f"{pre}.norm.weight"), 1)
        self._side(plan, f"bw.gnred2:{tag}", lib.gcpx_reduce_partials, part.data_ptr() + 4 * mid, nb, 2 * mid, mid, self.g(f"{pre}.norm.bias"), 1)

    def _mlp_in_dst(self, plan, tag, prefix, W):
        """Where the weight gradient of a Predictor's input layer is accumulated: the parameter's gradient itself — unless the layer's
        input was padded to a 16-column k-group (model._pack_predictor: the action encoder's n_actions columns): then rows of the padded
        width in a scratch block, whose first in_dim_raw columns `_unpad_input_grads` copies behind the last side lane."""
        raw = W.get("in_dim_raw", W["in_dim"])
        if raw == W["in_dim"]:
            return self.g(f"{prefix}.input.linear.weight")
        scratch = self.m._buf(f"bw.{tag}.dW_in", (W["mid"], W["in_dim"]))
        plan.add("bw.zero", self.m.lib.gcpx_fill_zero, scratch.data_ptr(), scratch.numel() * 4)
        self._pad_fixups.append((tag, self.g(f"{prefix}.input.linear.weight"), raw, scratch, W["in_dim"], W["mid"]))
        return scratch.data_ptr()

    def _unpad_input_grads(self, plan):
        for tag, dst, raw, scratch, pad, mid in self._pad_fixups:
            plan.add(f"bw.unpad:{tag}", self.m.lib.gcpx_rows_strided, dst, 0, raw, scratch.data_ptr(), 0, pad, 1, mid, raw, 0)
        self._pad_fixups = []

    def _mlp_bwd_fused(self, plan, tag, prefix, rec, T, dout, ldo, dx_outs, a_ptr, group=None):
        """The data-gradient chain of one Predictor as ONE launch (gcpx_mlp_bwd); weight gradients and the GroupNorm parameter
        reductions stay on the side lanes (the head's weight gradient was queued by the caller)."""
        m, lib, hp = self.m, self.m.lib, self.m._hp
        W, srcs, M, rpb, save = rec["W"], rec["srcs"], rec["M"], rec["rpb"], rec["save"]
        mid, n_mid, in_dim = W["mid"], W["n_mid"], W["in_dim"]
        nb = lib.gcpx_mlp_bwd_blocks(M)
        a = rt.MlpBwdArgs()
        a.dout, a.save, a.wT_out, a.ldo = dout, save.data_ptr(), T["wT_out"].data_ptr(), ldo
        a.M, a.rpb, a.mid, a.n_mid, a.out_pad, a.ndx = M, rpb, mid, n_mid, _c16(W["out_dim"]), len(dx_outs)
        a.gn_eps, a.lrelu_slope = hp.gn_eps, hp.leaky_slope
        du = [m._buf(f"bw.{tag}.du{l}", (M, mid)) for l in range(n_mid + 1)]
        parts = [m._buf(f"bw.{tag}.gnpart{l}", (nb, 2, mid)) for l in range(n_mid)]
        a.du[0] = du[0].data_ptr()
        for l in range(n_mid):
            pre = f"{prefix}.pyramid-{l}"
            a.wT_mid[l] = T[f"wT_mid{l}"].data_ptr()
            a.gn_gamma[l], a.gn_beta[l] = m.sd[f"{pre}.norm.weight"].data_ptr(), m.sd[f"{pre}.norm.bias"].data_ptr()
            a.du[1 + l], a.gn_partial[l] = du[1 + l].data_ptr(), parts[l].data_ptr()
        for i, (optr, ob, orow) in enumerate(dx_outs):
            wT = T[f"wT_in{i}"]
            a.dx[i].wT, a.dx[i].out, a.dx[i].ob, a.dx[i].orow, a.dx[i].width = wT.data_ptr(), optr, ob, orow, wT.shape[1] * 16
        plan.keep.append(a)
        if group is not None:
            group.append((tag, a))               # issued by _mlp_bwd_group (the queued weight gradients below go out with a later flush)
        else:
            plan.add(f"bw.mlp:{tag}", lib.gcpx_mlp_bwd, C.byref(a))
        for l in reversed(range(n_mid)):
            pre = f"{prefix}.pyramid-{l}"
            self._gn_param_grads(plan, f"{tag}.{l}", pre, parts[l], nb, mid)
            self._wgrad(plan, f"{tag}.mid{l}", du[1 + l].data_ptr(), mid, M, mid, a_ptr[l], mid, self.g(f"{pre}.linear.weight"), ldw=mid,
                        sr=mid, sb=M * mid, rpb=M, dbias=self.g(f"{pre}.linear.bias"))
        koff = 0
        w_in_dst = self._mlp_in_dst(plan, tag, prefix, W)
        for i, s in enumerate(srcs):
            self._wgrad(plan, f"{tag}.in{i}", du[0].data_ptr(), mid, M, mid, s.ptr, s.width, w_in_dst,
                        ldw=in_dim, k_off=koff, rpb=rpb, sb=s.sb, sr=s.sr, shift=s.shift,
                        rowidx=_PtrHolder(s.rowidx) if s.rowidx else None,
                        dbias=(self.g(f"{prefix}.input.linear.bias") if i == 0 else None))
            koff += s.width

    # ------------------------------------------------------------------------------------------------
    # the backward plan
    # ------------------------------------------------------------------------------------------------
    def _build_backward(self, fplan):
        m, hp, lib = self.m, self.m._hp, self.m.lib
        rec, o = fplan.rec, fplan.outs
        key, tin = rec["key"], rec["tin"]
        B = key[0]
        L, T, N = hp.hierarchy_levels, hp.max_seq_len, hp.n_nodes
        nz, nv, H, SD, nl = hp.nz_enc, hp.nz_vae, hp.nz_mid_lstm, hp.lstm_state_dim, hp.n_lstm_layers
        PS = 2 ** L + 1
        S = hp.img_sz
        pitch = m._head_pitch
        div = float(T * hp.input_nc * S * S)
        plan = _Plan(lib)
        E, Hid, QZ, PZ = o["E"], o["Hid"], o["QZ"], o["PZ"]
        buf = m._buf
        zero = lambda t: plan.add("bw.zero", lib.gcpx_fill_zero, t.data_ptr(), t.numel() * 4)

        dE, dHid = buf("bw.dE", (B, PS, nz)), buf("bw.dHid", (B, PS, SD))
        dET = buf("bw.dET", (B, PS, nz))
        dQZ, dPZ = buf("bw.dQZ", (B, PS, 2 * nv)), buf("bw.dPZ", (B, PS, 2 * nv))
        zero(dE); zero(dHid)
        if self.zero_on_side_lane and self.side_lanes and self.n_side >= 2 and not hp.adaptive:
            # the 293 MB gradient vector is cleared on lane 2 (lane 1 waits for it once; lane 0's first gradient write — the BatchNorm
            # parameter sums behind the head's data gradient — waits there, _decoder_backward): 54 us less in front of the decoder's chain
            plan.rec["zero_on_lane2"] = True
            plan.fork([2])
            plan.lane = 2
            zero(self.grad)
            plan.lane = 0
            plan.wait(1, 2)
        else:
            zero(self.grad)

        # ---- loss gradients (base_gcp.py:264-304) ----
        la = rec["loss_args"]
        adaptive, attentive = hp.adaptive, hp.attentive_inference
        if adaptive:
            # LossAveragingCriterion (binding_loss.py:19-42): gradient w.r.t. the decoded image of EVERY node, then back through
            # the mixture mean to the head's raw parameters; the matching weights are constants (adaptive.py:50 detaches)
            Dd = hp.input_nc * S * S
            dImg = buf("bw.dImg", (B, N, hp.input_nc, S, S))
            plan.add("bw.avg_nll", lib.gcpx_averaging_nll_bwd, o["match_dist_df"].data_ptr(), tin["pad_mask"].data_ptr(),
                     o["images_df"].data_ptr(), tin["traj_seq"].data_ptr(), o["cdist_sum"].data_ptr(),
                     m.sd["decoder.log_sigma"].data_ptr(), C.c_float(hp.dense_img_rec_weight / (B * div)), B, N, T, Dd, dImg.data_ptr(),
                     self.g("decoder.log_sigma"))
            dMD = buf("bw.dMD", (B * N, S, S, pitch))
            plan.add("bw.dlm_mean", lib.gcpx_dlm_mean_bwd, o["distr_df_kernel_order"].data_ptr(), dImg.data_ptr(), dMD.data_ptr(),
                     buf("bw.dMD.colsum", (B * N, pitch)).data_ptr(), B * N, S * S, pitch, hp.n_mixtures)
        else:
            dMD = buf("bw.dMD", (B * T, S, S, pitch))
            if not rec.get("nll_bwd_fused"):        # otherwise the forward plan already produced dMD together with the loss
                md = o["matched_distr_kernel_order"]
                plan.add("bw.dlm_nll", lib.gcpx_dlm_nll_bwd, md.data_ptr(), tin["traj_seq"].data_ptr(), tin["pad_mask"].data_ptr(),
                         C.c_float(hp.dense_img_rec_weight / (B * div)), dMD.data_ptr(), buf("bw.dMD.colsum", (B * T, pitch)).data_ptr(),
                         None, B * T, S * S, pitch, hp.n_mixtures)
        # The latent-space heads (KL, length / existence / state / inverse-model / cost Predictors: ~10 small launches, 0.25 ms on an
        # otherwise idle chip) run on side lane 1 beside the decoder's data-gradient chain, which needs none of their results; lane 0
        # picks them up where the decoder's gradient meets dE (bw.addrows below)
        heads_aside = self.heads_on_side_lane and self.side_lanes and self.n_side >= 1 and not adaptive
        if heads_aside:
            plan.fork([1])
            plan.lane = 1
        if m._kl_w is not None:                               # burn-in schedule: kl_weight(step) is read from device memory
            plan.add("bw.kl", lib.gcpx_kl_bwd_scheduled, _addr(QZ, 2 * nv), _addr(PZ, 2 * nv), _addr(dQZ, 2 * nv), _addr(dPZ, 2 * nv), B, N, nv,
                     PS * 2 * nv, 2 * nv, C.c_float(hp.free_nats), C.c_float(1.0 / (B * div)), None, 0, m._kl_w.data_ptr())
        else:
            plan.add("bw.kl", lib.gcpx_kl_bwd, _addr(QZ, 2 * nv), _addr(PZ, 2 * nv), _addr(dQZ, 2 * nv), _addr(dPZ, 2 * nv), B, N, nv,
                     PS * 2 * nv, 2 * nv, C.c_float(hp.free_nats), C.c_float(hp.kl_weight / (B * div)))
        ldl = _c16(T)
        dlen = buf("bw.dlen", (B, ldl)) if hp.regress_length else None
        Nex = N - 1 if adaptive else N                  # adaptive: the BCE is over the N - 1 consecutive-node pairs (adaptive.py:118-122)
        dexist = buf("bw.dexist", (B * Nex, 16))
        has_state = "regressed_state_padded" in o and "traj_seq_states" in tin
        dstate = buf("bw.dstate", (B * T, 16)) if has_state else None
        plan.add("bw.heads", lib.gcpx_loss_heads_bwd, C.byref(la), rt.ptr(dlen), dexist.data_ptr(), rt.ptr(dstate))

        # ---- latent-space heads ----
        if hp.regress_length:
            dXl = buf("bw.dX.len", (B, 2 * nz))
            self._mlp_bwd(plan, "length_pred", "length_pred.p", rec["mlp:length_pred"], self.bk["length_pred"], dlen.data_ptr(), ldl,
                          [(dXl.data_ptr(), 2 * nz, 0)])
            self._tree_accum(plan, "len", dE, PS * nz, 2 ** L * nz, B, 1, nz, [(dXl.data_ptr(), 2 * nz, 0, nz, -1, -1, 0)])
        if adaptive:
            dE_d0, dE_d1 = buf("bw.dE_d0", (B * Nex, nz)), buf("bw.dE_d1", (B * Nex, nz))
            self._mlp_bwd(plan, "distance", "tree_module.tree_modules.0.binding.distance_predictor", rec["mlp:distance"],
                          self.bk["distance"], dexist.data_ptr(), 16, [(dE_d0.data_ptr(), Nex * nz, nz), (dE_d1.data_ptr(), Nex * nz, nz)])
        else:
            dE_ex = buf("bw.dE_ex", (B * N, nz))
            self._mlp_bwd(plan, "existence", "tree_module.tree_modules.0.binding.existence_predictor", rec["mlp:existence"],
                          self.bk["existence"], dexist.data_ptr(), 16, [(dE_ex.data_ptr(), N * nz, nz)])
        if has_state:   # input detached (base_gcp.py:253-256): parameter gradients only
            self._mlp_bwd(plan, "state_regressor", "state_regressor", rec["mlp:state_regressor"], self.bk["state_regressor"],
                          dstate.data_ptr(), 16, [])

        # inverse model / cost model: inputs detached (inverse_mdl.py:160-162, cost_mdl.py:108-109): parameter gradients only
        has_inv, has_cost = bool(la.action_pred), bool(la.cost_pred)
        if has_inv or has_cost:
            daction = buf("bw.daction", (B, 16)) if has_inv else None
            dcost = buf("bw.dcost", (B, 16)) if has_cost else None
            plan.add("bw.aux_heads", lib.gcpx_loss_aux_heads_bwd, C.byref(la), rt.ptr(daction), rt.ptr(dcost))
            if has_inv:
                self._mlp_bwd(plan, "inv_mdl", "inv_mdl.action_pred", rec["mlp:inv_mdl"], self.bk["inv_mdl"], daction.data_ptr(), 16, [])
            if has_cost:
                self._mlp_bwd(plan, "cost_mdl", "cost_mdl.cost_pred", rec["mlp:cost_mdl"], self.bk["cost_mdl"], dcost.data_ptr(), 16, [])

        if heads_aside:
            self._flush(plan, only_lane=1)        # their weight gradients follow them on the same lane (they read the Predictors' du)
            plan.lane = 0
        else:
            self._flush(plan)
        # ---- decoder (tree_dense_rec.py:42 backward) ----
        dE_dec, dskip = self._decoder_backward(plan, fplan, dMD, B)
        if heads_aside:
            plan.wait(0, 1)
        held = []
        if self.side_lanes and 0 <= self.dec_side_level < L:
            held, plan.deferred = plan.deferred, []
        else:
            self._flush(plan)
        if adaptive:
            plan.add("bw.addrows", lib.gcpx_add_rows, _addr(dE, nz), PS * nz, nz, dE_dec.data_ptr(), None, B, N, nz)
            # distance predictor inputs were (node p, node p + 1), p < N - 1 (adaptive.py:66-67)
            plan.add("bw.addrows.d0", lib.gcpx_add_rows, _addr(dE, nz), PS * nz, nz, dE_d0.data_ptr(), None, B, Nex, nz)
            plan.add("bw.addrows.d1", lib.gcpx_add_rows, _addr(dE, 2 * nz), PS * nz, nz, dE_d1.data_ptr(), None, B, Nex, nz)
        else:
            plan.add("bw.addrows", lib.gcpx_add_rows, _addr(dE, nz), PS * nz, nz, dE_dec.data_ptr(), dE_ex.data_ptr(), B, N, nz)
        if attentive:
            kv = rec["attn_kv"]
            n_mod, dk = kv["n_mod"], hp.nz_attn_key
            dKp, dVp = buf("bw.dKp", (B * T, n_mod * dk)), buf("bw.dVp", (B * T, n_mod * nz))
            if n_mod < L:       # tied levels accumulate: not built
                raise NotImplementedError("attentive training with tied tree layers")

        # ---- tree levels, leaves first (tree_utils.py:21-44 backward) ----
        MERGE_LANE = 1 + self.n_side                    # the caller's stream (backward(): the last entry of the stream list)
        merge_lane = (self.merge_on_caller_lane and self.side_lanes and bool(hp.tree_lstm) and not self.parallel_level_chains and
                      not (m.use_graph and self.backward_graph))
        plan.rec["caller_lane"] = merge_lane
        merge_pending = False
        pid = hp.pred_inp_dim
        for l in reversed(range(L)):
            li = l if hp.untied_layers else 0
            Wt = self.bk[f"tree{li}"]
            p = f"tree_module.tree_modules.{li}"
            sp = f"{p}.subgoal_pred"
            s, n = 2 ** (L - 1 - l), 2 ** l
            M = B * n
            dEn = _addr(dE, s * nz)
            dpi = buf(f"bw.dpi{l}", (M, pid))
            if not hp.tree_lstm:
                # non-LSTM subgoal predictor: e = tanh(net([e_l, e_r, z (, e_0, e_g)])) (tree_module.py:109-110): d pre-activation, then the
                # Predictor's backward straight into the gradient of the predictor inputs
                dpre = buf(f"bw.dpre{l}", (M, _c16(nz)))
                plan.add(f"bw.tanh{l}", lib.gcpx_tanh_bwd_rows, dEn, _addr(E, s * nz), dpre.data_ptr(), PS * nz, 2 * s * nz, B, n, nz)
                self._mlp_bwd(plan, f"subgoal{l}", f"{sp}.net", rec[f"mlp:subgoal{l}"], Wt["sg"], dpre.data_ptr(), _c16(nz),
                              [(dpi.data_ptr(), n * pid, pid)])
            else:
                # out linear
                x_top = buf(f"x{l}.{nl}", (M, H))
                self._wgrad(plan, f"out{l}", dEn, 2 * s * nz, M, nz, x_top.data_ptr(), H, self.g(f"{sp}.out.weight"), ldw=H, sr=H,
                            sb=M * H, rpb=M, dy_rpb=n, dy_sb=PS * nz, dbias=self.g(f"{sp}.out.bias"))
                dxt = buf(f"bw.dxt{l}", (M, H))
                merged = buf(f"merged{l}", (M, 2 * nl * H))
                dmerged = buf(f"bw.dmerged{l}", (M, 2 * nl * H))
                dgs = buf(f"bw.dgates{l}", (nl, M, 4 * H))
                dxis = [buf(f"bw.dxi{l}.{i}", (M, H)) for i in range(nl)]
                cells = []
                for i in range(nl):
                    dh_src = dxt if i == nl - 1 else dxis[i + 1]
                    a = rt.LstmBwdArgs()
                    a.gates = rec[f"gates:lstm{l}.{i}"].data_ptr()
                    a.c_prev, a.c_prev_stride = _addr(merged, (2 * i + 1) * H), 2 * nl * H
                    a.c_new, a.pb, a.prow = _addr(Hid, s * SD + (2 * i + 1) * H), PS * SD, 2 * s * SD
                    a.dh_dense, a.dh_stride = dh_src.data_ptr(), H
                    a.dh_pos, a.dc_pos = _addr(dHid, s * SD + 2 * i * H), _addr(dHid, s * SD + (2 * i + 1) * H)
                    a.dgates, a.dc_prev, a.dcp_stride = dgs[i].data_ptr(), _addr(dmerged, (2 * i + 1) * H), 2 * nl * H
                    a.M, a.H, a.rpb = M, H, n
                    plan.keep.append(a)
                    cells.append(a)
                # each layer's cell backward rides in the epilogue of the GEMM that produces its d h (fuse_lstm_bwd): 3 launches per level
                # less on the chain
                fuse_cell = self.fuse_lstm_bwd
                if merge_pending:
                    # the level above wrote this level's d state (dHid) on the merge lane: the first cell backward reads it
                    if not fuse_cell:
                        self._dgemm(plan, f"out{l}", [m._rowsrc(dEn, PS * nz, 2 * s * nz, nz)], M, H, n, Wt["out.wT"], dxt.data_ptr(), n * H, H)
                    plan.wait(0, MERGE_LANE)
                    merge_pending = False
                    if fuse_cell:
                        self._dgemm(plan, f"out{l}", [m._rowsrc(dEn, PS * nz, 2 * s * nz, nz)], M, H, n, Wt["out.wT"], dxt.data_ptr(), n * H, H,
                                    lstm_bwd=cells[nl - 1])
                else:
                    self._dgemm(plan, f"out{l}", [m._rowsrc(dEn, PS * nz, 2 * s * nz, nz)], M, H, n, Wt["out.wT"], dxt.data_ptr(), n * H, H,
                                lstm_bwd=(cells[nl - 1] if fuse_cell else None))
                for i in reversed(range(nl)):
                    dg = dgs[i]
                    if not fuse_cell:
                        plan.add(f"bw.lstm{l}.{i}", lib.gcpx_lstm_bwd, C.byref(cells[i]))
                    x_i = buf(f"x{l}.{i}", (M, H))
                    self._wgrad(plan, f"lstm{l}.{i}.ih", dg.data_ptr(), 4 * H, M, 4 * H, x_i.data_ptr(), H,
                                self.g(f"{sp}.lstm.{i}.weight_ih"), ldw=H, sr=H, sb=M * H, rpb=M,
                                dbias=self.g(f"{sp}.lstm.{i}.bias_ih"), dbias2=self.g(f"{sp}.lstm.{i}.bias_hh"))
                    self._wgrad(plan, f"lstm{l}.{i}.hh", dg.data_ptr(), 4 * H, M, 4 * H, _addr(merged, 2 * i * H), H,
                                self.g(f"{sp}.lstm.{i}.weight_hh"), ldw=H, sr=2 * nl * H, sb=M * 2 * nl * H, rpb=M)
                    dxi = dxis[i]
                    src = [self._dense(dg.data_ptr(), 4 * H, 4 * H, M)]
                    self._dgemm(plan, f"lstm{l}.{i}.x", src, M, H, M, Wt[f"lstm{i}.wxT"], dxi.data_ptr(), 0, H,
                                lstm_bwd=(cells[i - 1] if (fuse_cell and i > 0) else None))
                    if not self.batch_dh:
                        self._dgemm(plan, f"lstm{l}.{i}.h", src, M, H, M, Wt["lstm.whT"][i], _addr(dmerged, 2 * i * H), 0, 2 * nl * H)
                    dh_src = dxi
                dx0 = dh_src
                if merge_lane:
                    plan.fork([MERGE_LANE])            # the merge chain (below) starts here: all gates' gradients are out
                def dh_batched():
                    # d h_prev of every layer (wanted by the merge only): one launch, blockIdx.z = layer — the level's chain is 2 of its
                    # 6 LSTM data-gradient GEMMs shorter, and the launch has nl times the workgroups of one (16 .. 256 rows below level 5)
                    if self.batch_dh:
                        self._dgemm(plan, f"lstm{l}.h", [self._dense(dgs.data_ptr(), 4 * H, 4 * H, M)], M, H, M, Wt["lstm.whT"],
                                    dmerged.data_ptr(), 0, 2 * nl * H, batch=(nl, M * 4 * H, Wt["lstm.whT"][0].numel(), 0, 2 * H))
                # embedding of [e_l, e_r, z, e_0, e_g]
                el = m._rowsrc(_addr(E), PS * nz, 2 * s * nz, nz)
                er = m._rowsrc(_addr(E, 2 * s * nz), PS * nz, 2 * s * nz, nz)
                zs = m._rowsrc(_addr(o["Z"], s * nv), PS * nv, 2 * s * nv, nv)
                esrcs = [el, er, zs]
                if hp.context_every_step:
                    esrcs += [m._rowsrc(_addr(E), PS * nz, 0, nz), m._rowsrc(_addr(E, 2 ** L * nz), PS * nz, 0, nz)]
                koff = 0
                for i, sc in enumerate(esrcs):
                    self._wgrad(plan, f"embed{l}.{i}", dx0.data_ptr(), H, M, H, sc.ptr, sc.width, self.g(f"{sp}.embed.weight"), ldw=pid,
                                k_off=koff, rpb=n, sb=sc.sb, sr=sc.sr, dbias=(self.g(f"{sp}.embed.bias") if i == 0 else None))
                    koff += sc.width
                dpi = buf(f"bw.dpi{l}", (M, pid))
                self._dgemm(plan, f"embed{l}", [self._dense(dx0.data_ptr(), H, H, M)], M, pid, M, Wt["embed.wT"], dpi.data_ptr(), 0, pid)
            def merge_backward():
                if not hp.tree_lstm:
                    return
                if hp.tree_lstm == "sum":
                    # SumTree (tree_lstm.py:14-16): the gradient of the merged state goes to both parents unchanged
                    self._tree_accum(plan, f"hid{l}", dHid, PS * SD, 2 * s * SD, B, n, SD, [(dmerged.data_ptr(), SD, 0, 0, -1, -1, 0)])
                    return
                if hp.tree_lstm == "linear":
                    # LinTree (tree_lstm.py:25-27): one Linear over [hidden_left | hidden_right]
                    for side, base in ((0, 0), (1, 2 * s * SD)):
                        self._wgrad(plan, f"proj{l}.{side}", dmerged.data_ptr(), SD, M, SD, _addr(Hid, base), SD,
                                    self.g(f"{sp}.projection.weight"), ldw=2 * SD, k_off=side * SD, rpb=n, sb=PS * SD, sr=2 * s * SD,
                                    dbias=(self.g(f"{sp}.projection.bias") if side == 0 else None))
                    dpar = buf(f"bw.dpar{l}", (M, 2 * SD))
                    self._dgemm(plan, f"merge{l}", [self._dense(dmerged.data_ptr(), SD, SD, M)], M, 2 * SD, M, Wt["proj.wT"], dpar.data_ptr(), 0, 2 * SD)
                    self._tree_accum(plan, f"hid{l}", dHid, PS * SD, 2 * s * SD, B, n, SD, [(dpar.data_ptr(), 2 * SD, 0, SD, -1, -1, 0)])
                    return
                # split_linear merge of the parents' hidden states
                # all 2*n_lstm_layers projections in one launch per parent side (blockIdx.z = projection)
                po = [m._poff[f"{sp}.projections.{j}.weight"][0] for j in range(2 * nl)]
                bo = [m._poff[f"{sp}.projections.{j}.bias"][0] for j in range(2 * nl)]
                zw, zb = po[1] - po[0], bo[1] - bo[0]
                assert all(po[j + 1] - po[j] == zw and bo[j + 1] - bo[j] == zb for j in range(2 * nl - 1))
                for side, base in ((0, 0), (1, 2 * s * SD)):
                    self._wgrad(plan, f"proj{l}.{side}", dmerged.data_ptr(), 2 * nl * H, M, H, _addr(Hid, base), H,
                                self.g(f"{sp}.projections.0.weight"), ldw=2 * H, k_off=side * H, rpb=n, sb=PS * SD, sr=2 * s * SD,
                                dbias=(self.g(f"{sp}.projections.0.bias") if side == 0 else None), batch=(2 * nl, H, H, zw, zb))
                dpar = buf(f"bw.dpar{l}", (2 * nl, M, 2 * H))
                self._dgemm(plan, f"merge{l}", [m._rowsrc(dmerged.data_ptr(), n * 2 * nl * H, 2 * nl * H, H)], M, 2 * H, n, Wt["proj.wT"],
                            dpar.data_ptr(), n * 2 * H, 2 * H, batch=(2 * nl, H, Wt["proj.wT"][0].numel(), 0, M * 2 * H))
                self._tree_accum(plan, f"hid{l}", dHid, PS * SD, 2 * s * SD, B, n, H,
                                 [(dpar.data_ptr() + 4 * j * M * 2 * H, 2 * H, 0, H, -1, -1, j * H) for j in range(2 * nl)])

            # Below the root the three remaining chains of a level are independent: posterior MLP (+ attention), prior MLP,
            # parent-state merge.  They run on three lanes (each ~10 dependent launches) and meet again at the accumulation.
            split = self.parallel_level_chains and self.side_lanes and l > 0
            dXi = None
            if not split:
                if hp.tree_lstm:
                    if merge_lane:
                        plan.lane = MERGE_LANE
                    dh_batched()
                merge_backward()
                if hp.tree_lstm and merge_lane:
                    plan.lane = 0
                    merge_pending = True
                    if l == 0:
                        plan.wait(0, MERGE_LANE)        # (the LSTM initialiser below reads the root states' gradient)
                        merge_pending = False
            if l == 0 and hp.tree_lstm and hp.lstm_init == "mlp":
                # MLP LSTM initialiser (tree_module.py:104-105): outputs live in Hid slots 0 and 2^L
                dinit = buf("bw.dinit", (B, 2 * SD))
                plan.add("bw.dinit.l", lib.gcpx_copy_rows, _addr(dHid), dinit.data_ptr(), B, 1, SD, PS, 2)
                plan.add("bw.dinit.r", lib.gcpx_copy_rows, _addr(dHid, 2 ** L * SD), _addr(dinit, SD), B, 1, SD, PS, 2)
                dXi = buf("bw.dX.init", (B, 2 * nz + nv))
                self._mlp_bwd(plan, "lstm_init", f"{p}.lstm_initializer.net", rec["mlp:lstm_init"], Wt["init"], dinit.data_ptr(),
                              2 * SD, [(dXi.data_ptr(), (2 * nz + nv), 0)])
            # sampled latent: z = mu_q + exp(log_sigma_q) * eps (tree_module.py:86-94)
            dq, dp = buf(f"bw.dq{l}", (M, 2 * nv)), buf(f"bw.dp{l}", (M, 2 * nv))
            plan.add(f"bw.latent{l}", lib.gcpx_latent_bwd, _addr(dQZ, s * 2 * nv), _addr(dPZ, s * 2 * nv), _addr(QZ, s * 2 * nv),
                     PS * 2 * nv, 2 * s * 2 * nv, _addr(tin["eps"], (n - 1) * nv), N * nv, nv, _addr(dpi, 2 * nz), pid,
                     (_addr(dXi, 2 * nz) if dXi is not None else None), 2 * nz + nv, dq.data_ptr(), dp.data_ptr(), M, n, nv)
            dXq, dXp = buf(f"bw.dXq{l}", (M, 2 * nz)), buf(f"bw.dXp{l}", (M, 2 * nz))
            dEt_l = buf(f"bw.dEt{l}", (M, nz)) if attentive else None
            et_out = (dEt_l.data_ptr(), n * nz, nz) if attentive else (_addr(dET, s * nz), PS * nz, 2 * s * nz)
            if split:
                plan.fork([1, 2])
                plan.lane = 1
            # posterior and prior chains are independent: one grouped launch of both (not with attention, whose backward sits between
            # them and reads the posterior's result; not on three lanes)
            grp = [] if (self.group_mlp_bwd and not attentive and not split) else None
            self._mlp_bwd(plan, f"posterior{l}", f"{p}.inference.q", rec[f"mlp:posterior{l}"], Wt["q"], dq.data_ptr(), 2 * nv,
                          [(dXq.data_ptr(), n * 2 * nz, 2 * nz), et_out], group=grp)
            dXa = None
            if attentive:
                dXa = self._attention_backward(plan, fplan, l, Wt, dEt_l, dKp, dVp, B)
                # this level's column block of dKp / dVp is final now: its k_proj / v_proj weight gradients belong to the level's
                # module (and to its bucket of the data-parallel exchange), so they go out with this level's flush, in front of
                # the bucket mark — issued after the tree loop they were written into a slice whose all-reduce had already started
                a_ = f"tree_module.tree_modules.{li}.inference.attention.attention_layers.0"
                self._wgrad(plan, f"attn.k_proj{l}", _addr(dKp, li * dk), n_mod * dk, B * T, dk, kv["keys"].data_ptr(), dk,
                            self.g(f"{a_}.k_proj.weight"), ldw=dk, sr=dk, sb=B * T * dk, rpb=B * T, dbias=self.g(f"{a_}.k_proj.bias"))
                self._wgrad(plan, f"attn.v_proj{l}", _addr(dVp, li * nz), n_mod * nz, B * T, nz, o["inf_enc_seq"].data_ptr(), nz,
                            self.g(f"{a_}.v_proj.weight"), ldw=nz, sr=nz, sb=B * T * nz, rpb=B * T, dbias=self.g(f"{a_}.v_proj.bias"))
            if split:
                plan.lane = 2
            self._mlp_bwd(plan, f"prior{l}", f"{p}.prior", rec[f"mlp:prior{l}"], Wt["prior"], dp.data_ptr(), 2 * nv,
                          [(dXp.data_ptr(), n * 2 * nz, 2 * nz)], group=grp)
            self._mlp_bwd_group(plan, f"level{l}", grp)
            if split:
                plan.lane = 0
                if hp.tree_lstm:
                    dh_batched()
                merge_backward()
                plan.join([1, 2])
            ctx = (2 * nz + nv, 3 * nz + nv) if hp.context_every_step else (-1, -1)
            srcs = [(dpi.data_ptr(), pid, 0, nz, ctx[0], ctx[1], 0), (dXq.data_ptr(), 2 * nz, 0, nz, -1, -1, 0),
                    (dXp.data_ptr(), 2 * nz, 0, nz, -1, -1, 0)]
            if dXi is not None:
                srcs.append((dXi.data_ptr(), 2 * nz + nv, 0, nz, -1, -1, 0))
            if dXa is not None:
                srcs.append((dXa.data_ptr(), 2 * nz, 0, nz, -1, -1, 0))
            self._tree_accum(plan, f"E{l}", dE, PS * nz, 2 * s * nz, B, n, nz, srcs)
            if held and l <= self.dec_side_level:
                plan.deferred, held = held + plan.deferred, []
            if merge_lane and hp.tree_lstm and plan.deferred:
                for sl in range(1, 1 + self.n_side):    # the projections' weight gradients read the merge lane's d merged
                    plan.wait(sl, MERGE_LANE)
            self._flush(plan)
            if f"tree{l}" in self._bucket_index:
                # every gradient of this level's module has been issued (main lane + the side lanes just flushed): its bucket of the
                # data-parallel exchange can start while the levels above are differentiated
                plan.mark("bucket", self._bucket_index[f"tree{l}"])

        if merge_lane:
            plan.mark("slices", None)         # (step(): the early optimizer slices held back for the merge chains go out here)
        # ---- temporal inference encoder + image encoders (base_gcp.py:184-213 backward) ----
        d_inf = buf("bw.d_inf", (B * T, nz))
        if attentive:
            # values: d inf_enc_seq = [dV'_0 | dV'_1 | ...] @ [Wv_0; ...]; keys: the same through k_proj, the per-frame key Linear
            # and the second temporal encoder (base_gcp.py:122-123, :200)
            dense = lambda t, w: m._rowsrc(t.data_ptr(), 0, w, w)
            self._dgemm(plan, "attn.v_proj", [dense(dVp, n_mod * nz)], B * T, nz, B * T, self.bk["attn.v_proj.wT"], d_inf.data_ptr(), 0, nz)
            dkeys = buf("bw.dkeys", (B * T, dk))
            self._dgemm(plan, "attn.k_proj", [dense(dKp, n_mod * dk)], B * T, dk, B * T, self.bk["attn.k_proj.wT"], dkeys.data_ptr(), 0, dk)
            self._wgrad(plan, "kseq.key", dkeys.data_ptr(), dk, B * T, dk, kv["kenc"].data_ptr(), nz,
                        self.g("inf_key_encoder.1.linear.weight"), ldw=nz, sr=nz, sb=B * T * nz, rpb=B * T,
                        dbias=self.g("inf_key_encoder.1.linear.bias"))
            dkenc = buf("bw.dkenc", (B * T, nz))
            self._dgemm(plan, "kseq.key", [dense(dkeys, dk)], B * T, nz, B * T, self.bk["kseq.key.wT"], dkenc.data_ptr(), 0, nz)
            d_enc_key = self._seq_backward(plan, fplan, dkenc, B, tag="kseq", prefix="inf_key_encoder.0.net")
        else:
            plan.add("bw.tscatter", lib.gcpx_timestep_scatter, _addr(dET, nz), PS * nz, nz, o["node_t"].data_ptr(), d_inf.data_ptr(),
                     B, N, T, nz)
        d_enc_traj = self._seq_backward(plan, fplan, d_inf, B)
        if attentive:
            plan.add("bw.addrows.kenc", lib.gcpx_add_rows, d_enc_traj.data_ptr(), T * nz, nz, d_enc_key.data_ptr(), None, B, T, nz)
        self._flush(plan)
        self._three_encoder_passes(plan, fplan, lambda: self._encoder_backward(plan, fplan, "traj", d_enc_traj.data_ptr(), nz, 0, 0, {}),
                                   lambda: self._encoder_backward(plan, fplan, "I0", _addr(dE), nz, 1, PS * nz, dskip),
                                   lambda: self._encoder_backward(plan, fplan, "Ig", _addr(dE, 2 ** L * nz), nz, 1, PS * nz, {}))
        if self.side_lanes:
            plan.join(list(range(1, 1 + self.n_side)))
        plan.outs = dict(dE=dE, dHid=dHid, dET=dET, dQZ=dQZ, dPZ=dPZ, dMD=dMD, d_inf=d_inf, d_enc_traj=d_enc_traj, dE_dec=dE_dec,
                         dE_ex=(None if adaptive else dE_ex), dlen=dlen, dexist=dexist, dstate=dstate)
        return plan

    def _three_encoder_passes(self, plan, fplan, traj, i0, ig):
        """The backward chains of the three encoder passes (trajectory frames, I_0, I_g: base_gcp.py:188,208,209) are independent — each a
        chain of ~15 small launches, 0.7 / 0.3 / 0.3 ms at c2 — so the two image passes run on the side lanes beside the trajectory pass
        instead of behind it.  Their weight gradients ACCUMULATE into the same parameters: the trajectory pass's go out first (over the
        lanes, one layer per tag), the image passes' behind them on one lane."""
        if not (self.side_lanes and self.n_side >= 2 and self.parallel_encoder_passes):
            traj()
            self._flush(plan)
            i0()
            ig()
            self._flush(plan, one_lane=True)      # same parameters as the trajectory pass: one lane, behind it
            return
        assert not plan.deferred
        plan.fork([1, 2])
        plan.lane = 1
        i0()
        plan.lane = 2
        ig()
        plan.lane = 0
        late, plan.deferred = plan.deferred, []
        traj()
        self._flush(plan)
        plan.deferred = late
        self._flush(plan, one_lane=True)

    def _tree_accum(self, plan, tag, dst, dst_sb, slot_stride, B, n, width, srcs):
        a = rt.TreeAccumArgs()
        for i, (ptr, ld, ol, orr, c0, cg, dcol) in enumerate(srcs):
            s = a.src[i]
            s.ptr, s.ld, s.off_left, s.off_right, s.off_ctx0, s.off_ctxg, s.dst_col = ptr, ld, ol, orr, c0, cg, dcol
        a.dst, a.dst_sb, a.slot_stride, a.nsrc, a.B, a.n, a.width = dst.data_ptr(), dst_sb, slot_stride, len(srcs), B, n, width
        plan.keep.append(a)
        plan.add(f"bw.accum:{tag}", self.m.lib.gcpx_tree_accum, C.byref(a))

    # ---- decoder ----
    def _decoder_backward(self, plan, fplan, dMD, B, maps=None):
        """maps (models whose decoded frames are not tree nodes — the flat VRNN): dict(R = rows of dMD, row2src [R] int32 = the
        decoded frame whose features row r of the head's weight gradient reads, frame2row [F] = row of frame f (-1: none),
        row2frame [R] = its inverse (-1 for rows no frame maps to))."""
        m, hp, lib = self.m, self.m._hp, self.m.lib
        rec, o = fplan.rec, fplan.outs
        T, N, nz, L = hp.max_seq_len, hp.n_nodes, hp.nz_enc, hp.hierarchy_levels
        S, pitch = hp.img_sz, m._head_pitch
        buf = m._buf
        dec = rec["dec"]
        F, rpb = dec["F"], dec["rpb"]
        ngf = hp.ngf
        perm32 = buf("bw.dlm_perm", (pitch,), torch.int32)
        perm32.copy_(m._dlm_perm.to(torch.int32))
        # output head: weight gradient over the frames that carry a loss gradient, data gradient to every node frame.
        # balanced: the matched frames (row b*T+t of dMD <- node matched to frame t); adaptive: every node frame
        all_frames = hp.adaptive and maps is None
        R = maps["R"] if maps is not None else (F if all_frames else B * T)
        row_map = None
        if maps is not None:
            row_map = maps["row2src"]
        elif not all_frames:
            row_map = buf("bw.f2n_abs", (B, T), torch.int32)
            plan.add("bw.f2n_abs", lib.gcpx_index_offset, o["frame2node"].data_ptr(), row_map.data_ptr(), B, T, N)
        hs = rec["head_src"]                                   # (pointer, channels, frame divisor, scale, shift, activation)
        if (self.fuse_stage and m.split_f16 and self.split_wgrad and ngf == 16 and hs[2] == 1 and hs[5] == rt.ACT_LRELU and
                hs[3] is not None and (S in (8, 16) or S % 32 == 0)):
            # the split-f16 kernel reads the last block's raw output at the rows' frames and applies BatchNorm affine + LeakyReLU on load
            head_bias_fused = bool(rec.get("head_grad_fused")) and pitch == 112
            self._wgrad_conv3(plan, "dec.head", dMD.data_ptr(), pitch, None, R, S, S, ngf, pitch, self.g("decoder.gen_head.conv.weight"),
                              n_map=perm32, src=(hs[0], rt.ptr(row_map), rt.ptr(hs[3]), rt.ptr(hs[4])),
                              dbias=(self.g("decoder.gen_head.conv.bias") if head_bias_fused else None))
        else:
            head_bias_fused = False
            featA = buf("bw.featA", (R, S, S, ngf))
            a = m._conv_args([hs], R, S, S, S, S, ngf, ngf, self._zeros, self._zeros, featA)
            if row_map is not None:
                a.src_row_map = row_map.data_ptr()
            plan.keep.append(a)
            # the materialised conv input is only read by the weight gradient: both go to a side lane
            self._side(plan, "bw.stage:dec.head", lib.gcpx_conv_stage, C.byref(a))
            self._wgrad_conv3(plan, "dec.head", dMD.data_ptr(), pitch, featA.data_ptr(), R, S, S, ngf, pitch,
                              self.g("decoder.gen_head.conv.weight"), n_map=perm32)
        if head_bias_fused:
            pass                       # (column sums of dMD came out of the weight-gradient launch)
        elif rec.get("head_grad_fused"):
            # the head kernel wrote the gradient rows itself: the bias gradient is their column sum over every pixel (a side-lane pass
            # over dMD next to the weight gradient, which reads the same rows)
            self._colsum(plan, "dec.head", dMD.data_ptr(), pitch, R * S * S, pitch, self.g("decoder.gen_head.conv.bias"), n_map=perm32)
        else:
            # bias: per-frame column sums come out of the loss-gradient kernel
            self._colsum(plan, "dec.head", buf("bw.dMD.colsum", (R, pitch)).data_ptr(), pitch, R, pitch,
                         self.g("decoder.gen_head.conv.bias"), n_map=perm32)
        if self.early_fork:
            self._flush(plan)
        dA = buf("bw.dA.head", (F, S, S, ngf))
        a = m._conv_args([(dMD.data_ptr(), pitch, 1, None, None, rt.ACT_NONE)], F, S, S, S, S, ngf, ngf, self.bk["dec.head.wT"],
                         self._zeros, dA)
        if maps is not None:
            a.src_row_map = maps["frame2row"].data_ptr()
            a.src_row_frames, a.n_src_rows = maps["row2frame"].data_ptr(), R
        elif not all_frames:
            a.src_row_map = o["node2row"].data_ptr()
            # inverse map: the kernel walks the B*T matched rows (padded rows, which no node maps to, are -1)
            row2frame = buf("bw.row2frame", (B * T,), torch.int32)
            plan.add("bw.row2frame", lib.gcpx_index_inverse, o["node2row"].data_ptr(), F, row2frame.data_ptr(), B * T)
            a.src_row_frames, a.n_src_rows = row2frame.data_ptr(), B * T
        m._set_split(a, "bw.dec.head")
        # the head's data gradient is the gradient of the last block's BatchNorm + LeakyReLU output: the split-f16 kernel applies the
        # activation's derivative and sums the BatchNorm statistics in its epilogue (gcpx_conv_args.bwd_r) — gcpx_act_bwd's pass over
        # 2 x 533 MB (c2) on the critical lane is gone
        head_fused = None
        last = dec["blocks"][-1]
        if self.fuse_head_act and bool(a.wpk_split) and ngf == 16 and last["cout"] == 16 and pitch // 16 >= 2:
            bn_l = rec[f"bn:dec.bn.{last['name']}"]
            nb_h = lib.gcpx_conv_grid() // 2
            st_h = buf("bw.st:dec.head_fused", (nb_h, 2, 16))
            a.bwd_r = last["out"].data_ptr()
            a.bwd_scale, a.bwd_shift = bn_l["scale"].data_ptr(), bn_l["shift"].data_ptr()
            a.bwd_mean, a.bwd_rstd = bn_l["mean"].data_ptr(), bn_l["rstd"].data_ptr()
            a.stats_partial = st_h.data_ptr()
            head_fused = (dA, st_h, nb_h)
        plan.keep.append(a)
        plan.add("bw.dgrad:dec.head", lib.gcpx_conv3x3, C.byref(a))
        if plan.rec.get("zero_on_lane2"):
            plan.wait(0, 2)                      # (the gradient vector is cleared there: _build_backward)

        gin = (dA.data_ptr(), ngf, 0)            # (pointer, channel pitch, upsampled?) of the incoming gradient
        dskip = {}
        pending_skip = None                      # skip half of the block behind this one, summed in this block's activation pass
        for blk in reversed(dec["blocks"]):
            name, res_in, cout, c_prev, c_skip = blk["name"], blk["res_in"], blk["cout"], blk["c_prev"], blk["c_skip"]
            res = 2 * res_in
            cin = c_prev + c_skip
            bn = rec[f"bn:dec.bn.{name}"]
            dy = self._bn_bwd(plan, f"dec.{name}", bn, gin[0], gin[1], 0, gin[2], blk["out"], F, res, res,
                              fused=(head_fused if blk is last else None), skip=pending_skip)
            pending_skip = None
            # 16-output-channel blocks: the split-f16 weight gradient interpolates its operand from the block's own sources; the others
            # materialise it first (gcpx_conv_stage)
            fused_up = (self.fuse_stage and m.split_f16 and self.split_wgrad and cout == 16 and cin % 32 == 0 and
                        all(sdesc[1] % 16 == 0 for sdesc in blk["srcs"]) and (res in (8, 16) or res % 32 == 0))
            U = None if fused_up else buf(f"bw.U.{name}", (F, res, res, cin))
            a = m._conv_args(blk["srcs"], F, res_in, res_in, res, res, cin, cin, self._zeros, self._zeros, U, upsample=1)
            plan.keep.append(a)
            if fused_up:
                self._wgrad_conv3(plan, f"dec.{name}", dy.data_ptr(), cout, None, F, res, res, cin, cout,
                                  self.g(f"decoder.net.{name}.conv.weight"), up_args=a)
            else:
                self._side(plan, f"bw.stage:dec.{name}", lib.gcpx_conv_stage, C.byref(a))
                self._wgrad_conv3(plan, f"dec.{name}", dy.data_ptr(), cout, U.data_ptr(), F, res, res, cin, cout,
                                  self.g(f"decoder.net.{name}.conv.weight"))
            if self.early_fork:
                self._flush(plan)
            dU = buf(f"bw.dU.{name}", (F, res, res, cin))
            quarters = (self.split_dgrad_wide and m.split_f16 and res % 16 == 0 and res >= 16 and cin % 32 == 0 and cout % 16 == 0 and
                        f"bw.dec.{name}.q0" in getattr(m, "pk_split", {}) and f"dec.{name}.wTq0" in self.bk)
            for h in range(cin // 32 if quarters else (cin + 63) // 64):
                if quarters:
                    # 32 output channels per launch on the split-f16 wave kernel (its f32 pack is not read: any valid pointer)
                    a = m._conv_args([(dy.data_ptr(), cout, 1, None, None, rt.ACT_NONE)], F, res, res, res, res, 32, cin,
                                     self.bk[f"dec.{name}.wTq{h}"], self._zeros, dU)
                    a.out = dU.data_ptr() + 4 * 32 * h
                    m._set_split(a, f"bw.dec.{name}.q{h}")
                    plan.keep.append(a)
                    plan.add(f"bw.dgrad:dec.{name}.q{h}", lib.gcpx_conv3x3, C.byref(a))
                    continue
                ch = min(64, cin - 64 * h)
                a = m._conv_args([(dy.data_ptr(), cout, 1, None, None, rt.ACT_NONE)], F, res, res, res, res, ch, cin,
                                 self.bk[f"dec.{name}.wT{h}"], self._zeros, dU)
                a.out = dU.data_ptr() + 4 * 64 * h
                if h == 0 and cin <= 64:
                    m._set_split(a, f"bw.dec.{name}")
                plan.keep.append(a)
                plan.add(f"bw.dgrad:dec.{name}.{h}", lib.gcpx_conv3x3, C.byref(a))
            if c_skip:
                ds = buf(f"bw.dskip.{name}", (B, res_in, res_in, c_skip))
                # Both halves of a pixel of dU share its 128-byte lines when the block is 16 + 16 channels wide: the activation pass of the
                # block in front (which reads the other half) then sums the skip half on the way (one pass over 1.07 GB at c2 instead of
                # two); wider blocks keep the two launches (their halves are whole lines, and a sequence-major pass has too few threads)
                nxt_i = dec["blocks"].index(blk) - 1
                fuse = (self.fuse_skip and nxt_i >= 0 and dec["blocks"][nxt_i]["cout"] == c_prev and F % rpb == 0 and F // rpb == B and
                        256 % ((c_prev + c_skip) // 4) == 0 and
                        B * res_in * res_in * ((c_prev + c_skip) // 4) >= int(os.environ.get("GCPX_SKIP_FUSION_MIN_ITEMS", "65536")))
                if fuse:
                    pending_skip = (ds, c_prev, c_skip, rpb)
                else:
                    a = rt.ActBwdArgs()
                    a.da, a.dy, a.ldc, a.c_off, a.up, a.fsum, a.act = dU.data_ptr(), ds.data_ptr(), cin, c_prev, 1, rpb, rt.ACT_NONE
                    a.F, a.H, a.W, a.C = B, res_in, res_in, c_skip
                    plan.keep.append(a)
                    plan.add(f"bw.skip:{name}", lib.gcpx_act_bwd, C.byref(a))
                dskip[blk["skip_idx"]] = ds
            gin = (dU.data_ptr(), cin, 1)
            if not self.defer_decoder_side:
                self._flush(plan)
        # input block: ConvTranspose 1x1 -> 4x4 as a GEMM + BatchNorm
        ctop = m._c_top
        bn0 = rec["bn:dec.bn0"]
        dy0 = self._bn_bwd(plan, "dec.input", bn0, gin[0], gin[1], 0, gin[2], dec["d0"], F, 4, 4)
        es = dec["e_src"]                                  # row source of the decoded latents (tree: E slots 1 .. N; flat VRNN: x_1 .. x_{T-1})
        self._wgrad(plan, "dec.input", dy0.data_ptr(), 16 * ctop, F, 16 * ctop, es.ptr, nz,
                    self.g("decoder.net.input.conv.weight"), rpb=rpb, sb=es.sb, sr=es.sr, wmap=rt.WMAP_CONVT, ntap=16, Cout=ctop)
        dE_dec = buf("bw.dE_dec", (F, nz))
        self._dgemm(plan, "dec.input", [self._dense(dy0.data_ptr(), 16 * ctop, 16 * ctop, F)], F, nz, F, self.bk["dec.input.wT"],
                    dE_dec.data_ptr(), 0, nz)
        return dE_dec, dskip

    # ---- ConvSeqEncodingModule (base_gcp.py:199) ----
    def _seq_backward(self, plan, fplan, d_inf, B, tag="seq", prefix="inf_encoder.net"):
        """backward of one ConvSeqEncodingModule (`tag` = "seq": inf_encoder, "kseq": the attention-key encoder)"""
        m, hp, lib = self.m, self.m._hp, self.m.lib
        rec, o = fplan.rec, fplan.outs
        T, nz, nm = hp.max_seq_len, hp.nz_enc, hp.nz_mid
        buf = m._buf
        R = B * T
        y1, y2, enc_traj = buf(f"{tag}.y1", (R, nm)), buf(f"{tag}.y2", (R, nm)), o["enc_traj_seq"]
        bn = rec[f"bn:{tag}.bn"]
        taps = lambda ptr, w: [m._rowsrc(ptr, T * w, w, w, shift=1 - tap) for tap in range(3)]
        pre = prefix
        self._wgrad(plan, f"{tag}.head", d_inf.data_ptr(), nz, R, nz, y2.data_ptr(), 3 * nm, self.g(f"{pre}.head.conv.weight"),
                    mode=rt.WG_CONV1D, Cin=nm, rpb=T, sb=T * nm, sr=nm, scale=bn["scale"], shiftv=bn["shift"], act=rt.ACT_LRELU,
                    wmap=rt.WMAP_CONV, ntap=3)
        self._colsum(plan, f"{tag}.head", d_inf.data_ptr(), nz, R, nz, self.g(f"{pre}.head.conv.bias"))
        da2 = buf(f"bw.{tag}.da2", (R, nm))
        self._dgemm(plan, f"{tag}.head", taps(d_inf.data_ptr(), nz), R, nm, T, self.bk[f"{tag}.head.wT"], da2.data_ptr(), T * nm, nm)
        dy2 = self._bn_bwd(plan, f"{tag}.bn", bn, da2.data_ptr(), nm, 0, 0, y2, R, 1, 1)
        self._wgrad(plan, f"{tag}.pyr", dy2.data_ptr(), nm, R, nm, y1.data_ptr(), 3 * nm, self.g(f"{pre}.pyramid-0.conv.weight"),
                    mode=rt.WG_CONV1D, Cin=nm, rpb=T, sb=T * nm, sr=nm, wmap=rt.WMAP_CONV, ntap=3)
        da1 = buf(f"bw.{tag}.da1", (R, nm))
        self._dgemm(plan, f"{tag}.pyr", taps(dy2.data_ptr(), nm), R, nm, T, self.bk[f"{tag}.pyramid-0.wT"], da1.data_ptr(), T * nm, nm)
        du1 = buf(f"bw.{tag}.du1", (R, nm))
        plan.add(f"bw.{tag}.lrelu", lib.gcpx_lrelu_bwd, y1.data_ptr(), da1.data_ptr(), du1.data_ptr(), R * nm, C.c_float(hp.leaky_slope))
        self._wgrad(plan, f"{tag}.input", du1.data_ptr(), nm, R, nm, enc_traj.data_ptr(), 3 * nz, self.g(f"{pre}.input.conv.weight"),
                    mode=rt.WG_CONV1D, Cin=nz, rpb=T, sb=T * nz, sr=nz, wmap=rt.WMAP_CONV, ntap=3)
        self._colsum(plan, f"{tag}.input", du1.data_ptr(), nm, R, nm, self.g(f"{pre}.input.conv.bias"))
        d_enc = buf(f"bw.d_enc_traj.{tag}", (R, nz))
        self._dgemm(plan, f"{tag}.input", taps(du1.data_ptr(), nm), R, nz, T, self.bk[f"{tag}.input.wT"], d_enc.data_ptr(), T * nz, nz)
        return d_enc

    # ---- attention of the attentive posterior (attentive_inference.py:47-86), one tree level ----
    def _attention_backward(self, plan, fplan, l, Wt, dEt, dKp, dVp, B):
        """dEt [M, nz] = gradient of e_tilde.  Writes this level's column block of dKp / dVp (projected keys / values) and
        returns dXa [M, 2 nz], the gradient w.r.t. the query network's inputs (e_l | e_r)."""
        m, hp, lib = self.m, self.m._hp, self.m.lib
        rec = fplan.rec
        ar = rec["attn"][l]
        kv = rec["attn_kv"]
        M, n, li = ar["M"], ar["n"], ar["li"]
        T, nz, dk = hp.max_seq_len, hp.nz_enc, hp.nz_attn_key
        n_mod = kv["n_mod"]
        buf = m._buf
        a_ = f"tree_module.tree_modules.{li}.inference.attention"
        dense = lambda t, w: m._rowsrc(t.data_ptr(), 0, w, w)
        lin = lambda tag, dy, x, N, K, name: self._wgrad(plan, tag, dy.data_ptr(), N, M, N, x.data_ptr(), K, self.g(f"{name}.weight"),
                                                         ldw=K, sr=K, sb=M * K, rpb=M, dbias=self.g(f"{name}.bias"))
        # e_tilde = out(raw); raw = out_proj(o)
        lin(f"attn.out{l}", dEt, ar["raw"], nz, nz, f"{a_}.out")
        draw = buf(f"bw.attn.draw{l}", (M, nz))
        self._dgemm(plan, f"attn.out{l}", [dense(dEt, nz)], M, nz, M, Wt["attn.out.wT"], draw.data_ptr(), 0, nz)
        lin(f"attn.out_proj{l}", draw, ar["o"], nz, nz, f"{a_}.attention_layers.0.out_proj")
        do = buf(f"bw.attn.do{l}", (M, nz))
        self._dgemm(plan, f"attn.out_proj{l}", [dense(draw, nz)], M, nz, M, Wt["attn.out_proj.wT"], do.data_ptr(), 0, nz)
        # softmax attention
        dS, dqp, dtr = buf(f"bw.attn.dS{l}", (M, T)), buf(f"bw.attn.dq{l}", (M, dk)), buf(f"bw.attn.dtemp{l}", (M,))
        plan.add(f"bw.attn{l}", lib.gcpx_attention_bwd, ar["qp"].data_ptr(), _addr(kv["Kp"], li * B * T * dk), _addr(kv["Vp"], li * B * T * nz),
                 ar["gamma"].data_ptr(), do.data_ptr(), fplan.rec["tin"]["end_ind"].data_ptr(), ar["temp"].data_ptr(), dS.data_ptr(),
                 dqp.data_ptr(), dtr.data_ptr(), _addr(dKp, li * dk), n_mod * dk, _addr(dVp, li * nz), n_mod * nz, M, n, T, dk, nz)
        self._side(plan, f"bw.attn.dtemp:{l}", lib.gcpx_reduce_partials, dtr.data_ptr(), M, 1, 1,
                   self.g(f"{a_}.attention_layers.0.temperature"), 1)
        # q' = q_proj(query MLP(e_l, e_r))
        lin(f"attn.q_proj{l}", dqp, ar["qin"], dk, dk, f"{a_}.attention_layers.0.q_proj")
        dqin = buf(f"bw.attn.dqin{l}", (M, dk))
        self._dgemm(plan, f"attn.q_proj{l}", [dense(dqp, dk)], M, dk, M, Wt["attn.q_proj.wT"], dqin.data_ptr(), 0, dk)
        dXa = buf(f"bw.dXa{l}", (M, 2 * nz))
        self._mlp_bwd(plan, f"attn.query{l}", f"{a_}.query_net", rec[f"mlp:attn.query{l}"], Wt["attn.query"], dqin.data_ptr(), dk,
                      [(dXa.data_ptr(), n * 2 * nz, 2 * nz)])
        return dXa

    # ---- conv encoder (one of the three passes) ----
    def _encoder_backward(self, plan, fplan, tag, dlat, ldy, dy_rpb, dy_sb, dskip):
        m, hp, lib = self.m, self.m._hp, self.m.lib
        rec = fplan.rec
        er = rec[f"enc:{tag}"]
        F, S, nz = er["F"], hp.img_sz, hp.nz_enc
        buf = m._buf
        layers, ctop = m._enc_layers, m._c_top
        nlay = len(layers)
        top = nlay - 1
        r_top, bn_top = er["r"][top], rec[f"bn:{tag}.bn{top}"]
        K = 16 * ctop
        self._wgrad(plan, f"enc.head:{tag}", dlat, ldy, F, nz, r_top.data_ptr(), K, self.g("encoder.net.head.weight"), rpb=F, sb=0, sr=K,
                    scale=bn_top["scale"], shiftv=bn_top["shift"], act=rt.ACT_LRELU, cmod=ctop, dy_rpb=dy_rpb, dy_sb=dy_sb,
                    wmap=rt.WMAP_CONV, Cin=ctop, ntap=16)
        self._colsum(plan, f"enc.head:{tag}", dlat, ldy, F, nz, self.g("encoder.net.head.bias"), dy_rpb=dy_rpb, dy_sb=dy_sb)
        dA = buf(f"bw.{tag}.dA{top}", (F, 4, 4, ctop))
        if dy_rpb:
            src = m._rowsrc(dlat, dy_sb, 0, nz)
            self._dgemm(plan, f"enc.head:{tag}", [src], F, K, 1, self.bk["enc.head.wT"], dA.data_ptr(), K, 0)
        else:
            self._dgemm(plan, f"enc.head:{tag}", [self._dense(dlat, ldy, nz, F)], F, K, F, self.bk["enc.head.wT"], dA.data_ptr(), 0, K)
        res = 4
        for li in reversed(range(1, nlay)):
            name, cin, cout, _ = layers[li]
            r = er["r"][li]
            bn = rec[f"bn:{tag}.bn{li}"]
            dy = self._bn_bwd(plan, f"{tag}.{name}", bn, dA.data_ptr(), cout, 0, 0, r, F, res, res, add=dskip.get(li),
                              defer_affine=(self.side_lanes and self.n_side >= 2 and self.parallel_encoder_passes))
            if li == 1:
                x, sc, sh, act = er["a0"], None, None, rt.ACT_NONE
            else:
                pbn = rec[f"bn:{tag}.bn{li - 1}"]
                x, sc, sh, act = er["r"][li - 1], pbn["scale"], pbn["shift"], rt.ACT_LRELU
            self._wgrad(plan, f"enc.{name}:{tag}", dy.data_ptr(), cout, F * res * res, cout, x.data_ptr(), 16 * cin,
                        self.g(f"encoder.net.{name}.conv.weight"), mode=rt.WG_CONV4X4S2, Cin=cin, H=2 * res, W=2 * res, scale=sc,
                        shiftv=sh, act=act, wmap=rt.WMAP_CONV, ntap=16)
            dcol = buf(f"bw.{tag}.dcol{li}", (F * res * res, 16 * cin))
            R = F * res * res
            self._dgemm(plan, f"enc.{name}:{tag}", [self._dense(dy.data_ptr(), cout, cout, R)], R, 16 * cin, R, self.bk[f"enc.{name}.wT"],
                        dcol.data_ptr(), 0, 16 * cin)
            dA = buf(f"bw.{tag}.dA{li - 1}", (F, 2 * res, 2 * res, cin))
            plan.add(f"bw.col2im:{tag}.{li}", lib.gcpx_col2im4x4s2, dcol.data_ptr(), dA.data_ptr(), F, 2 * res, 2 * res, cin)
            res *= 2
        # first layer: conv on the NCHW image + LeakyReLU (no norm)
        ngf = hp.ngf
        du0 = buf(f"bw.{tag}.du0", (F, res, res, ngf))
        a = rt.ActBwdArgs()
        a.da, a.r, a.dy = dA.data_ptr(), er["a0"].data_ptr(), du0.data_ptr()
        a.add = dskip[0].data_ptr() if 0 in dskip else None
        a.ldc, a.c_off, a.up, a.fsum, a.act, a.F, a.H, a.W, a.C = ngf, 0, 0, 1, rt.ACT_LRELU, F, res, res, ngf
        plan.keep.append(a)
        # the first layer has no data gradient to pass on (its input is the image): its activation backward and the im2col of the
        # image only feed the weight / bias gradient, so they leave the critical lane together with them (same tag = same side lane,
        # in order)
        if ngf == 16 and S in (32, 64, 128) and self.fused_image_wgrad:
            # one launch for the layer's whole backward (csrc/wgrad_image.hip): LeakyReLU slope, image patches and both sums
            grid = max(1, min(F * (res // 8), 3 * (lib.gcpx_conv_grid() // 2)))
            part = buf(f"bw.{tag}.wimg", (grid, 16 * 48 + 16))
            self._side(plan, f"bw.wgrad:enc.input:{tag}", lib.gcpx_wgrad_image4x4s2, dA.data_ptr(), a.add, er["a0"].data_ptr(), er["x_ptr"],
                       F, S, part.data_ptr(), grid)
            self._side(plan, f"bw.wreduce:enc.input:{tag}", lib.gcpx_reduce_partials, part.data_ptr(), grid, 16 * 48 + 16, 16 * 48,
                       self.g("encoder.net.input.conv.weight"), 1)
            self._side(plan, f"bw.creduce:enc.input:{tag}", lib.gcpx_reduce_partials, part.data_ptr() + 4 * 16 * 48, grid, 16 * 48 + 16, 16,
                       self.g("encoder.net.input.conv.bias"), 1)
            return
        self._side(plan, f"bw.act:enc.input:{tag}", lib.gcpx_act_bwd, C.byref(a))
        col = buf(f"bw.{tag}.col", (F * res * res, 48))
        self._side(plan, f"bw.im2col:enc.input:{tag}", lib.gcpx_im2col_image, er["x_ptr"], col.data_ptr(), F, S, S)
        R = F * res * res
        self._wgrad(plan, f"enc.input:{tag}", du0.data_ptr(), ngf, R, ngf, col.data_ptr(), 48, self.g("encoder.net.input.conv.weight"),
                    ldw=48, sr=48, sb=R * 48, rpb=R)
        self._colsum(plan, f"enc.input:{tag}", du0.data_ptr(), ngf, R, ngf, self.g("encoder.net.input.conv.bias"))

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
            bplan.run(self._backward_streams() + [caller.cuda_stream], on_mark=self._on_mark)
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
            if getattr(self, "_clip_state_dirty", True):
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
            if getattr(self, "_clip_part", None) is None:
                self._clip_part = torch.empty(1024, device=m.device)
            rt.check(m.lib.gcpx_grad_clip_coef(self.grad.data_ptr(), n, scale, float(self.gradient_clip), self._clip_part.data_ptr(), 1024,
                                               self.opt_state.data_ptr(), st), "grad_clip")
        elif getattr(self, "_clip_state_dirty", True):
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
        if self._early_on and getattr(self, "_clip_state_dirty", True):
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


class _PtrHolder:
    """wraps a raw device address so helper signatures that expect tensors (`.data_ptr()`) can take it"""

    def __init__(self, p):
        self._p = p

    def data_ptr(self):
        return self._p
