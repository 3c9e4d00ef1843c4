"""Parameters of the model: ONE flat fp32 vector `theta` in canonical torch layouts, and the packed forms the kernels read (MFMA
fragment order, gate-interleaved LSTM weights, split-f16 twins), re-derived from it by gather kernels after every optimizer step
(WeightsMixin, mixed into model.GCPTreeModel)."""
import ctypes as C
import os
from contextlib import contextmanager

import torch

from . import packing as pk
from . import runtime as rt
from .hparams import GCPHParams
from .params import init_params, encoder_layers, encoder_skip_layers, decoder_layers
from .plan_ops import _addr


class WeightsMixin:

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
        if self._arena is not None:
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
            if self.head32 and hp.n_mixtures == 10 and hp.head_channels == 100:
                # the same weights in 32x32x16 fragment order for csrc/conv3x3_head32.hip (mean-only and fused-likelihood modes)
                todo.append(("dec.head32", "decoder.gen_head.conv.weight", ("head32", pk.dlm_channel_perm(hp.n_mixtures))))
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
            elif isinstance(perm, tuple) and perm[0] == "head32":
                idx = pk.head32_index(shp, off, perm[1]).to(self.device)
            else:
                idx = pk.conv3x3_split_index(shp, off, perm).to(self.device)
            d.update(idx=idx, out=torch.zeros(2 * idx.numel(), dtype=torch.int16, device=self.device),
                     log2=torch.zeros(1, dtype=torch.int32, device=self.device))
            self.pk_split[name] = d
            if perm == "rowfold" and shp[1] == 32:
                # the same folded weights as two 16-channel packs (two horizontal taps per k-step, GCPX_SPLIT_ROWFOLD16): the node half
                # and the skip half of a block whose skip channels are convolved once per sequence (forward_plan: gcpx_conv_args.addend).
                # They gather from the block's fold scratch (folded once, above)
                for tag, cbase in (("f16a", 0), ("f16b", 16)):
                    i16 = pk.conv3x3_fold16_index(shp[1], cbase).to(self.device)
                    self.pk_split[f"{name}.{tag}"] = dict(fold=d["fold"], fold16=True, idx=i16,
                                                          out=torch.zeros(2 * i16.numel(), dtype=torch.int16, device=self.device),
                                                          log2=torch.zeros(1, dtype=torch.int32, device=self.device))
        self.repack_split()

    def repack_split(self, stream=None):
        """re-split every split-f16 weight tensor from the flat parameter vector: the row-folded blocks' weights are folded first, then ONE
        grouped launch splits all tensors side by side (one workgroup each; as separate launches they were 0.45 ms of a training step)"""
        st = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        if not self.pk_split:
            return
        tab = self._split_tab
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
            if "fold_src" in d:
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
        d = self.pk_split.get(name)
        layout = None
        if name == "dec.head" and self.head32 and a.head_mode in (rt.HEAD_DLM_MEAN, rt.HEAD_DLM_NLL, rt.HEAD_DLM_NLL_GRAD) \
                and "dec.head32" in self.pk_split:
            # the modes that store no raw parameters run the 32x32-tile head (csrc/conv3x3_head32.hip)
            d, layout = self.pk_split["dec.head32"], rt.SPLIT_HEAD32
        if self.split_f16 and d is not None:
            a.wpk_split, a.w_split_log2_dev = d["out"].data_ptr(), d["log2"].data_ptr()
            if layout is None:
                layout = rt.SPLIT_ROWFOLD16 if d.get("fold16") else (rt.SPLIT_ROWFOLD if "fold" in d else rt.SPLIT_PLAIN)
            a.split_layout = layout

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
        if hp.seq_enc == "none":                                   # Identity (base_gcp.py:131-132): nothing to pack
            seq_encs = []
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
        self._extra_packs = X0                       # (the caller's packs live in the same arena: a model's re-pack may read them)
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
        for name, (tab, n, scratch) in self._gsplit_tabs.items():
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
