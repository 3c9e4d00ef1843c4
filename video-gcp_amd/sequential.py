"""gcp_sequential: the flat VRNN goal-conditioned predictor, host side.

Mirrors /root/reference/gcp/prediction/models/sequential.py:13-131 (SequentialRecModule / SequentialModel) on top of the
same encoder / decoder / head kernels as the tree model.  The VRNN cell (blox.torch.models.vrnn.VRNNCell, absent) follows
this build's spec (oracle/gcp_sequential_oracle.py; training: training_sequential.py): three recurrent nets (embed Linear -> n LSTMCells -> out
Linear) with zero initial state, run T-1 steps from x_0 = e_0.  Each step is 16 small launches (batch rows = B); the prior
chain runs on a side lane next to the posterior chain.  Weight streaming dominates (3 nets x 3 layers x 4H x 2H floats per
step), so this path is HBM/L2 bound, not MFMA bound.
"""
import ctypes as C

import torch

from . import runtime as rt
from .model import GCPTreeModel, Outputs, _Plan, _addr
from .params import init_params_sequential


class GCPSequentialModel(GCPTreeModel):
    # the flat baseline always rolls out to the fed end_ind (no length predictor)
    _has_aux_training = True      # sampled inverse-model / cost-model training pairs (base_gcp.py:249-260), as for the tree model
    _has_pred_length = False
    _rng_in_plan = False              # (the flat model's plan keeps the torch draw in front of it)
    _has_training = True              # training_sequential.SequentialTrainStep

    def _check_hp(self, hp):
        assert hp.lstm_init in ("zero", "mlp")          # the cell state always starts at zero (hyperparameters.py:96)
        assert hp.n_actions <= 16 or not hp.action_conditioned_pred, "the action rows are padded to one 16-column k-group"

    @property
    def _nets(self):
        """the recurrent nets of the cell: a deterministic predictor (var_inf = 'deterministic', vmpc.py:14-15) has no latent, hence
        neither prior nor inference net"""
        return ("gen_lstm",) if self._hp.deterministic else ("prior_lstm", "inf_lstm", "gen_lstm")

    def _default_params(self, hp, seed):
        return init_params_sequential(hp, seed)

    def _n_latents(self):
        return self._hp.max_seq_len - 1

    def _pack_latent_model(self, P):
        hp = self._hp
        p = "dense_rec.lstm.cell"
        for net in self._nets:
            P[net] = self._pack_hsp(f"{p}.{net}", hp.n_lstm_layers)
        if hp.action_conditioned_pred:                  # sequential.py:108-110
            P["action_encoder"] = self._pack_predictor("action_encoder", hp.nz_enc)

    def _pack_fused_embed(self):
        """embed Linear folded into LSTM layer 0's input projection (float64 product, once per weight load): one launch less on
        every step of the three recurrent chains — see GCPTreeModel._pack_fused_embed"""
        from . import packing as pk
        sd = self.sd
        for net in self._nets:
            p = f"dense_rec.lstm.cell.{net}"
            We, be = sd[f"{p}.embed.weight"].double(), sd[f"{p}.embed.bias"].double()
            Wih = sd[f"{p}.lstm.0.weight_ih"].double()
            w, b = pk.lstm_gate_interleave((Wih @ We).float(), sd[f"{p}.lstm.0.weight_hh"],
                                           (Wih @ be).float() + sd[f"{p}.lstm.0.bias_ih"], sd[f"{p}.lstm.0.bias_hh"])
            self.pk[net]["lstm0f.w"], self.pk[net]["lstm0f.b"] = pk.pack_gemm(w), b
            if net == "gen_lstm":
                # the generator's own output projection folded in as well: x_{t+1} = W_out h_top(t) + b_out only enters step t + 1
                # through layer 0, so layer 0 of step t + 1 reads h_top(t) with (W_ih W_e)[:, x] W_out — the out Linear leaves the
                # dependent chain (3 launches per step instead of 4; x_{t+1} itself is still computed, beside the chain)
                nz = self._hp.nz_enc
                Wf, Wo, bo = Wih @ We, sd[f"{p}.out.weight"].double(), sd[f"{p}.out.bias"].double()
                W2 = torch.cat([Wf[:, :nz] @ Wo, Wf[:, nz:]], 1).float()
                b2 = (Wih @ be + Wf[:, :nz] @ bo).float() + sd[f"{p}.lstm.0.bias_ih"]
                w, b = pk.lstm_gate_interleave(W2, sd[f"{p}.lstm.0.weight_hh"], b2, sd[f"{p}.lstm.0.bias_hh"])
                self.pk[net]["lstm0ff.w"], self.pk[net]["lstm0ff.b"] = pk.pack_gemm(w), b

    # ------------------------------------------------------------------------------------------------
    # A trainer's model keeps the folded packs too (GCPX_SEQ_LIVE_FOLDS=0: not): they are products of parameters, which the arena's gather
    # cannot express, so they are re-formed on the device behind every re-pack — exact-f32 row GEMMs of this library (4H x H x in_dim)
    # against the trainer's live transposed packs + one gather each into the packed layout.  The training forward then runs the same 3-launch
    # generator steps as the plain forward instead of 5 (embed and out Linear off the dependent chain): 7.4 -> ~6 ms at c2.
    def repack(self, stream=None, bucket=None, max_blocks=0):
        super().repack(stream, bucket, max_blocks)
        live = self.__dict__.setdefault("_live_folds", __import__("os").environ.get("GCPX_SEQ_LIVE_FOLDS", "1") == "1")   # (read once per model)
        xp = getattr(self, "_extra_packs", None) or {}
        live = live and all(n in xp and "embed.wT" in xp[n] and "out.wT" in xp[n] for n in self._nets)   # (the trainer's transposed packs)
        if self._arena is not None and live and self._hp.tree_lstm and (bucket is None or bucket == self._arena_ranges[-1][0]):
            self._refresh_folds(stream)

    def _fold_index(self, net, key, rows, cols, H):
        """gather index of pack_gemm(lstm_gate_interleave([W | W_hh])) over the flat buffer [W.flatten(), W_hh.flatten()], and of the bias"""
        from . import packing as pk
        cache = self.__dict__.setdefault("_fold_idx", {})
        if (net, key) not in cache:
            dev = self.device
            iw = torch.arange(rows * cols, device=dev).view(rows, cols)
            ih = rows * cols + torch.arange(rows * H, device=dev).view(rows, H)
            z = torch.zeros(rows, dtype=torch.int64, device=dev)
            w_idx, _ = pk.lstm_gate_interleave(iw, ih, z, z)
            _, b_idx = pk.lstm_gate_interleave(iw, ih, torch.arange(rows, device=dev), z)
            cache[(net, key)] = (pk.pack_gemm(w_idx).contiguous(), b_idx.contiguous())
        return cache[(net, key)]

    def _refresh_folds(self, stream=None):
        """W_ih0 @ W_e (and, for the generator, its x columns @ W_out) on the exact-f32 row GEMM of this library against the transposed
        packs the trainer keeps live for its data gradients (`embed.wT`, `out.wT`): fixed summation order.  (A library GEMM was tried
        first: torch.matmul in float64 gave different last bits from call to call — rocBLAS splitting k with atomics, torch's
        deterministic switch notwithstanding — and two trainers fed the same data drifted apart by the third step.)  Biases: products
        and row sums in torch (no atomics); then one gather each into the packed, gate-interleaved layout."""
        sd, hp, lib = self.sd, self._hp, self.lib
        H, nz = hp.nz_mid_lstm, hp.nz_enc
        bw = self._extra_packs
        st = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        cur = torch.cuda.current_stream(self.device)
        # (torch ops of this function go to `st`: normally torch's current stream itself — wrapping the raw pointer of the DEFAULT stream
        # in an ExternalStream gave a stream torch did not order with its own default-stream launches: folds read half-updated parameters)
        ctx = torch.cuda.device(self.device) if int(st) == int(cur.cuda_stream) else \
            torch.cuda.stream(torch.cuda.ExternalStream(int(st), device=self.device))
        scratch = self.__dict__.setdefault("_fold_scratch", {})

        def gemm(rows_ptr, row_stride, M, K, wpk, N, out, out_stride):
            a = rt.GemmArgs()
            a.src[0] = self._rowsrc(rows_ptr, 0, row_stride, K)
            a.nsrc, a.M, a.N, a.K, a.rpb = 1, M, N, K, M
            a.wpk, a.bias, a.out, a.ob, a.orow, a.epi = wpk.data_ptr(), None, out, 0, out_stride, rt.EPI_NONE
            rt.check(lib.gcpx_gemm(C.byref(a), st), "fold gemm")
        with ctx, torch.no_grad():
            for net in self._nets:
                p, W = f"dense_rec.lstm.cell.{net}", self.pk[net]
                Wih, Whh = sd[f"{p}.lstm.0.weight_ih"], sd[f"{p}.lstm.0.weight_hh"]
                be, bsum = sd[f"{p}.embed.bias"], sd[f"{p}.lstm.0.bias_ih"] + sd[f"{p}.lstm.0.bias_hh"]
                in_dim = sd[f"{p}.embed.weight"].shape[1]
                key = (net, in_dim)
                if key not in scratch:
                    scratch[key] = (torch.empty(4 * H * in_dim + 4 * H * H, device=self.device),
                                    torch.empty(4 * H * (H + in_dim - nz) + 4 * H * H, device=self.device))
                src1, src2 = scratch[key]
                Wf = src1[:4 * H * in_dim].view(4 * H, in_dim)
                gemm(Wih.data_ptr(), H, 4 * H, H, bw[net]["embed.wT"], in_dim, Wf.data_ptr(), in_dim)
                bf = (Wih * be[None, :]).sum(1)
                forms = [("lstm0f", src1, 4 * H * in_dim, in_dim, bf)]
                if net == "gen_lstm":
                    w2c = H + in_dim - nz
                    W2 = src2[:4 * H * w2c].view(4 * H, w2c)
                    gemm(Wf.data_ptr(), in_dim, 4 * H, nz, bw[net]["out.wT"], H, W2.data_ptr(), w2c)
                    W2[:, H:].copy_(Wf[:, nz:])
                    Wo_b = sd[f"{p}.out.bias"]
                    forms.append(("lstm0ff", src2, 4 * H * w2c, w2c, bf + (Wf[:, :nz] * Wo_b[None, :]).sum(1)))
                for fkey, src, nW, cols, bvec in forms:
                    w_idx, b_idx = self._fold_index(net, fkey, 4 * H, cols, H)
                    if fkey + ".w" not in W:
                        W[fkey + ".w"] = torch.empty(w_idx.shape, dtype=torch.float32, device=self.device)
                        W[fkey + ".b"] = torch.empty(b_idx.shape, dtype=torch.float32, device=self.device)
                    src[nW:].copy_(Whh.reshape(-1))
                    torch.index_select(src, 0, w_idx.view(-1), out=W[fkey + ".w"].view(-1))
                    torch.index_select(bvec + bsum, 0, b_idx, out=W[fkey + ".b"])

    def _hsp_stages(self, plan, name, W, srcs, B, state, par, out_ptr, out_ob, N_out, w0="lstm0f", xs_buf=None):
        """One step of a recurrent predictor as its dependent stages: [LSTM layer 0 (with the folded embedding), layer 1, ...,
        out Linear].  Each stage is a function(group) that appends its GEMM to `group` — the same stage of nets that do not
        depend on each other then shares ONE launch (gcpx_gemm_group).  state[par][i] / state[1 - par][i]: the [h | c] rows the
        step reads / writes (two buffers in ping-pong for inference; the training forward hands in two consecutive slices of a
        buffer that keeps every step's state, and `xs_buf`, the step's slice of the stacked layer inputs)."""
        hp = self._hp
        H, nl = hp.nz_mid_lstm, hp.n_lstm_layers
        fused = "lstm0f.w" in W and (not self.save_for_backward or self._arena is not None)
        if xs_buf is None:
            xs_buf = [self._buf(f"{name}.x{i}", (B, H)) for i in range(nl + 1)]
        stages = []
        if not fused:
            stages.append(lambda g: self._gemm(plan, f"{name}.embed", srcs, B, H, 1, W["embed.w"], W["embed.b"],
                                               out=xs_buf[0].data_ptr(), ob=H, orow=0, group=g))
        for i in range(nl):
            def layer(g, i=i):
                s_in, s_out = state[par][i], state[1 - par][i]             # [B, 2H] = [h | c]
                xs = self._rowsrc(xs_buf[i].data_ptr(), H, 0, H)
                hs = self._rowsrc(s_in.data_ptr(), 2 * H, 0, H)
                lstm = (_addr(s_in, H), 2 * H, s_out.data_ptr(), _addr(s_out, H), 2 * H, 0, xs_buf[i + 1].data_ptr())
                if i == 0 and fused:
                    self._gemm(plan, f"{name}.lstm0", list(srcs) + [hs], B, 4 * H, 1, W[w0 + ".w"], W[w0 + ".b"], epi=rt.EPI_LSTM,
                               lstm=lstm, group=g)
                else:
                    self._gemm(plan, f"{name}.lstm{i}", [xs, hs], B, 4 * H, 1, W[f"lstm{i}.w"], W[f"lstm{i}.b"], epi=rt.EPI_LSTM,
                               lstm=lstm, group=g)
            stages.append(layer)
        stages.append(lambda g: self._gemm(plan, f"{name}.out", [self._rowsrc(xs_buf[nl].data_ptr(), H, 0, H)], B, N_out, 1,
                                           W["out.w"], W["out.b"], out=out_ptr, ob=out_ob, orow=0, group=g))
        return stages

    def _plan_hsp(self, plan, name, W, srcs, B, state, par, out_ptr, out_ob, N_out):
        """One step of a recurrent predictor on its own: embed, n LSTM layers (state ping-pong `par` -> 1-par), out."""
        for st in self._hsp_stages(plan, name, W, srcs, B, state, par, out_ptr, out_ob, N_out):
            st(None)

    def _build_plan(self, key, tin):
        hp, P, lib = self._hp, self.pk, self.lib
        B, has_traj, has_z, sample_prior, phase, with_loss = key[0], key[1], key[2], key[3], key[4], key[7]
        T, nz, nv, H, nl = hp.max_seq_len, hp.nz_enc, hp.nz_vae, hp.nz_mid_lstm, hp.n_lstm_layers
        S = hp.img_sz
        plan = _Plan(lib)
        X = self._buf("seq.X", (B, T, nz))                   # X[:, 0] = e_0, X[:, 1:] = encodings
        EG = self._buf("seq.EG", (B, nz))
        PZ = self._buf("seq.PZ", (B, T - 1, 2 * nv))
        QZ = self._buf("seq.QZ", (B, T - 1, 2 * nv), zero=True)
        Z = self._buf("seq.Z", (B, T - 1, nv))
        idx = self._buf("seq.idx", (B, T), torch.int32)
        seq_len = self._buf("seq.len", (B,), torch.int32)
        plan.add("seq_index", lib.gcpx_seq_index, tin["end_ind"].data_ptr(), B, T, idx.data_ptr(), seq_len.data_ptr())
        train_aux = has_traj and phase == "train" and not sample_prior    # the posterior path of a training / validation-loss forward
        if "aux_n" in tin:
            AUXK = ("inv_t0", "inv_t1", "cost_start_idx", "cost_end_idx")
            plan.add("aux_sample_indices", lib.gcpx_aux_sample_indices_gauss, tin["end_ind"].data_ptr(), tin["aux_n"].data_ptr(), B,
                     hp.inv_mdl_temp_dist, *[tin[k].data_ptr() for k in AUXK])

        # ---- run_encoder (base_gcp.py:184-213) ----
        enc_traj = None
        plan.fork([1, 2])
        plan.lane = 1
        skips = self._plan_encoder(plan, "I0", tin["I_0"].data_ptr(), B, _addr(X), T * nz, 0, 1)
        plan.lane = 2
        if hp.attach_cost_mdl and hp.run_cost_mdl and train_aux and "cost_start_idx" in tin:
            # ground-truth cost of the cost model's sampled segment (cost_mdl.py:101-117, EuclideanPathLength), beside the I_g encoder
            rows = hp.input_nc * hp.img_sz
            plan.add("path_cost", lib.gcpx_path_cost, tin["traj_seq"].data_ptr(), tin["cost_start_idx"].data_ptr(),
                     tin["cost_end_idx"].data_ptr(), B, T, rows, hp.img_sz, self._buf("cost_partial", (B, rows)).data_ptr(),
                     self._buf("cost_target", (B,)).data_ptr())
        self._plan_encoder(plan, "Ig", tin["I_g"].data_ptr(), B, _addr(EG), nz, 0, 1)
        plan.lane = 0
        if has_traj:
            enc_traj = self._buf("enc_traj", (B * T, nz))
            self._plan_encoder(plan, "traj", tin["traj_seq"].data_ptr(), B * T, enc_traj.data_ptr(), T * nz, nz, T)
        plan.join([1, 2])
        outs = {}
        e0 = lambda: self._rowsrc(_addr(X), T * nz, 0, nz)
        eg = lambda: self._rowsrc(_addr(EG), nz, 0, nz)
        if hp.regress_length:
            logits = self._buf("seq_len_logits", (B, T))
            self._mlp(plan, "length_pred", P["length_pred"], [e0(), eg()], B, 1, out=logits.data_ptr(), ob=T, orow=0)
            outs["seq_len_logits"] = logits

        # ---- encoded actions (base_gcp.py:211-213): `more_context` of every step of the cell (sequential.py:45-49) ----
        EA = None
        if hp.action_conditioned_pred:
            na = hp.n_actions
            assert tuple(tin["actions"].shape) == (B, T - 1, na), "actions [B, T-1, n_actions]: action t leads to frame t + 1 (sequential.py:50)"
            A16 = self._buf("seq.actions16", (B, T - 1, 16), zero=True)       # rows padded to one k-group of the Predictor's input layer
            plan.add("actions.pad", lib.gcpx_rows_strided, A16.data_ptr(), (T - 1) * 16, 16, tin["actions"].data_ptr(), (T - 1) * na, na,
                     B, T - 1, na, 0)
            EA = self._buf("seq.EA", (B, T - 1, nz))
            self._mlp(plan, "action_encoder", P["action_encoder"], [self._rowsrc(A16.data_ptr(), (T - 1) * 16, 16, 16)], B * (T - 1), T - 1,
                      out=EA.data_ptr(), ob=(T - 1) * nz, orow=nz)

        # ---- VRNN rollout (sequential.py:49-54) ----
        NETS = self._nets
        keep = self.save_for_backward
        if keep:
            # training forward: the backward pass through the recurrence needs every step's states and layer inputs.  S[net][t, i] =
            # [h | c] of layer i BEFORE step t (S[net][0] = 0), XS[net][t, i] = input of layer i at step t (i = 0: the embedding,
            # i = nl: the top hidden state) — stacked, so the weight gradients of all T - 1 steps are ONE GEMM per weight
            Sall = {net: self._buf(f"{net}.S", (T, nl, B, 2 * H)) for net in NETS}
            XSall = {net: self._buf(f"{net}.XS", (T - 1, nl + 1, B, H)) for net in NETS}
            plan.rec["seq"] = dict(S=Sall, XS=XSall, X=X, EG=EG, PZ=PZ, QZ=QZ, Z=Z, EA=EA)
            for net in NETS:
                plan.add(f"zero.{net}", lib.gcpx_fill_zero, Sall[net].data_ptr(), nl * B * 2 * H * 4)

            class _StepState:           # state[t & 1] -> S[t], state[1 - (t & 1)] -> S[t + 1]: what _hsp_stages indexes with `par`
                def __init__(self, S, t):
                    self.S, self.t = S, t

                def __getitem__(self, par):
                    tt = self.t if par == (self.t & 1) else self.t + 1
                    return [self.S[tt, i] for i in range(nl)]
            state_of = lambda net, t: _StepState(Sall[net], t)
            xs_of = lambda net, t: [XSall[net][t, i] for i in range(nl + 1)]
        else:
            state = {net: [[self._buf(f"{net}.s{par}.{i}", (B, 2 * H)) for i in range(nl)] for par in range(2)] for net in NETS}
            for net in state:
                for i in range(nl):
                    plan.add(f"zero.{net}.{i}", lib.gcpx_fill_zero, state[net][0][i].data_ptr(), B * 2 * H * 4)
            state_of = lambda net, t: state[net]
            xs_of = lambda net, t: None
        def ctx(t):
            c = [e0(), eg()] if hp.context_every_step else []
            return c + ([self._rowsrc(_addr(EA, t * nz), (T - 1) * nz, 0, nz)] if EA is not None else [])
        posterior = has_traj and not sample_prior and not has_z and not hp.deterministic
        # Three recurrent nets, ONE lane.  The inference net reads the ENCODED ground truth only (sequential.py:51-54): it does not
        # wait for the generator, so its step t + 1 is computed NEXT TO the generator's step t — stage by stage in the same
        # launches (gcpx_gemm_group), together with the prior net's step t (needs x_t, feeds only the KL term).  A step of the
        # posterior rollout is then 4 grouped launches + the sample instead of 13 launches on three lanes: the first version's
        # 1300-node multi-lane graph cost the host 9.5 ms per replay, more than the device needed.
        xt_of = lambda t: self._rowsrc(_addr(X, t * nz), T * nz, 0, nz)

        def prior_stages(t):
            return self._hsp_stages(plan, f"prior{t}", P["prior_lstm"], [xt_of(t)] + ctx(t), B, state_of("prior_lstm", t), t & 1,
                                    _addr(PZ, t * 2 * nv), (T - 1) * 2 * nv, 2 * nv, xs_buf=xs_of("prior_lstm", t))

        def inf_stages(t):
            xp = self._rowsrc(_addr(enc_traj, (t + 1) * nz), T * nz, 0, nz)
            return self._hsp_stages(plan, f"inf{t}", P["inf_lstm"], [xp] + ctx(t), B, state_of("inf_lstm", t), t & 1,
                                    _addr(QZ, t * 2 * nv), (T - 1) * 2 * nv, 2 * nv, xs_buf=xs_of("inf_lstm", t))

        def z_source(t):
            zt = (_addr(Z, t * nv), (T - 1) * nv, 0)
            if has_z:
                return self._rowsrc(_addr(tin["z"], t * nv), (T - 1) * nv, 0, nv)
            muls = QZ if posterior else PZ
            plan.add(f"sample{t}", lib.gcpx_gauss_sample, _addr(muls, t * 2 * nv), (T - 1) * 2 * nv, 0,
                     _addr(tin["eps"], t * nv), tin["eps"].shape[1] * nv, 0, zt[0], zt[1], zt[2], B, 1, nv)
            return self._rowsrc(zt[0], zt[1], zt[2], nv)

        def gen_stages(t, zsrc):
            return self._hsp_stages(plan, f"gen{t}", P["gen_lstm"], [xt_of(t)] + ([zsrc] if zsrc is not None else []) + ctx(t), B,
                                    state_of("gen_lstm", t), t & 1, _addr(X, (t + 1) * nz), T * nz, nz, xs_buf=xs_of("gen_lstm", t))

        def run(*chains):
            """stage i of every chain in one launch"""
            chains = [c for c in chains if c]
            for i in range(max(len(c) for c in chains)):
                g = []
                for c in chains:
                    if i < len(c):
                        c[i](g)
                self._gemm_group(plan, f"step.{g[0][0]}", g)

        def sample_stage(t):
            """z_t = mu + exp(log_sigma) eps of q(z_t) as a one-stage chain: the draw rides in a grouped launch (GCPX_EPI_GAUSS_SAMPLE)"""
            def st(g):
                a = rt.GemmArgs()
                a.src[0] = self._rowsrc(_addr(QZ, t * 2 * nv), (T - 1) * 2 * nv, 0, 2 * nv)
                a.src[1] = self._rowsrc(_addr(tin["eps"], t * nv), tin["eps"].shape[1] * nv, 0, nv)
                a.nsrc, a.M, a.N, a.K, a.rpb, a.epi = 2, B, nv, 3 * nv, 1, rt.EPI_GAUSS_SAMPLE
                a.out, a.ob, a.orow = _addr(Z, t * nv), (T - 1) * nv, 0
                plan.keep.append(a)
                g.append((f"sample{t}", a))
            return [st]

        # folded packs in a training forward: only a trainer's model has them live (repack above); the top hidden state of step t is
        # then row block [t, nl] of the stacked layer inputs
        folds_ok = not self.save_for_backward or self._arena is not None
        gen_top = lambda t: self._rowsrc((xs_of("gen_lstm", t)[nl] if keep else self._buf(f"gen{t}.x{nl}", (B, H))).data_ptr(), H, 0, H)
        z_from_prior = not has_z and not posterior          # prior sampling: z_t needs prior(t), which needs x_t: one serial chain
        if hp.deterministic:
            # x_{t+1} = gen([x_t, e_0, e_g, a_t]): ONE chain of T - 1 steps, nl dependent launches each when layer 0 reads the previous
            # step's top hidden state through the folded output projection (x_{t+1} itself is then computed beside layer 0 of the next step)
            fold = "lstm0ff.w" in P["gen_lstm"] and folds_ok
            top = gen_top
            pending = None
            for t in range(T - 1):
                if fold and t > 0:
                    g_st = self._hsp_stages(plan, f"gen{t}", P["gen_lstm"], [top(t - 1)] + ctx(t), B, state_of("gen_lstm", t), t & 1,
                                            _addr(X, (t + 1) * nz), T * nz, nz, w0="lstm0ff", xs_buf=xs_of("gen_lstm", t))
                else:
                    g_st = gen_stages(t, None)
                if not fold:
                    run(g_st)
                    continue
                run(g_st[:1], pending)
                if len(g_st) > 2:
                    run(g_st[1:-1])
                pending = g_st[-1:]
            if pending:
                run(pending)
        elif has_traj and not (posterior and "lstm0ff.w" in P["gen_lstm"] and folds_ok):
            run(inf_stages(0))
        fold_out = posterior and "lstm0ff.w" in P["gen_lstm"] and folds_ok
        if posterior and fold_out:
            # Launch schedule of the posterior rollout.  A generator step is THREE dependent launches (layer 0 reads the previous
            # step's top hidden state through the folded output projection); everything else rides in those launches:
            #   inf(t)      stages at 3 t .. 3 t + 3        (reads encoded ground truth only: 2 steps ahead of the generator)
            #   sample(t)   at 3 t + 5                       (q(z_t) is complete after 3 t + 3)
            #   gen(t)      layers at 3 t + 6 .. 3 t + 8
            #   out(t)      = x_{t+1}, at 3 t + 9            (beside layer 0 of gen(t + 1))
            #   prior(t)    stages at 3 t + 7 .. 3 t + 10    (needs x_t = out(t - 1) from 3 t + 6; feeds only the KL term)
            sched = {}

            def place(chain, start):
                for i, st in enumerate(chain):
                    sched.setdefault(start + i, []).append(st)
            top = gen_top
            for t in range(T - 1):
                place(inf_stages(t), 3 * t)
                place(sample_stage(t), 3 * t + 5)
                zsrc = self._rowsrc(_addr(Z, t * nv), (T - 1) * nv, 0, nv)
                if t == 0:
                    g_st = gen_stages(0, zsrc)
                else:
                    g_st = self._hsp_stages(plan, f"gen{t}", P["gen_lstm"], [top(t - 1), zsrc] + ctx(t), B, state_of("gen_lstm", t), t & 1,
                                            _addr(X, (t + 1) * nz), T * nz, nz, w0="lstm0ff", xs_buf=xs_of("gen_lstm", t))
                place(g_st[:-1], 3 * t + 6)
                place(g_st[-1:], 3 * t + 9)
                place(prior_stages(t), 3 * t + 7)
            for L in sorted(sched):
                g = []
                for st in sched[L]:
                    st(g)
                self._gemm_group(plan, f"launch{L}.{g[0][0]}", g)
        elif posterior:
            # The inference net runs TWO steps ahead of the generator: q(z_{t+1}) is complete when step t starts, so its draw shares
            # a launch of step t (79 launches of ~5 us off the dependent chain); only the first draw is a launch of its own.
            if T - 1 > 1:
                run(inf_stages(1))
            z_source(0)
        for t in range(T - 1):
            if (posterior and fold_out) or hp.deterministic:
                break
            if z_from_prior:
                run(prior_stages(t))
                run(gen_stages(t, z_source(t)))
            elif posterior:
                zsrc = self._rowsrc(_addr(Z, t * nv), (T - 1) * nv, 0, nv)
                run(gen_stages(t, zsrc), inf_stages(t + 2) if t + 2 < T - 1 else None, prior_stages(t),
                    sample_stage(t + 1) if t + 1 < T - 1 else None)
            else:
                zsrc = z_source(t)                            # z is fed; q(z_t) (for the KL term) was produced one step ahead
                run(gen_stages(t, zsrc), inf_stages(t + 1) if (has_traj and t + 1 < T - 1) else None, prior_stages(t))

        # ---- latent-space heads next to the decoder ----
        plan.fork([1])
        plan.lane = 1
        if keep and folds_ok and "lstm0f.w" in P[NETS[0]]:
            # the folded steps never formed their embeddings, which the backward's weight gradients read (XS[net][t, 0]): ONE GEMM per
            # net over the stacked rows (t, b) of all steps, beside the decoder
            st = lambda ptr, w, t_stride, b_stride: self._rowsrc(ptr, t_stride, b_stride, w)
            cx = [st(_addr(X), nz, 0, T * nz), st(_addr(EG), nz, 0, nz)] if hp.context_every_step else []
            cx += [st(_addr(EA), nz, nz, (T - 1) * nz)] if EA is not None else []
            xs = st(_addr(X), nz, nz, T * nz)
            zs = [] if hp.deterministic else [st(_addr(tin["z"]) if has_z else _addr(Z), nv, nv, (T - 1) * nv)]
            per_net = {"gen_lstm": [xs] + zs + cx, "prior_lstm": [xs] + cx}
            if has_traj:
                per_net["inf_lstm"] = [st(_addr(enc_traj, nz), nz, nz, T * nz)] + cx
            for net in NETS:
                if net in per_net:
                    self._gemm(plan, f"{net}.embed_all", per_net[net], (T - 1) * B, H, B, P[net]["embed.w"], P[net]["embed.b"],
                               out=XSall[net][0, 0].data_ptr(), ob=(nl + 1) * B * H, orow=H)
        mes = self._buf("model_enc_seq", (B, T, nz))
        plan.add("gather.model_enc_seq", lib.gcpx_gather_rows, X.data_ptr(), idx.data_ptr(), mes.data_ptr(), B, T, T, 0, nz)
        outs["model_enc_seq_padded"] = mes
        if hp.run_state_regressor:
            rs = self._buf("regressed_state", (B, T, hp.state_dim))
            self._mlp(plan, "state_regressor", P["state_regressor"], [self._rowsrc(mes.data_ptr(), T * nz, nz, nz)],
                      B * T, T, out=rs.data_ptr(), ob=T * hp.state_dim, orow=hp.state_dim)
            outs["regressed_state_padded"] = rs
        aux_ok = train_aux and "inv_t0" in tin
        if hp.attach_inv_mdl and phase == "train" and (sample_prior or hp.train_inv_mdl_full_seq or not has_traj or not aux_ok):
            # InverseModel.full_seq_forward (inverse_mdl.py:110-134): val_mode sets _inv_mdl_full_seq (base_gcp.py:44-53,250)
            act = self._buf("actions", (B, T - 1, hp.n_actions))
            first = enc_traj if has_traj else mes
            s0 = self._rowsrc(first.data_ptr(), T * nz, nz, nz)
            s1 = self._rowsrc(_addr(mes, nz), T * nz, nz, nz)
            self._mlp(plan, "inv_mdl", P["inv_mdl"], [s0, s1], B * (T - 1), T - 1, out=act.data_ptr(),
                      ob=(T - 1) * hp.n_actions, orow=hp.n_actions)
            outs["actions_padded"] = act
        if aux_ok and ((hp.attach_inv_mdl and not hp.train_inv_mdl_full_seq) or (hp.attach_cost_mdl and hp.run_cost_mdl)):
            # run_auxilliary_models on ONE sampled frame pair / segment per sequence, as GCPTreeModel._build_plan (base_gcp.py:249-260)
            aux_rows = self._buf("aux_rows", (4, B), torch.int32)
            plan.add("aux_index_rows", lib.gcpx_aux_index_rows, tin["inv_t0"].data_ptr(), tin["inv_t1"].data_ptr(),
                     tin["cost_start_idx"].data_ptr(), tin["cost_end_idx"].data_ptr(), B, T, T, aux_rows.data_ptr())
            gather = lambda t, i: self._rowsrc(t.data_ptr(), 0, nz, nz, rowidx=aux_rows[i])
            if hp.attach_inv_mdl and not hp.train_inv_mdl_full_seq:
                act = self._buf("actions_sampled", (B, hp.n_actions))
                self._mlp(plan, "inv_mdl", P["inv_mdl"], [gather(enc_traj, 0), gather(mes, 1)], B, B, out=act.data_ptr(), ob=0,
                          orow=hp.n_actions)
                outs["actions_sampled"] = act
            if hp.attach_cost_mdl and hp.run_cost_mdl:
                cost = self._buf("cost_pred", (B, 1))
                self._mlp(plan, "cost_mdl", P["cost_mdl"], [gather(mes, 2), gather(mes, 3)], B, B, out=cost.data_ptr(), ob=0, orow=1)
                outs["cost_pred"], outs["cost_target"] = cost, self._buf("cost_target", (B,))
        plan.lane = 0

        # ---- decoder over the T-1 predicted latents of every sequence (sequential.py:56) ----
        F = B * (T - 1)
        prev = self._plan_decoder_features(plan, self._rowsrc(_addr(X, nz), T * nz, nz, nz), F, T - 1, skips)
        dec_images = self._buf("seq.dec_images", (B, T - 1, hp.input_nc, S, S))
        dlm = hp.decoder_distribution == "discrete_logistic_mixture"
        head_out = row_map = matched = fused_nll = None
        if dlm:
            mode = rt.HEAD_DLM_MEAN
            if with_loss or self.materialize_distr:
                # parameters of frame (b, t) land in row b*T + t + 1: aligned with the target frame traj_seq[b, t+1]
                row_map = self._buf("seq.row_map", (F,), torch.int32)
                row_map.copy_((torch.arange(B)[:, None] * T + torch.arange(1, T)[None]).reshape(-1).to(torch.int32))
                if with_loss and self._head_nll_fusable():
                    # likelihood (and, in a training forward, its gradient) in the head's epilogue, as GCPTreeModel._build_plan
                    fused_nll = self._buf("nll_partial", ((S // 4) * (S // 16), B * T), zero=True)
                    mode = rt.HEAD_DLM_NLL
                    if self.save_for_backward:
                        mode, head_out = rt.HEAD_DLM_NLL_GRAD, self._buf("bw.dMD", (B * T, S, S, self._head_pitch))
                        # rows (b, 0) of the gradient belong to no decoded frame
                        r2f = self._buf("seq.row2frame", (B * T,), torch.int32)
                        inv = torch.arange(B * T, dtype=torch.int64).view(B, T) // T * (T - 1) + torch.arange(T)[None] - 1
                        inv[:, 0] = -1
                        r2f.copy_(inv.reshape(-1).to(torch.int32))
                        plan.add("zero_unmapped", lib.gcpx_zero_unmapped_rows, head_out.data_ptr(), S * S * self._head_pitch,
                                 r2f.data_ptr(), B * T)
                        plan.rec["nll_bwd_fused"] = plan.rec["head_grad_fused"] = True
                else:
                    mode, matched = rt.HEAD_DLM_BOTH, self._buf("matched_distr", (B, T, S, S, self._head_pitch))
                    head_out = matched
        else:
            mode = rt.HEAD_TANH_NCHW
        a = self._conv_args([prev], F, S, S, S, S, hp.head_channels, self._head_pitch, P["dec.head.w"], P["dec.head.b"],
                            head_out, upsample=0, head_mode=mode, images=dec_images)
        a.raw_row_map = row_map.data_ptr() if row_map is not None else None
        if fused_nll is not None:
            a.nll_target, a.nll_partial, a.nll_rows = tin["traj_seq"].data_ptr(), fused_nll.data_ptr(), B * T
            if mode == rt.HEAD_DLM_NLL_GRAD:
                # d total / d nll_bt = w_rec * w0 / (B * prod(traj_seq.shape[1:]))
                a.nll_row_weight = tin["w0"].data_ptr()
                a.nll_scale = hp.dense_img_rec_weight / (B * float(T * hp.input_nc * S * S))
        plan.keep.append(a)
        plan.join([1])
        self._set_split(a, "dec.head")
        plan.add("dec.head", lib.gcpx_conv3x3, C.byref(a))
        # images = cat(I_0, decoded) (sequential.py:57)
        row = hp.input_nc * S * S
        images = self._buf("seq.images", (B, T, hp.input_nc, S, S))
        plan.add("cat.I0", lib.gcpx_copy_rows, tin["I_0"].data_ptr(), images.data_ptr(), B, 1, row, 1, T)
        plan.add("cat.dec", lib.gcpx_copy_rows, dec_images.data_ptr(), _addr(images, row), B, T - 1, row, T - 1, T)
        outs.update(images=images, X=X, PZ=PZ, QZ=QZ, Z=Z, seq_len=seq_len, matched_distr_kernel_order=matched, e_g=EG)

        # ---- losses (sequential.py:60-68) ----
        if with_loss:
            nll_bt = self._buf("nll_bt", (B, T))
            if dlm and fused_nll is not None:
                # rows (b, 0) are written by no frame: they stay zero and weigh 0 in the combination below
                plan.add("loss.nll_reduce", lib.gcpx_reduce_partials, fused_nll.data_ptr(), fused_nll.shape[0], B * T, B * T,
                         nll_bt.data_ptr(), 0)
            elif dlm and self.save_for_backward:
                # training step: loss and its gradient w.r.t. the stored parameters in one pass (see GCPTreeModel._build_plan);
                # d total / d nll_bt = w_rec * w0 / (B * prod(traj_seq.shape[1:]))
                dMD = self._buf("bw.dMD", (B * T, S, S, self._head_pitch))
                plan.add("loss.dlm_nll+bwd", lib.gcpx_dlm_nll_bwd, matched.data_ptr(), tin["traj_seq"].data_ptr(), tin["w0"].data_ptr(),
                         C.c_float(hp.dense_img_rec_weight / (B * float(T * hp.input_nc * S * S))), dMD.data_ptr(),
                         self._buf("bw.dMD.colsum", (B * T, self._head_pitch)).data_ptr(), nll_bt.data_ptr(), B * T, S * S,
                         self._head_pitch, hp.n_mixtures)
                plan.rec["nll_bwd_fused"] = True
            elif dlm:
                plan.add("loss.dlm_nll", lib.gcpx_dlm_nll, matched.data_ptr(), tin["traj_seq"].data_ptr(), tin["w0"].data_ptr(),
                         nll_bt.data_ptr(), B * T, S * S, self._head_pitch, hp.n_mixtures)
            else:
                # rows (b, 0) compare I_0 with itself-as-target and carry weight 0 in the combine below
                plan.add("loss.gauss_nll", lib.gcpx_gauss_nll, images.data_ptr(), tin["traj_seq"].data_ptr(),
                         self.sd["decoder.log_sigma"].data_ptr(), nll_bt.data_ptr(), B * T, row)
            kl_b = self._buf("kl_b", (B,), zero=hp.deterministic)
            if not hp.deterministic:                     # (no latent: both distributions are empty and the KL term is 0)
                plan.add("loss.kl", lib.gcpx_kl_gauss, QZ.data_ptr(), PZ.data_ptr(), B, T - 1, nv, (T - 1) * 2 * nv, 2 * nv,
                         C.c_float(hp.free_nats), _addr(tin["pad_mask"], 1), T, kl_b.data_ptr())
            la = rt.LossArgs()
            la.nll_bt, la.pad_mask, la.kl_b = nll_bt.data_ptr(), tin["w0"].data_ptr(), kl_b.data_ptr()   # frame 0 weighs 0
            la.len_logits = outs["seq_len_logits"].data_ptr() if "seq_len_logits" in outs else None
            la.end_ind, la.seq_len = tin["end_ind"].data_ptr(), seq_len.data_ptr()
            loss_out = self._buf("losses", (16,), zero=True)
            la.out, la.B, la.T, la.N, la.state_dim = loss_out.data_ptr(), B, T, T - 1, hp.state_dim
            la.w_rec, la.w_kl, la.w_len, la.w_exist, la.w_state = hp.dense_img_rec_weight, hp.kl_weight, hp.length_pred_weight, 0.0, 0.0
            if self._kl_w is not None:                   # burn-in schedule (the 25-room gcp_sequential conf sets kl_weight_burn_in)
                la.w_kl_dev = self._kl_w.data_ptr()
            if "regressed_state_padded" in outs and "traj_seq_states" in tin:
                # state regression over the frames of the sequence, frame 0 included (base_gcp.py:281-286)
                la.regressed_state, la.state_target = outs["regressed_state_padded"].data_ptr(), tin["traj_seq_states"].data_ptr()
                la.state_mask, la.w_state = tin["pad_mask"].data_ptr(), 1.0
            if "actions_sampled" in outs and "actions" in tin:          # inverse_mdl.py:181-191
                la.action_pred, la.action_seq, la.inv_t0 = outs["actions_sampled"].data_ptr(), tin["actions"].data_ptr(), tin["inv_t0"].data_ptr()
                la.n_actions, la.w_action = hp.n_actions, hp.action_rec_weight
            if "cost_pred" in outs:                                     # cost_mdl.py:59-62
                la.cost_pred, la.cost_target, la.w_cost = outs["cost_pred"].data_ptr(), outs["cost_target"].data_ptr(), 1.0
            la.total_div = float(T * hp.input_nc * S * S)
            plan.keep.append(la)
            plan.add("loss.combine", lib.gcpx_loss_combine, C.byref(la))
            outs["losses"], outs["nll_bt"], outs["kl_b"] = loss_out, nll_bt, kl_b
            plan.rec["loss_args"] = la
        outs["enc_traj_seq"] = enc_traj
        plan.rec.update(head_src=prev, tin=tin, key=key, seq_row_map=row_map)
        plan.outs = outs
        return plan

    # ------------------------------------------------------------------------------------------------
    def forward(self, inputs, phase="train", noise=None):
        """noise: eps [B, T-1, nz_vae] for the per-step Gaussian samples; inputs['z'] [B, T-1, nz_vae] feeds latents."""
        if self._hp.non_goal_conditioned:
            # optional_preprocessing (base_gcp.py:163-170): the goal image and the sequence's end frame are zeroed before anything is
            # encoded; the reference edits `inputs` in place (so its losses see the zeroed frame too), here the caller's tensors stay
            inputs = dict(inputs, I_g=torch.zeros_like(torch.as_tensor(inputs["I_g"])))
            if "traj_seq" in inputs:
                ts = torch.as_tensor(inputs["traj_seq"]).to(self.device, torch.float32, copy=True)
                ts[torch.arange(ts.shape[0], device=self.device), torch.as_tensor(inputs["end_ind"]).to(self.device)] = 0.0
                inputs["traj_seq"] = ts
        if "pad_mask" in inputs and "traj_seq" in inputs and phase == "train":
            # NLL row weights: frame 0 is the conditioning frame, never reconstructed (sequential.py:63-66)
            pm = inputs["pad_mask"].to(self.device, torch.float32)
            w0 = pm.clone()
            w0[:, 0] = 0
            inputs = dict(inputs, w0=w0)
        return super().forward(inputs, phase, noise)

    def _wrap_outputs(self, o, tin, phase):
        out = Outputs()
        out.end_ind, out.raw = tin["end_ind"], o
        out.dense_rec = Outputs(images=o["images"], encodings=o["X"][:, 1:], p_z=o["PZ"], q_z=o["QZ"], z=o["Z"])
        if "seq_len_logits" in o:
            out.seq_len_logits = o["seq_len_logits"]
        return out

    def pruned_prediction(self, out):
        lens = out.raw["seq_len"].tolist()
        return [out.raw["images"][b, :lens[b]] for b in range(len(lens))]

    LOSS_NAMES = ("dense_img_rec", "kl", "len_pred", "state_regression", "action_reconst", "cost_estimation")

    def loss(self, inputs, outputs, log_error_arr=False):
        raw = outputs.raw
        if "losses" not in raw:
            raise ValueError("losses need traj_seq and pad_mask in the inputs of a phase='train' forward")
        hp, lv = self._hp, raw["losses"]
        res = Outputs()
        res["dense_img_rec"] = Outputs(value=lv[0], weight=hp.dense_img_rec_weight)
        res["kl"] = Outputs(value=lv[1], weight=self.kl_weight_now)
        if hp.regress_length:
            res["len_pred"] = Outputs(value=lv[2], weight=hp.length_pred_weight)
        if "regressed_state_padded" in raw and "traj_seq_states" in inputs:
            res["state_regression"] = Outputs(value=lv[4], weight=1.0)
        if raw.get("actions_sampled") is not None and "actions" in inputs:      # base_gcp.py:275-276, inverse_mdl.py:181-191
            res["action_reconst"] = Outputs(value=lv[7], weight=hp.action_rec_weight)
        if raw.get("cost_pred") is not None:                                    # base_gcp.py:279-280, cost_mdl.py:59-62
            res["cost_estimation"] = Outputs(value=lv[8], weight=1.0)
        res["_total"] = lv[5]
        return res
