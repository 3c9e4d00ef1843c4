"""Training step of the flat VRNN baseline `gcp_sequential` on MI355X: forward + explicit backward through the T - 1 recurrent
steps + RAdam, all in HIP.

The reference trains `SequentialModel` through the same loop as the tree model (/root/reference/gcp/prediction/train.py:155-163 with
`configuration['model'] = SequentialModel`, experiments/prediction/base_configs/gcp_sequential.py; model:
gcp/prediction/models/sequential.py:13-131) and gets the backward pass from torch autograd.  Here it is a second launch plan built
from the records of the training forward (`GCPSequentialModel._build_plan` with save_for_backward: every step's LSTM states, layer
inputs and gates are kept), like `GCPTrainStep`'s for the tree model, whose decoder / encoder / Predictor backward it reuses:

  loss gradients (NLL on frames 1 .. T-1, KL weighted by pad_mask[:, 1:], length CE; sequential.py:60-68, base_gcp.py:264-304)
  -> decoder backward                                              (gradient of every x_{t+1})
  -> prior chain   t = T-2 .. 0   (needs only the KL gradient: runs on a side lane next to the decoder backward)
  -> generator chain t = T-2 .. 0 (x_{t+1} = gen([x_t, z_t, ctx]): the gradient of x_t collects decoder, prior and generator terms)
  -> reparametrised sample backward for all steps in one launch, inference chain t = T-2 .. 0
  -> weight gradients of the three recurrent nets: ONE GEMM per weight over the stacked (step, sequence) rows
  -> encoders (trajectory frames through the inference net's inputs, I_0 through x_0 / context, I_g through the context).

A step of a chain is: out^T, n x (LSTM-cell backward, [W_ih | W_hh]^T in one GEMM), embed^T — eight dependent small launches at
16 rows.  The three nets' chains are each T - 1 steps long; they are latency-bound (weights stream from L2 / MALL), not MFMA-bound.

Variants of base_configs/vmpc.py (GCPSequentialModel._nets, hparams.action_conditioned_pred / deterministic): a deterministic predictor
has only the generator chain (no KL, no latent backward, no prior / inference chain); with action conditioning the last nz_enc input
columns of every net are the encoded action, whose gradient — summed over the nets — goes through the action encoder's Predictor backward.
"""
import ctypes as C

import torch

from . import packing as pk
from . import runtime as rt
from .model import _Plan, _addr
from .params import decoder_layers
from .training import GCPTrainStep, _c16


class SequentialTrainStep(GCPTrainStep):
    """`step(inputs)` = one optimisation step of a GCPSequentialModel on one minibatch (same interface as GCPTrainStep)."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None, **optim):
        assert model._hp.context_every_step, "the recurrent nets are built with the (e_0, e_g) context at every step (hyperparameters.py default)"
        super().__init__(model, lr=lr, betas=betas, eps=eps, process_group=process_group, **optim)
        self.side_lanes = True
        self.wgrad_per_cu = int(__import__("os").environ.get("GCPX_WGRAD_PER_CU", "0"))   # (full occupancy here: backward_ops._wgrad_conv3)
        # GCPX_SEQ_CHAINS: "lockstep" (default): prior chain ahead on its own lane, generator step t and inference step t + 1 in the SAME
        # five launches (gcpx_gemm_group) on the main lane; "overlap": three lanes, step-by-step events; "serial": round-5 order
        # "lockstep3": the prior net's step t rides in the same launches too (three problems each; no side chain, no events at all)
        mode = __import__("os").environ.get("GCPX_SEQ_CHAINS", "lockstep")
        self.chains_overlap = mode in ("overlap", "lockstep", "lockstep3")
        self.chains_lockstep = mode in ("lockstep", "lockstep3")
        self.chains_lockstep3 = mode == "lockstep3"

    # ------------------------------------------------------------------------------------------------
    def _pack_backward(self, sd):
        m, hp = self.m, self.m._hp
        nz = hp.nz_enc
        X = {}
        layers, ctop = m._enc_layers, m._c_top
        for name, cin, cout, norm in layers[1:]:
            w = sd[f"encoder.net.{name}.conv.weight"]
            X[f"enc.{name}.wT"] = pk.pack_gemm(w.permute(2, 3, 1, 0).reshape(16 * cin, cout))
        wh = sd["encoder.net.head.weight"]
        X["enc.head.wT"] = pk.pack_gemm(wh.permute(2, 3, 1, 0).reshape(16 * ctop, nz))
        wt = sd["decoder.net.input.conv.weight"]
        X["dec.input.wT"] = pk.pack_gemm(wt.permute(0, 2, 3, 1).reshape(nz, 16 * ctop))
        for name, c_prev, c_skip, skip_idx, cout in decoder_layers(hp):
            wT = sd[f"decoder.net.{name}.conv.weight"].flip(2, 3).transpose(0, 1).contiguous()
            for h in range((wT.shape[0] + 63) // 64):
                X[f"dec.{name}.wT{h}"] = pk.pack_conv3x3(wT[64 * h:64 * (h + 1)], 16)
        hw = sd["decoder.gen_head.conv.weight"]
        perm = torch.as_tensor(pk.dlm_channel_perm(hp.n_mixtures), device=hw.device)
        wk = torch.zeros((len(perm),) + tuple(hw.shape[1:]), dtype=hw.dtype, device=hw.device)
        wk[perm >= 0] = hw[perm[perm >= 0]]
        X["dec.head.wT"] = pk.pack_conv3x3(wk.flip(2, 3).transpose(0, 1).contiguous(), 16)
        if hp.regress_length:
            X["length_pred"] = self._pack_predictor_T(sd, "length_pred.p", [(0, 2 * nz)])
        if hp.attach_state_regressor:
            X["state_regressor"] = self._pack_predictor_T(sd, "state_regressor", [])
        if hp.attach_inv_mdl:
            X["inv_mdl"] = self._pack_predictor_T(sd, "inv_mdl.action_pred", [])
        if hp.attach_cost_mdl:
            X["cost_mdl"] = self._pack_predictor_T(sd, "cost_mdl.cost_pred", [])
        if hp.action_conditioned_pred:
            X["action_encoder"] = self._pack_predictor_T(sd, "action_encoder", [])      # its input, the actions, takes no gradient
        for net in m._nets:
            p = f"dense_rec.lstm.cell.{net}"
            T_ = {"embed.wT": pk.pack_gemm(sd[f"{p}.embed.weight"].t().contiguous()),          # [n = in_dim][k = H]
                  "out.wT": pk.pack_gemm(sd[f"{p}.out.weight"].t().contiguous())}               # [n = H][k = out]
            if net == "gen_lstm" and len(m._nets) > 1:
                # the same weights three times along k: d out = [d x (decoder) | d x (prior's input) | d x (generator's input)] @ [W^T; W^T; W^T]
                # sums the three gradient sources of x_{t+1} inside the GEMM: the generator chain need not wait for the prior chain's
                # LAST step, and no add-rows launch sits between two of its steps
                wt = sd[f"{p}.out.weight"].t().contiguous()
                T_["out.wT3"] = pk.pack_gemm(torch.cat([wt, wt, wt], 1).contiguous())
            for i in range(hp.n_lstm_layers):
                # d [x | h] = d gates @ [W_ih | W_hh]: both data gradients of a cell in one GEMM
                w = torch.cat([sd[f"{p}.lstm.{i}.weight_ih"], sd[f"{p}.lstm.{i}.weight_hh"]], 1)          # [4H, 2H]
                T_[f"lstm{i}.wxhT"] = pk.pack_gemm(w.t().contiguous())                                    # [n = 2H][k = 4H]
            X[net] = T_
        return X

    # ------------------------------------------------------------------------------------------------
    def _rows(self, plan, name, dst, dst_sb, dst_sr, src, src_sb, src_sr, B, rpb, width, mode):
        plan.add(name, self.m.lib.gcpx_rows_strided, dst, dst_sb, dst_sr, src, src_sb, src_sr, B, rpb, width, mode)

    def _chain(self, plan, net, dout_of, B, T, dIn, nrec, events=None, before_step=None, slots=None, event_every=1):
        """Backward of one recurrent net through all T - 1 steps, last step first.  dout_of(t) -> row source of the gradient of the
        step's output.  Writes dIn [B, T-1, in_dim] (gradient of every step's embedding input), the stacked gate gradients dG[i]
        [(T-1) B, 4H] and embedding-output gradients DX0 [(T-1) B, 2H] (columns < H) for the weight gradients.
        dout_of(t) may return a LIST of three row sources (summed by the tripled `out.wT3`).  events: dict filled with one recorded event per
        finished step (on the lane the chain is issued on; event_every = n: only behind steps t % n == 0); before_step(t): called before
        step t's first launch.  slots: nl + 2 lists — the step's GEMMs (cell backward in their epilogues) are appended to them, one per
        dependent launch of the step, instead of being launched: the caller groups them with another chain's (`_gemm_group`)."""
        m, hp, lib = self.m, self.m._hp, self.m.lib
        H, nl = hp.nz_mid_lstm, hp.n_lstm_layers
        rec, Wt = nrec["rec"], self.bk[net]
        S, in_dim = nrec["S"][net], nrec["in_dim"][net]
        buf = m._buf
        dG = [buf(f"bw.{net}.dG{i}", (T - 1, B, 4 * H)) for i in range(nl)]
        dxh = [buf(f"bw.{net}.dxh{i}", (B, 2 * H)) for i in range(1, nl)]              # layers 1..: [dx | dh], reused every step
        DX0 = buf(f"bw.{net}.dxh0", (T - 1, B, 2 * H))                                 # layer 0: kept per step (embedding weight gradient)
        dtop = buf(f"bw.{net}.dtop", (B, H))
        dcrec = [buf(f"bw.{net}.dc{i}", (B, 2 * H)) for i in range(nl)]                 # (pitch 2H: addressed with the strides of [h | c])
        fuse_cell = self.fuse_lstm_bwd or slots is not None   # a layer's cell backward in the epilogue of the GEMM that produces its d h (training.py)
        slot = (lambda k: slots[k]) if slots is not None else (lambda k: None)
        for t in reversed(range(T - 1)):
            last = t == T - 2
            cells = []
            for i in range(nl):
                above = dtop if i == nl - 1 else dxh[i]                                 # d h_i from the layer above (its dx columns)
                a = rt.LstmBwdArgs()
                a.gates = rec[f"gates:{nrec['tag'][net]}{t}.lstm{i}"].data_ptr()
                a.c_prev, a.c_prev_stride = _addr(S[t, i], H), 2 * H
                a.c_new, a.pb, a.prow = _addr(S[t + 1, i], H), 2 * H, 0
                a.dh_dense, a.dh_stride = above.data_ptr(), (H if i == nl - 1 else 2 * H)
                # recurrent terms from step t + 1: d h through W_hh (the dh columns of that step's [dx | dh]), d c through the forget gate
                prev_out = (DX0[t + 1] if i == 0 else dxh[i - 1]) if not last else None
                a.dh_pos = _addr(prev_out, H) if prev_out is not None else None
                a.dc_pos = dcrec[i].data_ptr() if not last else None
                # (dh_pos / dc_pos are addressed with the strides of c_new, 2H per row: the dh columns of [dx | dh] and the dc buffer
                # have that pitch; a thread reads its dc_pos element before it overwrites it with dc_prev)
                a.dgates, a.dc_prev, a.dcp_stride = dG[i][t].data_ptr(), dcrec[i].data_ptr(), 2 * H
                a.M, a.H, a.rpb = B, H, 1
                plan.keep.append(a)
                cells.append(a)
            if before_step is not None:
                before_step(t)
            dsrc = dout_of(t)
            dsrc = dsrc if isinstance(dsrc, list) else [dsrc]
            self._dgemm(plan, f"{net}{t}.out", dsrc, B, H, 1, Wt["out.wT" if len(dsrc) == 1 else "out.wT3"], dtop.data_ptr(), H, 0,
                        lstm_bwd=(cells[nl - 1] if fuse_cell else None), group=slot(0))
            for i in reversed(range(nl)):
                out_i = DX0[t] if i == 0 else dxh[i - 1]
                if not fuse_cell:
                    plan.add(f"bw.lstm:{net}{t}.{i}", lib.gcpx_lstm_bwd, C.byref(cells[i]))
                self._dgemm(plan, f"{net}{t}.lstm{i}", [self._dense(dG[i][t].data_ptr(), 4 * H, 4 * H, B)], B, 2 * H, B, Wt[f"lstm{i}.wxhT"],
                            out_i.data_ptr(), 0, 2 * H, lstm_bwd=(cells[i - 1] if (fuse_cell and i > 0) else None), group=slot(nl - i))
            self._dgemm(plan, f"{net}{t}.embed", [m._rowsrc(DX0[t].data_ptr(), 2 * H, 0, H)], B, in_dim, 1, Wt["embed.wT"],
                        _addr(dIn, t * in_dim), (T - 1) * in_dim, 0, group=slot(nl + 1))
            if events is not None and t % event_every == 0:
                events[t] = plan.record(plan.lane)
            yield t
        nrec.setdefault("dG", {})[net] = dG
        nrec.setdefault("DX0", {})[net] = DX0

    def _build_backward(self, fplan):
        m, hp, lib = self.m, self.m._hp, self.m.lib
        rec, o = fplan.rec, fplan.outs
        key, tin = rec["key"], rec["tin"]
        B = key[0]
        T, nz, nv, H, nl = hp.max_seq_len, hp.nz_enc, hp.nz_vae, hp.nz_mid_lstm, hp.n_lstm_layers
        S_ = hp.img_sz
        div = float(T * hp.input_nc * S_ * S_)
        plan = _Plan(lib)
        buf = m._buf
        sq = rec["seq"]
        X, EG, PZ, QZ, Z = sq["X"], sq["EG"], sq["PZ"], sq["QZ"], sq["Z"]
        enc_traj = o["enc_traj_seq"]
        zero = lambda t: plan.add("bw.zero", lib.gcpx_fill_zero, t.data_ptr(), t.numel() * 4)
        NETS = m._nets                                       # ('gen_lstm',) for a deterministic predictor (no latent: no KL, no prior / inference net)
        det, EA = hp.deterministic, sq.get("EA")
        na_ = nz if EA is not None else 0                     # the encoded action: last nz columns of every net's input (sequential.py:45-49)
        in_dim = {"prior_lstm": 3 * nz + na_, "inf_lstm": 3 * nz + na_, "gen_lstm": 3 * nz + nv + na_}
        nrec = dict(rec=rec, S=sq["S"], XS=sq["XS"], in_dim=in_dim, tag={"prior_lstm": "prior", "inf_lstm": "inf", "gen_lstm": "gen"})

        DX = buf("bw.seq.DX", (B, T, nz))                    # gradient of X[b, t] = x_t (x_0 = e_0)
        dEG = buf("bw.seq.dEG", (B, nz))
        dQZ, dPZ = buf("bw.seq.dQZ", (B, T - 1, 2 * nv)), buf("bw.seq.dPZ", (B, T - 1, 2 * nv))
        dIn = {net: buf(f"bw.{net}.dIn", (B, T - 1, in_dim[net])) for net in NETS}
        zero(self.grad); zero(DX); zero(dEG)

        # ---- loss gradients ----
        la = rec["loss_args"]
        assert rec.get("nll_bwd_fused"), "the training forward produces d NLL / d parameters together with the loss"
        dMD = buf("bw.dMD", (B * T, S_, S_, m._head_pitch))
        if det:
            pass                                              # no latent: no KL term, dQZ / dPZ are empty
        elif m._kl_w is not None:                               # burn-in schedule: kl_weight(step) is read from device memory
            plan.add("bw.kl", lib.gcpx_kl_bwd_scheduled, QZ.data_ptr(), PZ.data_ptr(), dQZ.data_ptr(), dPZ.data_ptr(), B, T - 1, nv,
                     (T - 1) * 2 * nv, 2 * nv, C.c_float(hp.free_nats), C.c_float(1.0 / (B * div)), _addr(tin["pad_mask"], 1), T,
                     m._kl_w.data_ptr())
        else:
            plan.add("bw.kl", lib.gcpx_kl_bwd_weighted, QZ.data_ptr(), PZ.data_ptr(), dQZ.data_ptr(), dPZ.data_ptr(), B, T - 1, nv,
                     (T - 1) * 2 * nv, 2 * nv, C.c_float(hp.free_nats), C.c_float(hp.kl_weight / (B * div)), _addr(tin["pad_mask"], 1), T)
        ldl = _c16(T)
        has_state = bool(la.regressed_state)
        dlen = buf("bw.dlen", (B, ldl)) if hp.regress_length else None
        dstate = buf("bw.dstate", (B * T, 16)) if has_state else None
        if hp.regress_length or has_state:
            plan.add("bw.heads", lib.gcpx_loss_heads_bwd, C.byref(la), rt.ptr(dlen), None, rt.ptr(dstate))
        if has_state:   # input detached (base_gcp.py:253-256): parameter gradients only
            self._mlp_bwd(plan, "state_regressor", "state_regressor", rec["mlp:state_regressor"], self.bk["state_regressor"],
                          dstate.data_ptr(), 16, [])
        # inverse model / cost model on the sampled pairs: inputs detached (inverse_mdl.py:160-162, cost_mdl.py:108-109) — parameter
        # gradients only
        has_inv, has_cost = bool(la.action_pred), bool(la.cost_pred)
        if has_inv or has_cost:
            daction = buf("bw.daction", (B, 16)) if has_inv else None
            dcost = buf("bw.dcost", (B, 16)) if has_cost else None
            plan.add("bw.aux_heads", lib.gcpx_loss_aux_heads_bwd, C.byref(la), rt.ptr(daction), rt.ptr(dcost))
            if has_inv:
                self._mlp_bwd(plan, "inv_mdl", "inv_mdl.action_pred", rec["mlp:inv_mdl"], self.bk["inv_mdl"], daction.data_ptr(), 16, [])
            if has_cost:
                self._mlp_bwd(plan, "cost_mdl", "cost_mdl.cost_pred", rec["mlp:cost_mdl"], self.bk["cost_mdl"], dcost.data_ptr(), 16, [])
        if hp.regress_length:
            dXl = buf("bw.dX.len", (B, 2 * nz))
            self._mlp_bwd(plan, "length_pred", "length_pred.p", rec["mlp:length_pred"], self.bk["length_pred"], dlen.data_ptr(), ldl,
                          [(dXl.data_ptr(), 2 * nz, 0)])
            self._rows(plan, "bw.len.e0", DX.data_ptr(), T * nz, 0, dXl.data_ptr(), 2 * nz, 0, B, 1, nz, 1)
            self._rows(plan, "bw.len.eg", dEG.data_ptr(), nz, 0, _addr(dXl, nz), 2 * nz, 0, B, 1, nz, 1)

        DQ, DPd = buf("bw.seq.DQ", (T - 1, B, 2 * nv)), buf("bw.seq.DPd", (B, 2 * nv))
        # ---- weight gradients of a recurrent net: ONE GEMM per weight over the stacked rows r = (t, b) of all its steps ----
        def stacked(t_stride, b_stride):
            return dict(rpb=B, sb=t_stride, sr=b_stride)
        e0s = dict(ptr=_addr(X), **stacked(0, T * nz))
        egs = dict(ptr=_addr(EG), **stacked(0, nz))
        eas = [dict(ptr=_addr(EA), w=nz, **stacked(nz, (T - 1) * nz))] if EA is not None else []
        zs_ = [dict(ptr=_addr(Z), w=nv, **stacked(nv, (T - 1) * nv))] if not det else []
        srcs = {"gen_lstm": [dict(ptr=_addr(X), w=nz, **stacked(nz, T * nz))] + zs_ + [dict(w=nz, **e0s), dict(w=nz, **egs)] + eas}
        if not det:
            srcs["prior_lstm"] = [dict(ptr=_addr(X), w=nz, **stacked(nz, T * nz)), dict(w=nz, **e0s), dict(w=nz, **egs)] + eas
            srcs["inf_lstm"] = [dict(ptr=_addr(enc_traj, nz), w=nz, **stacked(nz, T * nz)), dict(w=nz, **e0s), dict(w=nz, **egs)] + eas
        # (gradient of a net's output, rows (t, b): pointer, pitch of b, stride of t — 0 = dense t-major rows —, width)
        douts = {"prior_lstm": (dPZ.data_ptr(), (T - 1) * 2 * nv, 2 * nv, 2 * nv), "inf_lstm": (DQ.data_ptr(), 2 * nv, 0, 2 * nv),
                 "gen_lstm": (_addr(DX, nz), T * nz, nz, nz)}
        R = (T - 1) * B
        wgrads_out = set()

        def net_wgrads(net):
            wgrads_out.add(net)
            p = f"dense_rec.lstm.cell.{net}"
            XS, S = sq["XS"][net], sq["S"][net]
            dG, DX0 = nrec["dG"][net], nrec["DX0"][net]
            dy, ldy, dy_sb, N_out = douts[net]
            self._wgrad(plan, f"{net}.out", dy, ldy, R, N_out, XS[0, nl].data_ptr(), H, self.g(f"{p}.out.weight"), ldw=H, rpb=B,
                        sb=(nl + 1) * B * H, sr=H, dy_rpb=(B if dy_sb else 0), dy_sb=dy_sb, dbias=self.g(f"{p}.out.bias"))
            for i in range(nl):
                self._wgrad(plan, f"{net}.lstm{i}.ih", dG[i].data_ptr(), 4 * H, R, 4 * H, XS[0, i].data_ptr(), H, self.g(f"{p}.lstm.{i}.weight_ih"),
                            ldw=H, rpb=B, sb=(nl + 1) * B * H, sr=H, dbias=self.g(f"{p}.lstm.{i}.bias_ih"), dbias2=self.g(f"{p}.lstm.{i}.bias_hh"))
                self._wgrad(plan, f"{net}.lstm{i}.hh", dG[i].data_ptr(), 4 * H, R, 4 * H, S[0, i].data_ptr(), H, self.g(f"{p}.lstm.{i}.weight_hh"),
                            ldw=H, rpb=B, sb=nl * B * 2 * H, sr=2 * H)
            koff = 0
            for k, sc in enumerate(srcs[net]):
                self._wgrad(plan, f"{net}.embed{k}", DX0.data_ptr(), 2 * H, R, H, sc["ptr"], sc["w"], self.g(f"{p}.embed.weight"),
                            ldw=in_dim[net], k_off=koff, rpb=sc["rpb"], sb=sc["sb"], sr=sc["sr"],
                            dbias=(self.g(f"{p}.embed.bias") if k == 0 else None))
                koff += sc["w"]

        # ---- prior chain on a side lane (needs the KL gradient only), decoder backward on the main lane ----
        # chains_overlap: the generator's step t needs the prior's input gradient of step t + 1 only, so it follows the prior chain
        # step by step (one event per prior step) instead of waiting for its last one: the prior chain starts beside the decoder
        # backward as before, the generator starts when the decoder backward ends, wherever the prior chain is by then
        overlap = self.chains_overlap and not det
        lockstep = overlap and self.chains_lockstep
        lock3 = lockstep and self.chains_lockstep3
        prior_done = {} if (overlap and not lock3) else None
        PE = 4 if lockstep else 1                             # the prior chain hands over every PE steps (it is far ahead anyway)

        def prior_chain():
            plan.lane = 1
            i0 = len(plan.ops)
            if not det:
                for _ in self._chain(plan, "prior_lstm", lambda t: m._rowsrc(_addr(dPZ, t * 2 * nv), (T - 1) * 2 * nv, 0, 2 * nv), B, T, dIn["prior_lstm"],
                                     nrec, events=prior_done, event_every=PE):
                    pass
            plan.lane = 0
            return i0
        if lock3:
            pass                                              # (the prior's steps are issued with the generator's, below)
        elif overlap:
            # the HOST issues the plan in order: the decoder backward's launches go out first (the main lane starts at once), the prior
            # chain's ~870 host calls follow while the device is busy with them; lane 1 starts from where the main lane stood HERE
            prior_start = plan.record(0)
        else:
            plan.fork([1])
            prior_chain()
        F = B * (T - 1)
        row2frame = buf("bw.seq.row2frame", (B * T,), torch.int32)       # row (b, t) of the matched arrays <- decoded frame (b, t - 1); (b, 0): none
        r2f = torch.arange(B * T, dtype=torch.int64).view(B, T)
        row2frame.copy_(torch.where(r2f % T == 0, r2f // T * (T - 1), r2f // T * (T - 1) + r2f % T - 1).reshape(-1).to(torch.int32))
        inv = torch.arange(B * T, dtype=torch.int64).view(B, T) // T * (T - 1) + torch.arange(T)[None] - 1
        inv[:, 0] = -1
        row2frame_inv = buf("bw.seq.row2frame_inv", (B * T,), torch.int32)
        row2frame_inv.copy_(inv.reshape(-1).to(torch.int32))
        dE_dec, dskip = self._decoder_backward(plan, fplan, dMD, B, maps=dict(R=B * T, row2src=row2frame, frame2row=rec["seq_row_map"],
                                                                               row2frame=row2frame_inv))
        pd = in_dim["prior_lstm"]
        prior_dx = lambda: self._rows(plan, "bw.prior.dx", DX.data_ptr(), T * nz, nz, dIn["prior_lstm"].data_ptr(), (T - 1) * pd, pd, B, T - 1, nz, 1)
        if lock3:
            self._flush(plan, only_lane=1)                    # decoder weight gradients on lane 1, beside the one chain
            inf_lane = 2
        elif overlap:
            plan.await_event(1, prior_start)
            i0 = prior_chain()
            # the prior chain has hours of slack (it runs beside the decoder backward and only has to stay ahead of the generator): its
            # steps are replayed as small linear graphs, one host call per step instead of eight (training.py: segment_ranges)
            plan.rec.setdefault("segment_ranges", []).append((i0, len(plan.ops)))
            net_wgrads("prior_lstm")                          # due at the end of the step: behind the chain on its lane, with the decoder's
                                                              # (measured: no gain, no loss — the tail is the trajectory encoder's backward)
            # decoder weight gradients behind the prior chain on lane 1 (they are due at the end of the step only); lane 2 is the
            # inference chain's
            self._flush(plan, only_lane=1)                    # (in the tail instead, beside the encoder backward: +0.5 ms)
            inf_lane = 2
            plan.fork([inf_lane])
        else:
            plan.join([1])
            self._flush(plan, only_lane=2)                    # decoder weight gradients: on lane 2, beside the generator / inference chains
            inf_lane = 1
        # gradient of x_{t+1}, t = 0 .. T-2: decoder + the prior's input at step t + 1
        plan.add("bw.addrows.dec", lib.gcpx_add_rows, _addr(DX, nz), T * nz, nz, dE_dec.data_ptr(), None, B, T - 1, nz)
        if not det and not overlap:
            prior_dx()

        # ---- generator chain on the main lane, inference chain one step behind it on lane 1 ----
        # gen(t): the gradient of x_t is complete once step t has added its input gradient.  z_t = mu_q + exp(log_sigma_q) eps
        # (sequential.py:51-54): step t's d z_t turns into d q_t right away (one tiny launch), lane 1 waits for exactly that and runs the
        # inference net's step t while the generator goes on to step t - 1 — the two 79-step chains overlap instead of queueing.
        gd = in_dim["gen_lstm"]
        inf_chain = iter(()) if det else \
            self._chain(plan, "inf_lstm", lambda t: m._rowsrc(DQ[t].data_ptr(), 2 * nv, 0, 2 * nv), B, T, dIn["inf_lstm"], nrec)

        def gen_dout(t):
            own = m._rowsrc(_addr(DX, (t + 1) * nz), T * nz, 0, nz)
            if not overlap or t + 1 > T - 2:                  # (the last frame is no input of the prior)
                return own
            return [own, m._rowsrc(_addr(dIn["prior_lstm"], (t + 1) * pd), (T - 1) * pd, 0, nz),
                    m._rowsrc(_addr(dIn["gen_lstm"], (t + 1) * gd), (T - 1) * gd, 0, nz)]

        awaited = set()

        def gen_before(t):
            if overlap and not lock3 and t + 1 <= T - 2:
                tp = (t + 1) // PE * PE                        # the hand-over that covers the prior's step t + 1 (it counts t downwards)
                if tp not in awaited:
                    awaited.add(tp)
                    plan.await_event(0, prior_done[tp])

        def latent(t):
            plan.add(f"bw.latent{t}", lib.gcpx_latent_bwd, _addr(dQZ, t * 2 * nv), _addr(dPZ, t * 2 * nv), _addr(QZ, t * 2 * nv), (T - 1) * 2 * nv, 0,
                     _addr(tin["eps"], t * nv), tin["eps"].shape[1] * nv, 0, _addr(dIn["gen_lstm"], t * gd + nz), (T - 1) * gd, None, 0,
                     DQ[t].data_ptr(), DPd.data_ptr(), B, 1, nv)
        if lockstep:
            # super-step k: generator step T-2-k and inference step T-1-k (one behind: it needs that step's d z) share five launches on
            # the main lane — no lane of its own, no events between the two chains
            slots = [[] for _ in range(nl + 2)]
            gen_chain = self._chain(plan, "gen_lstm", gen_dout, B, T, dIn["gen_lstm"], nrec, before_step=gen_before, slots=slots)
            inf_chain = self._chain(plan, "inf_lstm", lambda t: m._rowsrc(DQ[t].data_ptr(), 2 * nv, 0, 2 * nv), B, T, dIn["inf_lstm"], nrec, slots=slots)
            # lockstep3: the prior's step t in the same launches as the generator's step t (which reads the prior's step t + 1: issued one
            # super-step earlier)
            pri_chain = self._chain(plan, "prior_lstm", lambda t: m._rowsrc(_addr(dPZ, t * 2 * nv), (T - 1) * 2 * nv, 0, 2 * nv), B, T,
                                    dIn["prior_lstm"], nrec, slots=slots) if lock3 else iter(())
            for k in range(T):
                if k < T - 1:
                    next(gen_chain)
                    if lock3:
                        next(pri_chain)
                if k >= 1:
                    latent(T - 1 - k)
                    next(inf_chain)
                for j, g in enumerate(slots):
                    m._gemm_group(plan, f"bw.dgrad:gen{T - 2 - k}+inf{T - 1 - k}.{j}", list(g))
                    g.clear()
            for ch in (gen_chain, inf_chain, pri_chain):
                for _ in ch:                                  # (exhausts the generators: they record their stacked buffers on the way out)
                    pass
        for t in (() if lockstep else self._chain(plan, "gen_lstm", gen_dout, B, T, dIn["gen_lstm"], nrec, before_step=gen_before)):
            if not overlap:
                self._rows(plan, f"bw.gen.dx{t}", _addr(DX, t * nz), T * nz, 0, _addr(dIn["gen_lstm"], t * gd), (T - 1) * gd, 0, B, 1, nz, 1)
            if det:
                continue
            if overlap:                                       # the sample's backward feeds the inference chain only: on its lane
                plan.wait(inf_lane, 0)
                plan.lane = inf_lane
            latent(t)
            if not overlap:
                plan.wait(inf_lane, 0)
            plan.lane = inf_lane
            next(inf_chain)
            plan.lane = 0
        for _ in inf_chain:                                   # (exhausts the generator: it records its stacked buffers on the way out)
            pass
        plan.join([1, 2] if overlap else [1])
        if overlap:
            # x_t's prior and generator terms join DX for the weight gradients and the I_0 encoder
            prior_dx()
            self._rows(plan, "bw.gen.dx", DX.data_ptr(), T * nz, nz, dIn["gen_lstm"].data_ptr(), (T - 1) * gd, gd, B, T - 1, nz, 1)

        # ---- weight gradients of the generator's and the inference net's chains (the prior's went out behind its chain) ----
        for net in NETS:
            if net not in wgrads_out:
                net_wgrads(net)
        self._flush(plan)

        # ---- context and x_0 gradients -> the I_0 / I_g encoder outputs; inference inputs -> the trajectory encoder ----
        d_enc_traj = buf("bw.d_enc_traj", (B, T, nz))
        zero(d_enc_traj)
        if not det:
            idm = in_dim["inf_lstm"]
            self._rows(plan, "bw.inf.dx", _addr(d_enc_traj, nz), T * nz, nz, dIn["inf_lstm"].data_ptr(), (T - 1) * idm, idm, B, T - 1, nz, 0)
        if EA is not None:
            # gradient of the encoded action a_t: the last nz input columns of every net at step t -> the action encoder's parameters
            dEA = buf("bw.seq.dEA", (B, T - 1, nz))
            for k, net in enumerate(NETS):
                self._rows(plan, f"bw.{net}.dea", dEA.data_ptr(), (T - 1) * nz, nz, _addr(dIn[net], in_dim[net] - nz), (T - 1) * in_dim[net],
                           in_dim[net], B, T - 1, nz, int(k > 0))
            self._mlp_bwd(plan, "action_encoder", "action_encoder", rec["mlp:action_encoder"], self.bk["action_encoder"], dEA.data_ptr(), nz, [])
        for net in NETS:
            c0 = in_dim[net] - 2 * nz - na_                   # columns of e_0 / e_g in the embedding input
            self._rows(plan, f"bw.{net}.de0", DX.data_ptr(), T * nz, 0, _addr(dIn[net], c0), (T - 1) * in_dim[net], in_dim[net], B, T - 1, nz, 2)
            self._rows(plan, f"bw.{net}.deg", dEG.data_ptr(), nz, 0, _addr(dIn[net], c0 + nz), (T - 1) * in_dim[net], in_dim[net], B, T - 1, nz, 2)
        self._flush(plan)
        self._three_encoder_passes(plan, fplan, lambda: self._encoder_backward(plan, fplan, "traj", d_enc_traj.data_ptr(), nz, 0, 0, {}),
                                   lambda: self._encoder_backward(plan, fplan, "I0", _addr(DX), nz, 1, T * nz, dskip),
                                   lambda: self._encoder_backward(plan, fplan, "Ig", _addr(dEG), nz, 1, nz, {}))
        plan.join(list(range(1, 1 + self.n_side)))
        self._unpad_input_grads(plan)
        plan.outs = dict(DX=DX, dEG=dEG, dQZ=dQZ, dPZ=dPZ, DQ=DQ, dIn=dIn, dE_dec=dE_dec, d_enc_traj=d_enc_traj)
        return plan
