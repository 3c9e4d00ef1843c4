"""Construction of the backward plan from the records the forward plan left behind: losses -> decoder -> tree levels (leaves first) ->
encoder passes, with the marks that tell the step which slice of the flat gradient is final (BackwardPlanMixin, mixed into
training.GCPTrainStep).  Mirrors what autograd does for /root/reference/gcp/prediction/train.py:155-163."""
import ctypes as C
import os
import re

import torch

from . import packing as pk
from . import runtime as rt
from .plan_ops import _Plan, _addr, N_LANES
from .params import decoder_layers


def _c16(n):
    return (n + 15) // 16 * 16


class BackwardPlanMixin:

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
            # the mixture mean to the head's raw parameters; the matching weights carry no gradient to the images (adaptive.py:50 detaches the cost)
            Dd = hp.input_nc * S * S
            dImg = buf("bw.dImg", (B, N, hp.input_nc, S, S))
            plan.add("bw.avg_nll", lib.gcpx_averaging_nll_bwd, o["match_dist_df"].data_ptr(), tin["pad_mask"].data_ptr(),
                     o["images_df"].data_ptr(), tin["traj_seq"].data_ptr(), o["cdist_sum"].data_ptr(),
                     m.sd["decoder.log_sigma"].data_ptr(), C.c_float(hp.dense_img_rec_weight / (B * div)), B, N, T, Dd, dImg.data_ptr(),
                     self.g("decoder.log_sigma"))
            if hp.learn_matching_temp:
                # adaptive.py:19-21, :51: only the COST is detached; the division by the learned temperature is not, so the criterion
                # reaches `temp` through the matching weights (forward-mode sweep along the forward pass's accumulators)
                self._side(plan, "bw.dtw_dtemp", lib.gcpx_soft_dtw_dtemp, o["cdist_sum"].data_ptr(), C.c_float(float(Dd)),
                           m.sd["tree_module.tree_modules.0.binding.temp"].data_ptr(), tin["end_ind"].data_ptr(),
                           o["dtw_acc"].data_ptr(), tin["pad_mask"].data_ptr(), m.sd["decoder.log_sigma"].data_ptr(),
                           C.c_float(hp.dense_img_rec_weight / (B * div)), B, N, T,
                           buf("bw.dtw.tangent", (2 * B, N, T), torch.float64).data_ptr(),
                           buf("bw.dtw.partial", (B,), torch.float64).data_ptr(), self.g("tree_module.tree_modules.0.binding.temp"))
            dMD = buf("bw.dMD", (B * N, S, S, pitch))
            # (lean: the mean depends on the leading 8 x mixtures slots only — the gradient's other slots are neither written here nor read
            # by the head's data / weight gradient launches, backward_stages._lean_head)
            plan.add("bw.dlm_mean", lib.gcpx_dlm_mean_bwd, o["distr_df_kernel_order"].data_ptr(), dImg.data_ptr(), dMD.data_ptr(),
                     buf("bw.dMD.colsum", (B * N, pitch)).data_ptr(), B * N, S * S, pitch, hp.n_mixtures,
                     8 * hp.n_mixtures if self._lean_head(rec) else pitch)
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
