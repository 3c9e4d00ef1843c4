"""Construction of the forward plan: BaseGCPModel.forward (/root/reference/gcp/prediction/models/base_gcp.py:140-304) ->
run_encoder -> get_end_ind -> TreeModel.predict_sequence (tree.py:42-77, tree_module.py:67-114) -> TreeDenseRec -> bindings, heads and
losses, recorded once per (shapes, phase) as C-ABI calls over the model's buffers on three lanes (ForwardPlanMixin, mixed into
model.GCPTreeModel)."""
import ctypes as C
import os
from contextlib import contextmanager

import torch

from . import packing as pk
from . import runtime as rt
from .hparams import GCPHParams
from .params import init_params, encoder_layers, encoder_skip_layers, decoder_layers
from .plan_ops import _Plan, _addr, N_LANES


class ForwardPlanMixin:

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
            # A conv is linear in its input channels and the skip channels are the SAME for the rpb nodes of a sequence: the blocks
            # whose split-f16 kernel takes an addend (32 / 64 output channels) convolve the skip half once per sequence — F / rpb frames,
            # on a side lane as soon as the I_0 encoder is done — and the per-node launch walks the node's own channels only: half of
            # the block's MFMAs (pyramid-1 at c2: 128 -> 32 channels at 16 x 16, 38 of 455 GFLOP).  GCPX_NO_SKIP_HOIST=1: one launch
            hoist = (skip_idx >= 0 and rpb > 1 and F % rpb == 0 and self.split_f16 and cout in (32, 64) and c_prev % 32 == 0 and
                     c_skip % 32 == 0 and f"dec.{name}" in self.pk_split and "fold" not in self.pk_split[f"dec.{name}"] and
                     os.environ.get("GCPX_NO_SKIP_HOIST") is None)
            # ... and the 16 + 16 -> 16 channel block (additional_conv_layer) through the 16-channel row-folded packs
            # (opt-in, GCPX_SKIP_HOIST16=1: measured SLOWER than the one-launch 32-channel row-folded kernel, 282 against 234 us at c2 — with
            # 144 MFMAs per item the wavefront's latency chain (patch loads -> LDS -> interpolation -> LDS -> fragments, the addend's loads
            # in the epilogue) is no longer covered: 57 % of the wave cycles parked in s_waitcnt, profiles/r06_skip_hoist.txt)
            hoist16 = (skip_idx >= 0 and rpb > 1 and F % rpb == 0 and self.split_f16 and cout == 16 and c_prev == 16 and c_skip == 16 and
                       f"dec.{name}.f16a" in self.pk_split and res % 4 == 0 and os.environ.get("GCPX_SKIP_HOIST16") == "1")
            addend = None
            if hoist16:
                addend = self._buf(f"dec.{name}.skip", (F // rpb, 2 * res, 2 * res, cout))
                t, C_, ssc, ssh, sact = skips[skip_idx]
                zb = self._buf(f"dec.{name}.zero_bias", (cpad,), zero=True)
                a_s = self._conv_args([(t.data_ptr(), C_, 1, ssc, ssh, sact)], F // rpb, res, res, 2 * res, 2 * res, cout, cout,
                                      P[f"dec.{name}.w"], zb, addend, upsample=1)
                self._set_split(a_s, f"dec.{name}.f16b")
                plan.keep.append(a_s)
                plan.add(f"dec.{name}.skip", lib.gcpx_conv3x3, C.byref(a_s))
                srcs_main = [prev]
            elif hoist:
                addend = self._buf(f"dec.{name}.skip", (F // rpb, 2 * res, 2 * res, cout))
                t, C_, ssc, ssh, sact = skips[skip_idx]
                zb = self._buf(f"dec.{name}.zero_bias", (cpad,), zero=True)
                a_s = self._conv_args([(t.data_ptr(), C_, 1, ssc, ssh, sact)], F // rpb, res, res, 2 * res, 2 * res, cout, cout,
                                      P[f"dec.{name}.w"], zb, addend, upsample=1)
                self._set_split(a_s, f"dec.{name}")
                # the skip channels' k-steps follow the previous block's in the pack: [chunk][tap][CT][2][64] x 16 B, 32 channels per chunk
                a_s.wpk_split = a_s.wpk_split + (c_prev // 32) * 9 * (cpad // 16) * 2048
                plan.keep.append(a_s)
                # (16 frames: a latency-bound 18 us launch, in line.  On lane 1 beside the previous block it measured SLOWER — 2.62-2.66 ms
                # against 2.60-2.61 in line, three rounds: a fork / join pair in the middle of the decoder chain costs more than it hides)
                plan.add(f"dec.{name}.skip", lib.gcpx_conv3x3, C.byref(a_s))
                srcs_main = [prev]
            else:
                srcs_main = srcs
            a = self._conv_args(srcs_main, F, res, res, 2 * res, 2 * res, cout, cout, P[f"dec.{name}.w"], P[f"dec.{name}.b"],
                                o, upsample=1, stats=(o if self.training else None))     # placeholder pointer for the query
            if hoist or hoist16:
                a.addend, a.addend_frame_div = addend.data_ptr(), rpb
            self._set_split(a, f"dec.{name}.f16a" if hoist16 else f"dec.{name}")
            Gl = lib.gcpx_conv3x3_grid(C.byref(a))
            assert Gl > 0, rt.lib().gcpx_last_error()
            st = self._buf(f"dec.st.{name}", (Gl, 2, cpad)) if self.training else None
            a.stats_partial = st.data_ptr() if st is not None else None
            plan.keep.append(a)
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
        # The trajectory encoder (B T frames: ~0.3 ms of the forward) is ENQUEUED FIRST: eager replay issues the ops in plan order at
        # ~4 us per launch, and two lanes can share a hardware queue — behind the ~25 small launches of the two image encoders and the
        # index bookkeeping its first kernel started 0.25 ms into the forward (profiles/r05c_fwd_trace.txt: t = 271 us, queue q1 behind
        # the I_g pass).  GCPX_ENC_ORDER=small_first restores the old order (A/B aid).
        traj_first = os.environ.get("GCPX_ENC_ORDER") != "small_first"
        def plan_traj_encoder():
            nonlocal enc_traj, inf_enc
            plan.lane = 0
            if has_traj:
                enc_traj = self._buf("enc_traj", (B * T, nz))
                self._plan_encoder(plan, "traj", tin["traj_seq"].data_ptr(), B * T, enc_traj.data_ptr(), T * nz, nz, T)
                if hp.seq_enc == "none":                       # build_temporal_inf_encoder -> Identity (base_gcp.py:131-132)
                    inf_enc = enc_traj
                else:
                    inf_enc = self._buf("inf_enc_seq", (B * T, nz))
                    self._plan_seq_encoder(plan, "seq", "inf_encoder", enc_traj, inf_enc, B)
                if attentive:
                    # attention keys: second temporal encoder + per-frame Linear (base_gcp.py:122-123, :200); then the key /
                    # value projections of every level's attention in one batched launch each
                    dk = hp.nz_attn_key
                    n_mod = L if hp.untied_layers else 1
                    if hp.seq_enc == "none":
                        kenc = enc_traj
                    else:
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

        if traj_first:
            plan_traj_encoder()
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
        if not traj_first:
            plan_traj_encoder()
        plan.lane = 0
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
                    et = self._plan_attention(plan, l, W, el(), er(), M, n, B, plan.rec["attn_kv"]["Kp"], plan.rec["attn_kv"]["Vp"], tin)
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
            if (l + 1 < L and merge_with_predictors and hp.tree_lstm == "split_linear" and B * 2 ** (l + 1) >= self._merge_side_rows):
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
            if hp.run_state_regressor:
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
                outs["dtw_acc"] = self._buf("dtw.acc", (2 * B, N, T), torch.float64)
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
