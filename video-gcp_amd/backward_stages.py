"""Backward of the conv stacks and the temporal / attentive encoders: decoder blocks (bilinear x2 + conv3x3 + BatchNorm + LeakyReLU, the
output head), the conv1d sequence encoder, attention, the strided-conv encoder (BackwardStagesMixin, mixed into training.GCPTrainStep)."""
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


class BackwardStagesMixin:

    # ---- decoder ----
    def _lean_head(self, rec):
        """adaptive model, 10-mixture head at pitch 112 on the split-f16 kernels: the head's backward launches walk the 80 slots the mixture
        mean reads (GCPX_NO_LEAN_MEAN_GRAD: all 112)"""
        m, hp = self.m, self.m._hp
        hs = rec["head_src"]
        return (self.lean_mean_grad and hp.adaptive and m._head_pitch == 112 and hp.n_mixtures == 10 and self.fuse_stage and m.split_f16 and
                self.split_wgrad and hp.ngf == 16 and hs[2] == 1 and hs[5] == rt.ACT_LRELU and hs[3] is not None and hp.img_sz % 32 == 0)

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
        # adaptive: the gradient comes through the mixture MEAN, which the green / blue log-scales (slots 80..99) do not enter: the head's
        # weight and data gradients walk the 80 leading slots of the 112 (5 of 7 channel tiles) and gcpx_dlm_mean_bwd moves only those
        hslots = 8 * hp.n_mixtures if (all_frames and self._lean_head(rec)) else pitch
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
            self._wgrad_conv3(plan, "dec.head", dMD.data_ptr(), pitch, None, R, S, S, ngf, hslots, self.g("decoder.gen_head.conv.weight"),
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
        if hslots != pitch:
            assert bool(a.wpk_split), "the lean head gradient runs on the split-f16 wave kernel"
            a.Cin = hslots                       # the leading slots of every pixel; src[0].C stays the pitch (gcpx_conv3x3: Cin < src[0].C)
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
                        f"bw.dec.{name}.q0" in m.pk_split and f"dec.{name}.wTq0" in self.bk)
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
        if hp.seq_enc == "none":                 # Identity (base_gcp.py:131-132): the gradient is the encoded frames' as it is
            return d_inf
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
