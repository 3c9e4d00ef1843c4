"""One layer's backward as C-ABI calls of the backward plan: weight gradients (gcpx_wgrad*, grouped and deferred to the side lanes), data
gradients (gcpx_gemm on transposed packs), BatchNorm / GroupNorm / Predictor backward (BackwardOpsMixin, mixed into training.GCPTrainStep)."""
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


class _PtrHolder:
    """wraps a raw device address so helper signatures that expect tensors (`.data_ptr()`) can take it"""

    def __init__(self, p):
        self._p = p

    def data_ptr(self):
        return self._p


class BackwardOpsMixin:

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
        # ... but a weight gradient is side-lane work: with ONE workgroup per CU the launches take 1.3-1.8x longer alone (head 760 -> 1016 us)
        # and the c2 step is 0.1-0.2 ms SHORTER (c5: 0.33) — the data-gradient chain on the main lane gets the other half of every CU.  Not
        # for the flat model (its weight gradients run beside chains of 10 us launches: +0.5 ms).  training.py: wgrad_per_cu
        if self.wgrad_per_cu:
            per_cu = self.wgrad_per_cu
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

    def _dgemm(self, plan, tag, srcs, M, N, rpb, wpk, out, ob, orow, batch=None, lstm_bwd=None, group=None):
        """data-gradient GEMM: out = concat(srcs) @ packed(W^T).  lstm_bwd: LstmBwdArgs of the LSTM layer this gradient is the d h of —
        its cell backward then runs in the GEMM's epilogue (gcpx_gemm_args.lstm_bwd) instead of a launch of its own.  group: a list —
        the problem is appended to it instead of being launched (`_gemm_group` issues the list as one launch)."""
        dev = None
        if lstm_bwd is not None:
            t = torch.frombuffer(bytearray(bytes(lstm_bwd)), dtype=torch.uint8).to(self.m.device)
            plan.keep += [t, lstm_bwd]
            dev = t.data_ptr()
        self.m._gemm(plan, f"bw.dgrad:{tag}", srcs, M, N, rpb, wpk, None, out=out, ob=ob, orow=orow, batch=batch, lstm_bwd=dev, group=group)

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

    def _tree_accum(self, plan, tag, dst, dst_sb, slot_stride, B, n, width, srcs):
        a = rt.TreeAccumArgs()
        for i, (ptr, ld, ol, orr, c0, cg, dcol) in enumerate(srcs):
            s = a.src[i]
            s.ptr, s.ld, s.off_left, s.off_right, s.off_ctx0, s.off_ctxg, s.dst_col = ptr, ld, ol, orr, c0, cg, dcol
        a.dst, a.dst_sb, a.slot_stride, a.nsrc, a.B, a.n, a.width = dst.data_ptr(), dst_sb, slot_stride, len(srcs), B, n, width
        plan.keep.append(a)
        plan.add(f"bw.accum:{tag}", self.m.lib.gcpx_tree_accum, C.byref(a))
