"""The forward plan's container (ops over preallocated buffers on three lanes, replayed eagerly or captured as a hipGraph) and the
helpers that turn one layer of the model into one C-ABI call of the plan: argument structs of gcpx_gemm / gcpx_mlp / gcpx_conv3x3 /
gcpx_bn_finalize over the model's buffers (PlanOpsMixin, mixed into model.GCPTreeModel)."""
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
        self.tuned = None    # (graph, eager) seconds per replay measured by the one-time comparison (replay._eager_replays_faster)
        self.timed_ops = None   # (op name, op list with event marks around it): the eager replay with one op timed (replay._run_timed)
        self.split = None    # (graph before, graph behind) the timed op: the graph replay with one op timed

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

    def record(self, lane):
        """an event at lane's current position, for a LATER `await_event` on another lane (a chain that runs far ahead hands over
        step by step to one that starts much later in plan order)"""
        ev = self._events(1)[0]
        self.ops.append(("@record", None, (lane, ev), 0))
        return ev

    def await_event(self, lane, ev):
        self.ops.append(("@await", None, (lane, ev), 0))

    def mark(self, tag, payload):
        """a host-side callback point in the launch sequence (eager replay only): `run(..., on_mark=f)` calls f(tag, payload)"""
        self.ops.append(("@mark", None, (tag, payload), 0))

    def compact(self, streams, min_len=3, ranges=None):
        """The op list with every run of >= min_len consecutive launches on ONE lane replaced by a captured linear graph (`@graph`):
        lane order, events and marks stay what they are — only the host issues one call per run instead of one per launch.
        ranges: [(first, end)] index ranges of self.ops to look at (None: everywhere)."""
        lib, out, run = self.lib, [], []
        inside = (lambda i: True) if ranges is None else (lambda i: any(a <= i < b for a, b in ranges))

        def flush():
            if len(run) >= min_len:
                st = streams[run[0][3]]
                rt.check(lib.gcpx_graph_begin(st), "graph_begin")
                for name, fn, args, _ in run:
                    rt.check(fn(*args, st), name)
                g = C.c_void_p()
                rt.check(lib.gcpx_graph_end(st, C.byref(g)), "graph_end")
                out.append(("@graph", None, (g, tuple(o[0] for o in run)), run[0][3]))
            else:
                out.extend(run)
            run.clear()
        for i, op in enumerate(self.ops):
            ctl = op[0].startswith("@") or not inside(i)
            if ctl or (run and op[3] != run[0][3]):
                flush()
            (out if ctl else run).append(op)
        flush()
        return out

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
            elif name == "@graph":
                rt.check(lib.gcpx_graph_launch(args[0], streams[lane]), "graph_launch")
            elif name == "@record":
                rt.check(lib.gcpx_event_record(args[1], streams[args[0]]), "record")
            elif name == "@await":
                rt.check(lib.gcpx_stream_wait_event(streams[args[0]], args[1]), "await")
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


class PlanOpsMixin:

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
        gs = self._gsplit.get(wpk.data_ptr())
        # (a trainer's model: only packs the trainer re-splits behind every optimizer step — training.py: _live_gemm_split)
        if gs is not None and self.split_f16 and (self._arena is None or self._gsplit_live):
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
        if a.wpk_split and M >= self._planes_min_rows and N >= 1024 and group is None and not a.stats_partial:
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
