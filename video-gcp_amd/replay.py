"""Running a forward: inputs into the plan's persistent buffers, the plan replayed as a hipGraph or by eager launches over the three
lanes (whichever a one-time comparison per plan finds faster), one op optionally bracketed by HIP events for the roofline measurement
(ReplayMixin, mixed into model.GCPTreeModel)."""
import ctypes as C
import os
from contextlib import contextmanager

import torch

from . import packing as pk
from . import runtime as rt
from .hparams import GCPHParams
from .params import init_params, encoder_layers, encoder_skip_layers, decoder_layers
from .plan_ops import _Plan, _addr, N_LANES


# Replay policy "auto".  One process: graph against eager replay timed once per plan (ReplayMixin._eager_replays_faster).  Under a
# process group no rank times anything inside forward() — 24 extra forwards and 6 device synchronisations on the first call of every
# plan key, and ranks that could decide differently —: every rank replays the hipGraph.  With the trajectory encoder captured first
# (forward_plan.py) the graph is at least as fast as the eager plan at every size measured (c1: host-bound eager plan, graph wins;
# c2: 2.62-2.72 ms graph against 2.68-2.73 ms eager, profiles/r06_replay_order.txt) and costs the host one launch per forward.
REPLAY_RULE_PIXELS = None           # decoded pixels per forward from which the eager plan would be replayed under a process group: never


def choose_replay(decoded_pixels, world, measure):
    """True = replay by eager launches, False = replay the hipGraph.  `measure()` is the one-time comparison; it is NOT called when
    world > 1 (tests/test_dist_cpu.py)."""
    if world > 1:
        return REPLAY_RULE_PIXELS is not None and decoded_pixels >= REPLAY_RULE_PIXELS
    return bool(measure())


class ReplayMixin:

    # ------------------------------------------------------------------------------------------------
    # forward
    def _sync_rng_state(self):
        """{key, offset} of the plan's Philox stream on the device.  The key follows torch's CUDA seed (what the torch draw used:
        `torch.cuda.manual_seed` — per rank in train.py — keeps its meaning) mixed with this model's index in the process, so two
        models of one process draw different noise; seeding torch with ANOTHER value restarts the stream.  Seeding torch again with the
        same value cannot be seen from here: call `reseed()` (train.py does, after it seeds)."""
        seed = int(torch.cuda.initial_seed()) & ((1 << 63) - 1)
        if self._rng_seed != seed:
            self._rng_seed = seed
            self._write_rng_state(self._rng_key(seed), 0)

    def _rng_key(self, seed):
        return (seed + self._rng_stream_id * 0x9E3779B97F4A7C15) & ((1 << 63) - 1)

    def _write_rng_state(self, key, offset):
        self._buf("rng_state", (2,), torch.int64).copy_(torch.tensor([key, offset], dtype=torch.int64), non_blocking=True)

    def reseed(self, seed=None):
        """Restart the in-plan noise stream: from torch's current CUDA seed (default) or from `seed`."""
        seed = (int(torch.cuda.initial_seed()) if seed is None else int(seed)) & ((1 << 63) - 1)
        self._rng_seed = int(torch.cuda.initial_seed()) & ((1 << 63) - 1)
        with torch.cuda.stream(self._stream):
            self._write_rng_state(self._rng_key(seed), 0)

    def rng_state(self):
        """(key, offset) of the in-plan noise stream (a checkpoint keeps it: a resumed run continues the sequence); synchronises."""
        if "rng_state" not in self._bufs or self._rng_seed is None:
            return None
        self._stream.synchronize()
        return tuple(int(v) for v in self._buf("rng_state", (2,), torch.int64).cpu().tolist())

    def set_rng_state(self, state):
        """Continue a checkpointed noise stream.  Checkpoints are written by rank 0 only (train.py), so under a process group the stored
        KEY is rank 0's: every rank keeps its OWN key (derived from its per-rank seed, train.py:100-103) and takes only the offset —
        the ranks' streams stay different after --resume (checkpoint.resumed_rng_state)."""
        if state is None:
            return
        from .checkpoint import resumed_rng_state
        seed = int(torch.cuda.initial_seed()) & ((1 << 63) - 1)
        self._rng_seed = seed                                                  # (the restored stream survives until torch is seeded with another value)
        world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        key, offset = resumed_rng_state(state, self._rng_key(seed), world)
        with torch.cuda.stream(self._stream):
            self._write_rng_state(key, offset)

    # ------------------------------------------------------------------------------------------------
    def forward(self, inputs, phase="train", noise=None):
        """BaseGCPModel.forward (base_gcp.py:140-161).

        inputs: dict with I_0, I_g [B,3,H,W], end_ind int64 [B]; optional traj_seq [B,T,3,H,W], z [B,N,nz_vae]
        (depth-first node order).  `noise` [B,N,nz_vae] (breadth-first node order) replaces the RNG draws of
        Gaussian.sample(); when None it is drawn with torch.randn on the device.
        """
        hp = self._hp
        B = inputs["I_0"].shape[0]
        has_traj = "traj_seq" in inputs and not self._sample_prior
        has_z = "z" in inputs
        if not has_traj and not has_z and not self._sample_prior and not hp.deterministic:
            raise ValueError("posterior path needs traj_seq (or use val_mode() / feed z)")
        # get_end_ind (base_gcp.py:215-229): under val_mode(pred_length=True) the length is drawn from the length predictor whenever
        # its loss is trained (or no end_ind is fed); otherwise the fed end_ind is used
        pred_len = bool(self._has_pred_length and self._use_pred_length and hp.regress_length and
                        (hp.length_pred_weight > 0 or "end_ind" not in inputs))
        if "end_ind" not in inputs and not pred_len:
            raise ValueError("end_ind must be fed unless val_mode(pred_length=True) draws it from the length predictor")
        with_loss = has_traj and phase == "train" and "pad_mask" in inputs
        train_aux = has_traj and phase == "train"
        need_idx = self._has_aux_training and train_aux and ((hp.attach_inv_mdl and not hp.train_inv_mdl_full_seq) or
                                                             (hp.attach_cost_mdl and hp.run_cost_mdl))
        AUX = ("inv_t0", "inv_t1", "cost_start_idx", "cost_end_idx")
        fed_idx = need_idx and all(k in inputs for k in AUX)
        opt = tuple(k for k in ("pad_mask", "traj_seq_states", "w0", "actions") if with_loss and k in inputs)
        if hp.action_conditioned_pred and "actions" not in opt:
            opt += ("actions",)                      # the action-conditioned predictor reads them on every path (sequential.py:45-47)
        if not self._decode and (with_loss or has_traj):
            raise ValueError("decode=False is the planner's prior / given-z path: no ground-truth sequence, no losses")
        # inputs are copied into persistent buffers (one D2D copy; 63 MB for traj_seq at c2 = ~25 us) so that the
        # captured graph — which bakes in device pointers — stays valid whatever tensors the caller passes
        # The copies are enqueued on the MODEL's stream (ordered behind the caller's stream by one event), so the launch that
        # follows needs no second cross-stream hand-over before its first kernel.
        caller = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(caller)
        tin = {}
        # under pred_len the fed end_ind is replaced by the draw (base_gcp.py:219-226): it is not read, and the draw goes to a buffer
        # of its own so that a caller who filled input_buffer('end_ind') in place keeps its ground-truth lengths
        names = ("I_0", "I_g") + (("end_ind",) if ("end_ind" in inputs and not pred_len) else ()) + (("traj_seq",) if has_traj else ()) + \
            (("z",) if has_z else ()) + opt + (AUX if fed_idx else ())
        with torch.cuda.stream(self._stream):
            for k in names:
                t = inputs[k]
                want = torch.int64 if (k == "end_ind" or k in AUX) else torch.float32
                buf = self._buf("in." + k, tuple(t.shape), want)
                if not (t.is_cuda and t.data_ptr() == buf.data_ptr() and t.dtype == want):
                    # (a caller that fills `input_buffer(k, shape)` directly — a loader writing its batch in place — skips the copy)
                    buf.copy_(t, non_blocking=True)
                    if t.is_cuda:
                        t.record_stream(self._stream)
                tin[k] = buf
            if "end_ind" not in tin:
                tin["end_ind"] = self._buf("out.end_ind", (B,), torch.int64)      # written by the length draw inside the plan
            if pred_len:
                # the OneHotCategorical draw of the sequence length (misc.py:49) as one uniform number per sequence
                lu = self._buf("in.len_u", (B,))
                if "len_u" in inputs:
                    lu.copy_(inputs["len_u"], non_blocking=True)
                else:
                    lu.uniform_()
                tin["len_u"] = lu
            # One generator launch per call: the latent noise of Gaussian.sample() and the four numbers per sequence behind the
            # inverse / cost model index draws (InverseModel.sample_offsets / CostModel._general_cost draw with np.random on the host,
            # inverse_mdl.py:84-104, cost_mdl.py:105-107) share one buffer [noise | 4 B numbers]; the index kernel reads the
            # latter as standard-normal draws (u = Phi(n)) and is an op of the plan, i.e. inside the graph.
            n_eps = 0 if has_z else B * self._n_latents() * hp.nz_vae
            draw_idx = need_idx and not fed_idx
            rng = self._buf("rng", (n_eps + (4 * B if draw_idx else 0),)) if (n_eps or draw_idx) else None
            # drawn by the plan itself (a side-lane op of the graph, off the encoder chain) when nothing is fed
            in_plan = bool(self._rng_in_plan and rng is not None and noise is None and os.environ.get("GCPX_TORCH_RNG") is None and
                           not (pred_len and draw_idx))       # (there the index draw sits in front of the fork: keep the torch draw)
            if draw_idx:
                tin["aux_n"] = rng[n_eps:].view(4, B)
                for k in AUX:
                    tin[k] = self._buf("in." + k, (B,), torch.int64)
            if not has_z and n_eps:
                # the draws of Gaussian.sample() live in a persistent buffer as well
                eps = rng[:n_eps].view(B, self._n_latents(), hp.nz_vae)
                if noise is None:
                    if not in_plan:
                        rng.normal_()
                else:
                    if not (noise.is_cuda and noise.data_ptr() == eps.data_ptr()):
                        eps.copy_(noise)
                        if noise.is_cuda:
                            noise.record_stream(self._stream)
                    if draw_idx:
                        tin["aux_n"].normal_()
                tin["eps"] = eps
            elif draw_idx and not in_plan:
                rng.normal_()                        # (z is fed, or the predictor is deterministic: only the index draws)
            if in_plan:
                tin["rng_all"] = rng
                self._sync_rng_state()
        # the plan (and its captured graph) bakes in buffer addresses and sizes: everything that selects buffers is part of the key
        shapes = tuple((k, tuple(tin[k].shape)) for k in sorted(tin))
        key = (B, has_traj, has_z, self._sample_prior, phase, self.training, self.materialize_distr, with_loss, self._decode, pred_len,
               shapes)             # (a plan that draws its own noise has "rng_all" among its inputs: part of `shapes`)
        if key not in self._plans:
            plan = self._build_plan(key, tin)
            plan.keep.append(tin)
            self._plans[key] = (None, plan)
        plan = self._plans[key][1]
        stream = self._stream.cuda_stream
        if self._timed_op is not None:
            self._run_timed(plan, stream)
        elif self.use_graph:
            if plan.graph is None:
                plan.run(self._streams)               # warm-up (sets kernel attributes) outside capture
                plan.graph = self._capture(plan, plan.ops, stream)
                plan.eager = self.use_graph == "auto" and self._choose_eager(plan, stream, B)
            if plan.eager:
                plan.run(self._streams)
            else:
                rt.check(self.lib.gcpx_graph_launch(plan.graph, stream), "graph_launch")
        else:
            plan.run(self._streams)
        caller.wait_stream(self._stream)
        return self._wrap_outputs(plan.outs, tin, phase)

    def _choose_eager(self, plan, stream, B):
        dist = torch.distributed
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        hp = self._hp
        frames = (B if B is not None else hp.batch_size) * hp.n_nodes if self._decode else 0
        return choose_replay(frames * hp.img_sz * hp.img_sz, world, lambda: self._eager_replays_faster(plan, stream))

    def _eager_replays_faster(self, plan, stream, reps=4, trials=3):
        """time `reps` consecutive replays of the plan as a hipGraph and as eager launches — each replay between the same two stream
        hand-overs a forward() call makes (caller -> model stream -> caller), host enqueue included: what a caller's loop pays —,
        `trials` times in turn, and say whether the best eager time beats the best graph time by more than 2 %.  (Timed WITHOUT the
        hand-overs, back-to-back graph launches pipeline into each other and look 0.2 ms faster per forward than they are inside a
        loop of forward() calls.)  One-time cost per plan: 2 x trials x (reps + 1) forwards."""
        import time
        caller = torch.cuda.current_stream(self.device)

        def one(eager):
            self._stream.wait_stream(caller)
            if eager:
                plan.run(self._streams)
            else:
                rt.check(self.lib.gcpx_graph_launch(plan.graph, stream), "graph_launch")
            caller.wait_stream(self._stream)

        def timed(eager):
            one(eager)
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(reps):
                one(eager)
            torch.cuda.synchronize(self.device)
            return time.perf_counter() - t0
        tg = te = float("inf")
        for _ in range(trials):
            tg = min(tg, timed(False))
            te = min(te, timed(True))
        plan.tuned = (tg / reps, te / reps)
        return te < 0.98 * tg

    def replay_info(self):
        """How the plan of the latest forward() is replayed and what the one-time comparison measured (bench.py reports it):
        {"mode": "graph" | "eager", "policy": the GCPX_FORWARD_REPLAY setting, "tuned_ms": {"graph", "eager"} or None}."""
        plan = [v[1] for v in self._plans.values()][-1]
        tuned = plan.tuned
        return {"mode": "eager" if (plan.eager or not self.use_graph) else "graph",
                "policy": {True: "graph", False: "eager"}.get(self.use_graph, "auto"),
                "tuned_ms": None if tuned is None else {"graph": round(1e3 * tuned[0], 4), "eager": round(1e3 * tuned[1], 4)}}

    def force_replay(self, mode):
        """Replay every existing plan as a hipGraph ("graph"), by eager launches ("eager") or as the one-time comparison chose ("auto",
        re-measured on the next call).  Measurement aid (bench.py's also.forward_graph / forward_eager legs)."""
        self.use_graph = {"graph": True, "eager": False, "auto": "auto"}[mode]
        for _, plan in self._plans.values():
            if mode == "auto":
                plan.graph = None
            plan.eager = mode == "eager"

    def _capture(self, plan, ops, stream):
        rt.check(self.lib.gcpx_graph_begin(stream), "graph_begin")
        # GCPX_GRAPH_LINEAR=1: every lane captured on the main stream — the graph is one chain of kernel nodes (no cross-branch edges,
        # no lane overlap)
        lanes = [self._streams[0]] * len(self._streams) if os.environ.get("GCPX_GRAPH_LINEAR") == "1" else self._streams
        plan.run(lanes, ops)
        g = C.c_void_p()
        rt.check(self.lib.gcpx_graph_end(stream, C.byref(g)), "graph_end")
        return g

    # ---- one op of the plan bracketed by HIP events on the launch stream (roofline measurement) ----
    def set_timed_op(self, name):
        self._timed_op = name
        self._timed_events = []

    def _run_timed(self, plan, stream):
        names = [op[0] for op in plan.ops]
        i = names.index(self._timed_op)
        e0, e1 = C.c_void_p(), C.c_void_p()
        rt.check(self.lib.gcpx_event_create(C.byref(e0)), "event_create")
        rt.check(self.lib.gcpx_event_create(C.byref(e1)), "event_create")
        if self.use_graph == "auto" and plan.graph is None:
            plan.run(self._streams)
            plan.graph = self._capture(plan, plan.ops, stream)
            plan.eager = self._choose_eager(plan, stream, None)
        if plan.eager or not self.use_graph:
            # the plan as it is replayed (eager launches over the lanes), with an event on the main lane in front of and behind the op
            if plan.timed_ops is None or plan.timed_ops[0] != self._timed_op:
                assert plan.ops[i][3] == 0, "the timed op must be on the main lane"
                plan.timed_ops = (self._timed_op, plan.ops[:i] + [("@mark", None, ("timed", 0), 0), plan.ops[i], ("@mark", None, ("timed", 1), 0)] +
                                  plan.ops[i + 1:])
            plan.run(self._streams, ops=plan.timed_ops[1],
                     on_mark=lambda tag, k: rt.check(self.lib.gcpx_event_record(e1 if k else e0, stream), "event_record") if tag == "timed" else None)
            self._timed_events.append((e0, e1))
            return
        if plan.split is None:
            plan.run(self._streams)
            assert plan.ops[i][3] == 0, "the timed op must be on the main lane"
            plan.split = (self._capture(plan, plan.ops[:i], stream), self._capture(plan, plan.ops[i + 1:], stream))
        rt.check(self.lib.gcpx_graph_launch(plan.split[0], stream), "graph_launch")
        rt.check(self.lib.gcpx_event_record(e0, stream), "event_record")
        name, fn, args, _ = plan.ops[i]
        rt.check(fn(*args, stream), name)
        rt.check(self.lib.gcpx_event_record(e1, stream), "event_record")
        rt.check(self.lib.gcpx_graph_launch(plan.split[1], stream), "graph_launch")
        self._timed_events.append((e0, e1))

    def profile_ops(self, inputs, phase="train", noise=None, repeats=5):
        """Per-op device time of the current plan (eager launches bracketed by events): [(name, microseconds)].
        Tuning aid; not used on the hot path."""
        self.forward(inputs, phase, noise)
        torch.cuda.synchronize()
        plan = [v[1] for v in self._plans.values()][-1]
        res = []
        with torch.cuda.stream(self._stream):
            for name, fn, args, _ in plan.ops:
                if name.startswith("@"):
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                rt.check(fn(*args, self._stream.cuda_stream), name)
                e0.record(self._stream)
                for _ in range(repeats):
                    rt.check(fn(*args, self._stream.cuda_stream), name)
                e1.record(self._stream)
                self._stream.synchronize()
                res.append((name, 1e3 * e0.elapsed_time(e1) / repeats))
        return res

    def timed_op_ms(self):
        """Durations (ms) of the timed op for every forward since set_timed_op(); synchronises."""
        out = []
        for e0, e1 in self._timed_events:
            ms = C.c_float()
            rt.check(self.lib.gcpx_event_elapsed_ms(e0, e1, C.byref(ms)), "event_elapsed")
            out.append(ms.value)
            self.lib.gcpx_event_destroy(e0)
            self.lib.gcpx_event_destroy(e1)
        self._timed_events = []
        return out
