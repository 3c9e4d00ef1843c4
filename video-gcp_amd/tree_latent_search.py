"""Hierarchical latent search for tree-structured predictors (host side, numpy).

Behavioural mirror of the reference's `ImageHierarchicalTreeLatentOptimizer` / `HierarchicalTreeLatentOptimizer`
(/root/reference/gcp/planning/tree_optimizer.py:7-260): the top `len(sampling_rates)` tree levels are optimised one
level per CEM iteration by scoring each sampled subgoal with a pairwise cost to both parents (:86-132); what remains
below is optimised jointly as dense "segments" (:79-84, 175-180).  Pinned bit-exactly — RNG draw order included — by
tests/golden/ref_tree_optimizer.npz, which was produced by executing the reference class
(tests/golden/make_ref_planner_goldens.py).

Layout contract (tree_optimizer.py:45-68): `sample()` returns [n, 2^depth - 1, latent_dim] with the
node axis in depth-first (in-order) order, i.e. [left subtree | this node | right subtree] — what
`GCPTreeModel.forward` expects for `inputs['z']`.

Written as an explicit search-state tree plus free functions instead of the reference's self-recursive class.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


@dataclass
class _Level:
    """Search state of one position in the hierarchy."""
    depth: int
    n_samples: int
    n_latents: int
    left: Optional[List["_Level"]] = None      # one child search per sample (None at the dense bottom)
    right: Optional[List["_Level"]] = None
    done: bool = False
    best_z: Optional[np.ndarray] = None        # [n_latents, dim] once optimised
    last_draw: Optional[np.ndarray] = None     # [k, n_latents, dim] most recent draw


def _build(rates, depth, n_dense):
    if rates:
        k = rates[0]
        return _Level(depth, k, 1, [_build(rates[1:], depth - 1, n_dense) for _ in range(k)],
                      [_build(rates[1:], depth - 1, n_dense) for _ in range(k)])
    return _Level(depth, n_dense, 2 ** depth - 1)


class HierarchicalTreeLatentOptimizer:
    """Same constructor / methods as the reference class (image variant: states are flattened 3xRxR images)."""

    def __init__(self, latent_dim, sampling_rates, depth, subgoal_cost_fcn, ll_cost_fcn, final_layer_samples,
                 image_states=True, rng=None):
        """rng: None = the reference's draws (module-level np.random, the legacy Gaussian stream, every draw the reference makes —
        bit-exact with its class); a np.random.Generator = the same search on that generator with only the KEPT rows drawn, see
        _draw; a torch.Generator (of the device the rollouts run on) = the same, drawn on that device (_draw_device: `sample()` then
        returns a device tensor)"""
        self._rng = rng
        self._dim = latent_dim
        self._pair_cost = subgoal_cost_fcn
        self._seq_cost = ll_cost_fcn
        self._image_states = image_states
        self._root = _build(list(sampling_rates), depth, final_layer_samples)

    # ------------------------------------------------------------------ sampling
    def sample(self):
        return self._draw(self._root, False)

    def _draw_device(self, lv, below):
        """`_draw` in generator mode with a torch.Generator: the population is built where the model reads it (a dozen small launches
        per call instead of ~300 k host Gaussians and an upload); same tree walk, rows i.i.d. N(0, 1)"""
        import torch
        g = self._rng
        if below:
            assert not lv.done
            return torch.randn(1, 2 ** lv.depth - 1, self._dim, generator=g, device=g.device)
        if lv.done:
            z = lv.best_z[None]
        else:
            z = torch.randn(lv.n_samples, lv.n_latents, self._dim, generator=g, device=g.device)
            lv.last_draw = z
        if lv.left is None:
            return z
        rows = []
        for cl, cr, zi in zip(lv.left, lv.right, z):
            zl, zr = self._draw_device(cl, not lv.done), self._draw_device(cr, not lv.done)
            rows.append(torch.cat([zl, zi[:1].expand(zl.shape[0], 1, self._dim), zr], dim=1))      # depth-first: left | node | right
        return torch.cat(rows)

    def _draw(self, lv, below):
        if self._rng is not None and not isinstance(self._rng, np.random.Generator):
            return self._draw_device(lv, below)
        if below and self._rng is not None:
            # generator mode, below the level being optimised: nothing in this subtree is fixed yet (a level is optimised only after
            # its parent) and one row is kept of every draw — the whole subtree is one block of i.i.d. Gaussians, whatever its layout
            assert not lv.done
            return self._rng.standard_normal(size=(1, 2 ** lv.depth - 1, self._dim), dtype=np.float32)
        if lv.done:
            z = lv.best_z.copy()[None]
        else:
            # the reference draws np.random.normal(loc=zeros, scale=ones, size=...) (tree_optimizer.py:76-78): with array arguments
            # numpy walks the broadcast element by element, 0.0 + 1.0 * gauss — the same legacy Gaussian stream, in the same order, as
            # standard_normal fills in one call (bit-identical draws, pinned by tests/golden/ref_tree_optimizer.npz; 92 of the 97 ms
            # of a planner call were spent in the slow form)
            if self._rng is None:
                z = np.random.standard_normal(size=(lv.n_samples, lv.n_latents, self._dim))
                if below:                  # below the level being optimised only one latent is decoded
                    z = z[:1]
            else:
                # Below the level being optimised the reference draws n_samples rows and keeps the first (tree_optimizer.py:76-82):
                # 3.5 M Gaussians per planner call at rates [10, 10], 5 dense samples, 256-d latents, 26 ns each on the legacy
                # stream = 92 of the 98 ms of a call.  The rows are i.i.d., so drawing only the kept row gives the same
                # distribution of searches; float32 from the generator's ziggurat (what the model reads anyway).
                z = self._rng.standard_normal(size=(1 if below else lv.n_samples, lv.n_latents, self._dim), dtype=np.float32)
            lv.last_draw = z.copy()
        child_below = below or not lv.done
        if lv.left is None:
            return z
        rows = []
        for cl, cr, zi in zip(lv.left, lv.right, z):
            zl, zr = self._draw(cl, child_below), self._draw(cr, child_below)
            assert zl.shape == zr.shape
            mid = np.tile(zi[0], (zl.shape[0], 1, 1))
            rows.append(np.concatenate([zl, mid, zr], axis=1))      # depth-first: left | node | right
        return np.concatenate(rows)

    # ------------------------------------------------------------------ rollout helpers
    def _split(self, rollouts):
        d = self._pair_cost.input_dim
        states, latents = [], []
        for r in rollouts:
            s = r[..., :-d]
            if self._image_states:
                assert s.ndim == 2
                res = int(np.sqrt(s.shape[1] / 3))
                s = s.reshape(s.shape[0], 3, res, res)
            states.append(s)
            latents.append(r[..., -d:])
        return states, latents

    def _segment_inputs(self, rollouts, goal):
        states, latents = self._split(rollouts)
        if self._image_states:
            if goal.ndim > 2:              # raw goal image: score against each rollout's own last latent
                goals = [l[-1:] for l in latents]
            else:
                goals = [self._split([goal[None]])[1][0] for _ in latents]
            return latents, goals
        joined = goal.shape[-1] == rollouts[0].shape[-1]
        return states, (self._split([goal])[0][0] if joined else goal)

    def _best_segment(self, rollouts, goal):
        seqs, goals = self._segment_inputs(rollouts, goal)
        cost = self._seq_cost(seqs, goals)
        k = int(np.argmin(cost))
        return self._split(rollouts)[0][k], cost[k], k

    # ------------------------------------------------------------------ optimisation
    def optimize(self, all_rollouts, goal):
        return self._opt(self._root, all_rollouts, goal)

    def _opt(self, lv, rollouts, goal):
        if lv.left is None:
            best, cost, k = self._best_segment(rollouts, goal)
            lv.best_z, lv.done = lv.last_draw[k], True
            return best, cost
        if not lv.done:
            return self._opt_subgoal(lv, rollouts, goal)
        return self._opt_children(lv, rollouts, goal)

    def _opt_subgoal(self, lv, rollouts, goal):
        states, latents = self._split(rollouts)
        joined = goal.shape[-1] == rollouts[0].shape[-1]          # goal given as a (state ++ latent) vector
        mids = [int(np.floor(s.shape[0] / 2)) for s in states]
        start_s = np.stack([s[0] for s in states])
        start_l = np.stack([l[0] for l in latents])
        sub_s = np.stack([s[m] for s, m in zip(states, mids)])
        sub_l = np.stack([l[m] for l, m in zip(latents, mids)])
        if joined:
            gs, gl = self._split([goal[None]])
            goal_s = np.stack([gs[0][0] for _ in states])
            goal_l = np.stack([gl[0][0] for _ in latents])
        else:
            goal_s = np.stack([goal for _ in states])
            goal_l = np.stack([l[-1] for l in latents])
        total = self._pair_cost(start_l, sub_l) + self._pair_cost(sub_l, goal_l)
        k = int(np.argmin(total))
        lv.best_z = lv.last_draw[k]
        plan = [start_s[k]]
        if (sub_s[k] != plan[-1]).any():                          # identical when the sequence is too short
            plan.append(sub_s[k])
        if not joined:                                            # the final goal is appended exactly once
            g = goal_s[k]
            plan.append(g if g.shape == plan[-1].shape else g[0].transpose(2, 0, 1))
        lv.left, lv.right = lv.left[:1], lv.right[:1]             # keep only the winning branch
        lv.n_samples, lv.done = 1, True
        return np.stack(plan), total[k]

    def _opt_children(self, lv, rollouts, goal):
        results = []
        for cl, cr, group in zip(lv.left, lv.right, np.array_split(rollouts, lv.n_samples)):
            group = [r for r in group]
            short = []
            for i, r in enumerate(group):
                if r.shape[0] < 3:                                # nothing left to expand hierarchically
                    short.append(r)
                    group[i] = np.stack([np.full_like(r[0], np.inf), np.zeros_like(r[0]), np.full_like(r[0], np.inf)])
            cut = [int(np.floor(r.shape[0] / 2)) for r in group]
            subgoal = group[0][cut[0]]
            lr, lc = self._opt(cl, [r[:c] for r, c in zip(group, cut)], subgoal)
            rr, rc = self._opt(cr, [r[c:] for r, c in zip(group, cut)], goal)
            best, cost = np.concatenate([lr, rr]), lc + rc
            if short:
                sb, sc, _ = self._best_segment(short, goal)
                if sc < cost or np.isnan(cost):
                    best, cost = sb, sc
            results.append((best, cost))
        k = int(np.argmin(np.array([c for _, c in results])))
        return results[k]

    @property
    def fully_optimized(self):
        def full(lv):
            if lv.left is None:
                return lv.done
            return lv.done and all(full(c) for c in lv.left) and all(full(c) for c in lv.right)
        return full(self._root)


class ImageHierarchicalTreeLatentOptimizer(HierarchicalTreeLatentOptimizer):
    def __init__(self, *args, **kw):
        kw.setdefault("image_states", True)
        super().__init__(*args, **kw)


class DeviceHierarchicalTreeLatentOptimizer(ImageHierarchicalTreeLatentOptimizer):
    """The same search with the rollouts left on the device.

    The reference moves every rollout (image ++ latent, tens of MB per CEM iteration) to numpy and slices it on the host
    (tree_optimizer.py:86-180 via cem_simulator.py:68-70).  Only LATENTS enter the costs (subgoal pair cost :100-110, dense
    segment cost :70-84), so here a rollout is a (row, first frame, end frame) view into the simulator's device tensors:
    pair costs are one batched Predictor launch per branch (`cost.pair_cost`), segment costs one `cost.sequence_cost_device`
    call, and the host sees the sequence lengths and one argmin index per decision.  Plans are lists of (row, frame) references
    that are gathered into images only when somebody asks (`materialize`).  Draws, selections, costs and `best_z` are identical
    to the numpy class on identical np.random state: it stays in the tests as the checker."""

    RAW = ("raw",)                                   # goal given as the raw environment image: score against own last latent

    def sample(self):
        return super().sample()                      # np.random on the host: the search state is a few hundred KB

    def optimize(self, rollout, goal=None):
        """rollout: Outputs(latents [n, T, nz] device, lengths (device int32 or list), images [n, T, ...] or None)"""
        import torch
        lat = rollout.latents
        # one extra row holds the reference's stand-in for a rollout too short to split, [inf, 0, inf] (tree_optimizer.py:151-156)
        dummy = torch.zeros(1, lat.shape[1], lat.shape[2], device=lat.device)
        dummy[0, 0] = float("inf")
        if lat.shape[1] > 2:
            dummy[0, 2] = float("inf")
        self._lat = torch.cat([lat, dummy])
        self._dummy = lat.shape[0]
        lens = rollout.lengths.tolist() if torch.is_tensor(rollout.lengths) else list(rollout.lengths)
        self._rollout = rollout
        views = [(i, 0, l) for i, l in enumerate(lens)]
        plan, cost = self._dopt(self._root, views, self.RAW)
        return plan, cost

    def materialize(self, plan, goal_image=None):
        """plan references -> [len, 3, H, W] images on the device (the raw goal, when the plan ends in it, is `goal_image`)"""
        import torch
        img = self._rollout.images
        frames = [img[i, t] if i >= 0 else goal_image for i, t in plan]
        return torch.stack(frames)

    # ---- device mirrors of _opt / _opt_subgoal / _opt_children / _best_segment ----
    def _rows(self, refs):
        import torch
        idx = torch.tensor([[i, t] for i, t in refs], device=self._lat.device)
        return self._lat[idx[:, 0], idx[:, 1]]

    def _goal_rows(self, views, goal):
        if goal is self.RAW:
            return self._rows([(i, b - 1) for i, a, b in views])          # l[-1] of every rollout
        return self._rows([goal] * len(views))

    def _dbest_segment(self, views, goal):
        import torch
        n = len(views)
        tmax = max(b - a for _, a, b in views)
        lat = torch.zeros(n, tmax, self._lat.shape[-1], device=self._lat.device)
        for r, (i, a, b) in enumerate(views):
            lat[r, :b - a] = self._lat[i, a:b]
        lens = torch.tensor([b - a for _, a, b in views], dtype=torch.int32, device=self._lat.device)
        cost = self._seq_cost.sequence_cost_device(lat, lens, self._goal_rows(views, goal))
        k = int(torch.argmin(cost))
        i, a, b = views[k]
        return [(i, t) for t in range(a, b)], np.float32(cost[k].item()), k        # float32 like the numpy class: sums of costs stay bit-equal

    def _dopt(self, lv, views, goal):
        if lv.left is None:
            plan, cost, k = self._dbest_segment(views, goal)
            lv.best_z, lv.done = lv.last_draw[k], True
            return plan, cost
        if not lv.done:
            return self._dopt_subgoal(lv, views, goal)
        return self._dopt_children(lv, views, goal)

    def _dopt_subgoal(self, lv, views, goal):
        import torch
        mids = [(b - a) // 2 for _, a, b in views]
        start = self._rows([(i, a) for i, a, b in views])
        sub = self._rows([(i, a + m) for (i, a, b), m in zip(views, mids)])
        gl = self._goal_rows(views, goal)
        n = len(views)
        # both pair costs of all candidates in ONE Predictor launch
        c = self._pair_cost.pair_cost(torch.cat([start, sub]), torch.cat([sub, gl]))
        total = c[:n] + c[n:]
        k = int(torch.argmin(total.reshape(-1)))
        lv.best_z = lv.last_draw[k]
        i, a, b = views[k]
        plan = [(i, a)]
        if mids[k] != 0:
            plan.append((i, a + mids[k]))
        if goal is self.RAW:
            plan.append((-1, -1))                                 # the raw goal image closes the plan (tree_optimizer.py:127-129)
        lv.left, lv.right = lv.left[:1], lv.right[:1]
        lv.n_samples, lv.done = 1, True
        return plan, np.float32(total.reshape(-1)[k].item())

    def _dopt_children(self, lv, views, goal):
        results = []
        for cl, cr, group in zip(lv.left, lv.right, np.array_split(np.arange(len(views)), lv.n_samples)):
            group = [views[j] for j in group]
            short = [v for v in group if v[2] - v[1] < 3]                # nothing left to expand hierarchically
            group = [v if v[2] - v[1] >= 3 else (self._dummy, 0, 3) for v in group]
            cut = [(b - a) // 2 for _, a, b in group]
            i0, a0, _ = group[0]
            subgoal = (i0, a0 + cut[0])
            lr, lc = self._dopt(cl, [(i, a, a + c) for (i, a, b), c in zip(group, cut)], subgoal)
            rr, rc = self._dopt(cr, [(i, a + c, b) for (i, a, b), c in zip(group, cut)], goal)
            best, cost = lr + rr, lc + rc
            if short:
                sb, sc, _ = self._dbest_segment(short, goal)
                if sc < cost or np.isnan(cost):
                    best, cost = sb, sc
            results.append((best, cost))
        k = int(np.argmin(np.array([c for _, c in results])))
        return results[k]
