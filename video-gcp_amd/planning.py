"""CEM planning over the gcp_tree predictor: host-side mirror of the reference's planning stack with the rollout,
the learned cost and the elite selection kept on the device, and the candidate population sharded across ranks.

Reference (paths relative to /root/reference):
  GCPSimulator.rollout / GCPImageSimulator        gcp/planning/cem/cem_simulator.py:14-96
  CEMPlanner.__call__ / _rollout / _get_best_rollouts  gcp/planning/cem/cem_planner.py:55-135
  FlatCEMSampler / SimpleTreeCEMSampler           gcp/planning/cem/sampler.py:33-76
  LearnedCostEstimate                             gcp/planning/cem/cost_fcn.py:79-101
  TestTimeCostModel.forward                       gcp/prediction/models/auxilliary_models/cost_mdl.py:138-145

Deviations, on purpose:
  * D2 (cem_planner.py:115-122 drops the remainder of `batch_size // max_rollout_bs`): every candidate is evaluated.
  * D3: the reference's rollout runs under `val_mode()` = pred_length=True (cem_simulator.py:29), so the length of every candidate
    is a DRAW from the length predictor (base_gcp.py:219-226).  `GCPImageSimulator(pred_length=True)` (the default) does the same,
    with the categorical draw fed as one uniform number per candidate so that sharded ranks agree; `pred_length=False` pins every
    rollout to `rollout_len` (the fixed-work setting bench.py and the configs[3] tests use).
  * The reference moves every rollout to numpy (~2 GB per 512-candidate call, cem_simulator.py:68-70).  `rollout()` keeps
    that contract; the planner itself uses `rollout_device()` and only the final plan crosses to the host.
  * Multi-GPU: candidates are sharded over ranks; every rank draws the SAME population from a shared
    seed, rolls out its own slice, and one all-gather of the per-candidate costs lets all ranks pick identical elites.
"""
import numpy as np
import torch

from . import dist as D
from . import runtime as rt
from .model import Outputs


def env2planner(img):
    """GCPImageSimulator._env2planner (cem_simulator.py:87-96): env image(s) -> NCHW in [-1, 1]."""
    img = torch.as_tensor(img, dtype=torch.float32)
    if img.max() > 1.0:
        img = img / 255.0
    if img.dim() == 5:
        img = img[0]
    if img.dim() == 4:
        img = img.permute(0, 3, 1, 2)
    return img * 2 - 1.0


class GCPImageSimulator:
    """Simulator interface of the planner: candidate latents -> model rollouts."""

    supports_latent_only = True      # rollout_device(..., decode=False) skips the image decoder

    def __init__(self, model, append_latent=True, pred_length=True):
        self._model = model
        self._append_latent = append_latent
        self.pred_length = pred_length

    def _bcast(self, img, n):
        """the (single) start / goal image is converted once and broadcast on the device (cem_simulator.py:18 repeats it on the host:
        512 x 64 x 64 x 3 numpy copies + two 25 MB pageable uploads were 42 ms of a 60 ms scoring rollout)"""
        t = env2planner(np.asarray(img) if not torch.is_tensor(img) else img).to(self._model.device)
        assert t.shape[0] in (1, n), "one image for all candidates, or one per candidate"
        return t.expand(n, *t.shape[1:]) if t.shape[0] == 1 else t

    def rollout_device(self, state, goal_state, samples, rollout_len, decode=True, len_u=None):
        """Device-resident rollout.  state/goal_state: env images [1,H,W,3]; samples [n, N, nz_vae] (depth-first node
        order, tree.py:38).  Returns padded device tensors + lengths.  decode=False: latents only (images = None) — what the
        CEM scoring loop needs; the decoder is 3/4 of the rollout's FLOPs."""
        m = self._model
        n = samples.shape[0]
        I0, Ig = self._bcast(state, n), self._bcast(goal_state, n)
        z = torch.as_tensor(samples, dtype=torch.float32, device=m.device)
        inp = dict(I_0=I0, I_g=Ig, z=z,
                   end_ind=torch.full((n,), rollout_len - 1, dtype=torch.long, device=m.device))
        if self.pred_length and len_u is not None:
            inp["len_u"] = torch.as_tensor(len_u, dtype=torch.float32, device=m.device)
        with m.val_mode(pred_length=self.pred_length, decode=decode):
            out = m(inp, "train")                    # cem_simulator.py:29-31 (phase defaults to 'train')
        raw = out.raw
        return Outputs(images=raw.get("pruned_padded"), latents=raw["model_enc_seq_padded"], lengths=raw["seq_len"],
                       actions=raw.get("actions_padded"), states=raw.get("regressed_state_padded"),
                       e_goal=raw["E"][:, -1], out=out)

    def predictions_device(self, r):
        """GCPSimulator._get_state_rollouts (cem_simulator.py:48-61) on the padded device rollout: [n, T, 3 H W (+ nz_enc)] — the flat
        image of every step, with the latent appended when the simulator was built with append_latent"""
        assert r.images is not None, "hand-written costs read the decoded rollouts: the planner must decode its candidates"
        n, T = r.images.shape[:2]
        img = r.images.reshape(n, T, -1)
        return torch.cat((img, r.latents), dim=-1) if self._append_latent else img

    def rollout(self, state, goal_state, samples, rollout_len, prune=False, len_u=None):
        """Reference contract (cem_simulator.py:14-43): lists of numpy arrays, one per candidate."""
        r = self.rollout_device(state, goal_state, samples, rollout_len, len_u=len_u)
        lens = r.lengths.tolist()
        n = len(lens)
        img = r.images.reshape(n, r.images.shape[1], -1)
        preds = []
        for i, l in enumerate(lens):
            p = img[i, :l]
            if self._append_latent and not prune:
                p = torch.cat((p, r.latents[i, :l]), dim=-1)      # cem_simulator.py:54-59: image ++ latent
            elif prune:
                p = r.images[i, :l]
            preds.append(p.cpu().numpy())
        cap = lambda t, off=0: [t[i, :max(l - off, 0)].cpu().numpy() for i, l in enumerate(lens)]
        return Outputs(predictions=preds, actions=cap(r.actions, 1) if r.actions is not None else None,
                       states=cap(r.states) if r.states is not None else None, latents=cap(r.latents))


class ActCondGCPImageSimulator(GCPImageSimulator):
    """cem_simulator.py:99-104: the planner's samples are ACTION sequences (they arrive in the `z` slot), rolled out by an
    action-conditioned flat predictor (`action_conditioned_pred`, sequential.py:24-25,45-49; base_configs/vmpc.py) —
    `GCPSequentialModel`.  Selected by the policy's `act_cond` flag (planner_policy.py:197,231).  Such a model never draws its length
    (base_gcp.py:224-226): every rollout has `rollout_len` frames, frame 0 being the start image (sequential.py:57)."""

    supports_latent_only = False     # the flat predictor's plan always decodes

    def __init__(self, model, append_latent=True, pred_length=False):
        assert model._hp.action_conditioned_pred, "the action-conditioned simulator needs an action-conditioned predictor"
        super().__init__(model, append_latent=append_latent, pred_length=False)

    def rollout_device(self, state, goal_state, samples, rollout_len, decode=True, len_u=None):
        """samples [n, max_seq_len - 1, n_actions]: action t leads from frame t to frame t + 1"""
        m = self._model
        hp = m._hp
        a = torch.as_tensor(samples, dtype=torch.float32, device=m.device)
        if a.dim() == 5:
            a = a[..., 0, 0]                                      # (cem_simulator.py:102: the image simulator appended two unit axes)
        n = a.shape[0]
        assert tuple(a.shape[1:]) == (hp.max_seq_len - 1, hp.n_actions), "one action per predicted frame"
        inp = dict(I_0=self._bcast(state, n), I_g=self._bcast(goal_state, n), actions=a,
                   end_ind=torch.full((n,), rollout_len - 1, dtype=torch.long, device=m.device))
        with m.val_mode(pred_length=False):
            out = m(inp, "train")
        raw = out.raw
        return Outputs(images=raw["images"], latents=raw["model_enc_seq_padded"], lengths=raw["seq_len"],
                       actions=raw.get("actions_padded"), states=raw.get("regressed_state_padded"), e_goal=raw["e_g"], out=out)


class LearnedCostEstimate:
    """Learned pairwise cost on latents (cost_fcn.py:79-101) backed by the model's `cost_mdl.cost_pred` Predictor."""

    def __init__(self, model):
        self._model = model
        self.lib = model.lib

    @property
    def input_dim(self):
        return self._model._hp.nz_enc

    def pair_cost(self, enc1, enc2):
        """cost of moving from enc1 to enc2, rows [R, nz] -> [R, 1]."""
        return self._model.predictor_rows("cost_mdl", torch.as_tensor(enc1), torch.as_tensor(enc2))

    def __call__(self, start_enc, goal_enc):
        if isinstance(start_enc, np.ndarray):                       # single start/goal pairs
            return self.pair_cost(start_enc, goal_enc).cpu().numpy()
        if isinstance(start_enc, list):                             # summed cost per sequence
            T = max(s.shape[0] for s in start_enc)
            n, nz = len(start_enc), start_enc[0].shape[-1]
            lat = torch.zeros(n, T, nz)
            for i, s in enumerate(start_enc):
                lat[i, :s.shape[0]] = torch.as_tensor(s)
            goal = torch.stack([torch.as_tensor(g).reshape(-1, nz)[0] for g in goal_enc])
            lens = torch.tensor([s.shape[0] for s in start_enc], dtype=torch.int32)
            dev = self._model.device
            return self.sequence_cost_device(lat.to(dev), lens.to(dev), goal.to(dev)).cpu().numpy()
        raise ValueError("Dimensionality of input to learned cost function not supported!")

    def sequence_cost_device(self, lat, lengths, goal=None):
        """sum_t cost(lat[i,t], next) over the first len_i steps, next = lat[i,t+1] or the goal latent after the last
        step (cost_fcn.py:91-94).  lat [n,T,nz], lengths int32 [n], goal [n,nz] -> [n]."""
        n, T, nz = lat.shape
        dev = lat.device
        lat = lat.contiguous()
        nxt = torch.empty_like(lat)
        stream = torch.cuda.current_stream(dev).cuda_stream
        g = goal.contiguous() if goal is not None else None
        rt.check(self.lib.gcpx_seq_pairs(lat.data_ptr(), lengths.data_ptr(), g.data_ptr() if g is not None else None,
                                         nxt.data_ptr(), n, T, nz, stream), "seq_pairs")
        per_step = self._model.predictor_rows("cost_mdl", lat.reshape(n * T, nz), nxt.reshape(n * T, nz))
        out = torch.empty(n, device=dev)
        rt.check(self.lib.gcpx_masked_row_sum(per_step.data_ptr(), lengths.data_ptr(), out.data_ptr(), n, T, stream),
                 "masked_row_sum")
        return out


class CostFcn:
    """Hand-written CEM costs (cost_fcn.py:10-77).  Host contract of the reference: `cost(cem_outputs, goal)` with `cem_outputs` a list of
    per-candidate arrays [len_i, D] -> one score per candidate; a subclass says what ONE rollout costs per step (`per_step`), this class
    turns the step costs into the score: the last step weighted by `final_step_weight`, then either everything summed (`dense_cost`) or
    the last step alone.  `rollout_cost_device` scores a padded device rollout ([n, T, D] + lengths) with one HIP launch
    (gcpx_rollout_cost, same four kinds) for the device-resident planner."""
    KIND = None

    def __init__(self, dense_cost, final_step_weight=1.0, *unused_args):
        self._dense_cost = dense_cost
        self._final_step_weight = final_step_weight

    def per_step(self, rollout, goal):
        """[len] step costs of one rollout [len, D]"""
        raise NotImplementedError

    def _prepare_goal(self, goal):
        return goal

    def __call__(self, cem_outputs, goal):
        # (cost_fcn.py:15-22: the step costs keep the rollout's dtype — float32 for model rollouts — through the final-step weight and
        # the sum, and so do the scores: near-tied candidates rank as in the reference)
        goal = self._prepare_goal(goal)
        scores = []
        for rollout in self._prepare_rollouts(cem_outputs):
            steps = np.array(self.per_step(rollout, goal))
            steps[-1] *= self._final_step_weight
            scores.append(np.sum(steps) if self._dense_cost else steps[-1])
        return np.array(scores)

    def _prepare_rollouts(self, cem_outputs):
        return cem_outputs

    def _device_goal(self, goal, D, device):
        return torch.as_tensor(np.asarray(goal, dtype=np.float32).reshape(-1)[:D].copy(), device=device)

    def rollout_cost_device(self, x, lengths, goal):
        """x [n, T, ld] device rollouts (predictions), lengths int32 [n], goal: what the host contract takes -> costs [n]"""
        from . import runtime as rt_
        n, T, ld = x.shape
        D = self._device_columns(ld)
        g = self._device_goal(goal, D, x.device)
        out = torch.empty(n, device=x.device)
        x = x.contiguous()
        rt_.check(rt_.load_library().gcpx_rollout_cost(x.data_ptr(), ld, lengths.data_ptr(), g.data_ptr(), 0, out.data_ptr(), n, T, D, self.KIND,
                                                       int(bool(self._dense_cost)), float(self._final_step_weight),
                                                       torch.cuda.current_stream(x.device).cuda_stream), "rollout_cost")
        return out

    def _device_columns(self, ld):
        return ld


class ImageCost:
    """cost_fcn.py:26-39: a rollout row is an image (3 x S x S, flattened) followed by `input_dim` latent columns"""

    def _split_state_rollout(self, rollouts):
        images, latents = [], []
        for r in rollouts:
            r = np.asarray(r)
            assert r.ndim == 2
            n_pix = r.shape[1] - self.input_dim
            side = int(np.sqrt(n_pix / 3))                        # three colour planes of side x side pixels
            images.append(r[:, :n_pix].reshape(r.shape[0], 3, side, side))
            latents.append(r[:, n_pix:])
        return Outputs(image_rollout=images, latent_rollout=latents)


def _row_norms(a):
    return np.sqrt(np.square(a).sum(axis=-1))


class EuclideanDistance(CostFcn):
    """distance of every step to the goal (cost_fcn.py:42-46)"""
    KIND = 0

    def per_step(self, rollout, goal):
        return _row_norms(rollout - np.asarray(goal).reshape(1, -1))


class EuclideanPathLength(CostFcn):
    """length of the path through the steps and on to the goal (cost_fcn.py:49-54); only meaningful summed over the steps"""
    KIND = 1

    def per_step(self, rollout, goal):
        assert self._dense_cost, "a path length is the sum over its steps: dense_cost must be set"
        successor = np.vstack([rollout[1:], np.asarray(goal).reshape(1, -1)])
        return _row_norms(successor - rollout)


class StepPathLength(CostFcn):
    """the number of steps, charged to the last one (cost_fcn.py:57-62)"""
    KIND = 2

    def per_step(self, rollout, goal):
        steps = np.zeros(len(rollout))
        steps[-1] = float(len(rollout))
        return steps

    def _device_goal(self, goal, D, device):
        return torch.zeros(1, device=device)


class L2ImageCost(CostFcn, ImageCost):
    """pixel-space distance of every predicted image to the goal image (cost_fcn.py:65-77); goal_raw: env image [1, H, W, 3] in [0, 1]"""
    KIND = 3
    LATENT_SIZE = 128

    @property
    def input_dim(self):
        return self.LATENT_SIZE

    @staticmethod
    def _goal_planes(goal_raw, dtype=None):
        g = np.asarray(goal_raw, dtype=dtype)
        return np.moveaxis(g, 3, 1) * 2 - 1.0                     # HWC in [0, 1] -> CHW in [-1, 1]

    def _prepare_goal(self, goal_raw):
        return self._goal_planes(goal_raw)

    def _prepare_rollouts(self, cem_outputs):
        return self._split_state_rollout(cem_outputs).image_rollout

    def per_step(self, images, goal):
        return np.sqrt(np.square(images - goal).sum(axis=(1, 2, 3)))

    def _device_columns(self, ld):
        return ld - self.input_dim

    def _device_goal(self, goal_raw, D, device):
        return torch.as_tensor(self._goal_planes(goal_raw, np.float32).reshape(-1)[:D].copy(), device=device)


class FlatCEMSampler:
    """Gaussian sampler over [n_steps, dim] (sampler.py:33-48), on the device, seedable so that all ranks of a
    sharded planner draw the same population."""

    def __init__(self, clip_val, n_steps, action_dim, initial_std, device="cuda", seed=0, n_shards=None, dtype=torch.float32):
        """dtype: of the distribution and its draws (the reference's sampler is float64 numpy; float32 on the device by default).
        n_shards: the population is the concatenation of n_shards independently seeded sub-streams (seed, shard); default = the
        number of ranks, so that under a process group every rank draws ONLY its own 1/world of the candidates ("each GPU
        samples its slice (seed = base + rank)").  A single process asked for n_shards = k draws the same population a k-rank
        group does, which is how the sharded planner is checked against the unsharded one."""
        self._clip_val, self._n_steps, self._action_dim, self._initial_std = clip_val, n_steps, action_dim, initial_std
        self.device = torch.device(device)
        self._seed, self._n_shards, self._dtype = seed, n_shards, dtype
        self._gen = torch.Generator(device=self.device)             # rank-shared stream (length draws, one-shard populations)
        self._gen.manual_seed(seed)
        self._shard_gens = {}
        self.init()

    def init(self):
        self.mean = torch.zeros(self._n_steps, self._action_dim, device=self.device, dtype=self._dtype)
        self.std = self._initial_std * torch.ones(self._n_steps, self._action_dim, device=self.device, dtype=self._dtype)

    def n_shards(self):
        if self._n_shards is not None:
            return self._n_shards
        return D.dist.get_world_size() if D.dist.is_initialized() else 1

    def _draw(self, n, gen):
        return self.from_unit_noise(torch.randn(n, self._n_steps, self._action_dim, device=self.device, generator=gen, dtype=self._dtype))

    def from_unit_noise(self, eps):
        """the population that standard-normal numbers eps [n, n_steps, action_dim] stand for (sampler.py:40-42: np.random.normal(loc=mean,
        scale=std) = mean + std * eps, clipped) — the deterministic part of `sample`, pinned to the reference's on its own draws"""
        raw = self.mean[None] + self.std[None] * eps
        return raw.clamp(-self._clip_val, self._clip_val) if np.isfinite(self._clip_val) else raw

    def sample_shard(self, n_samples, shard):
        """rows [shard * n / n_shards, (shard + 1) * n / n_shards) of the population of n_samples candidates"""
        ns = self.n_shards()
        assert n_samples % ns == 0, "candidate population must divide evenly over the shards"
        if ns == 1:
            return self._draw(n_samples, self._gen)
        g = self._shard_gens.get(shard)
        if g is None:
            g = self._shard_gens[shard] = torch.Generator(device=self.device)
            g.manual_seed(self._seed * 1000003 + 7919 * (shard + 1))
        return self._draw(n_samples // ns, g)

    def sample(self, n_samples):
        """the whole population (every shard, in order)"""
        ns = self.n_shards()
        if ns == 1:
            return self._draw(n_samples, self._gen)
        return torch.cat([self.sample_shard(n_samples, s_) for s_ in range(ns)], 0)

    def sample_uniform(self, n_samples):
        """one uniform number per candidate from the sampler's own (rank-shared) stream: the length draw of a rollout"""
        return torch.rand(n_samples, device=self.device, generator=self._gen)

    def fit(self, data, scores=None):
        self.mean = data.mean(0)                                    # np.mean / np.std(axis=0), sampler.py:44-46
        self.std = data.std(0, unbiased=False)

    def get_dists(self):
        return Outputs(mean=self.mean, std=self.std)


class PDDMSampler(FlatCEMSampler):
    """Correlated noise + path-integral refit (sampler.py:52-71): n_i = BETA u_i + (1 - BETA) n_{i-1} along the steps, the mean refit
    as the exp(-GAMMA score)-weighted average of the candidates (scores: lower is better); the std is never refit."""
    BETA = 0.5      # noise correlation factor
    GAMMA = 1.0     # reward weighting factor

    def from_unit_noise(self, eps):
        noise = self.std[None] * eps
        # the recurrence unrolled: n_i = BETA sum_{k <= i} (1 - BETA)^(i - k) u_k — one lower-triangular mixing of the step axis
        i = torch.arange(self._n_steps, device=self.device)
        w = self.BETA * (1.0 - self.BETA) ** (i[:, None] - i[None]).clamp(min=0).to(torch.float32) * (i[:, None] >= i[None])
        raw = torch.einsum("ik,nkd->nid", w, noise) + self.mean[None]
        return raw.clamp(-self._clip_val, self._clip_val) if np.isfinite(self._clip_val) else raw

    def fit(self, actions, scores):
        wgt = torch.exp(-self.GAMMA * torch.as_tensor(scores, dtype=actions.dtype, device=actions.device))
        self.mean = (actions * wgt[:, None, None]).sum(0) / wgt.sum()


class SimpleTreeCEMSampler(FlatCEMSampler):
    """All 2^L - 1 tree latents optimised at once (sampler.py:68-76)."""

    def __init__(self, clip_val, n_steps, action_dim, initial_std, n_level_hierarchy, **kw):
        super().__init__(clip_val, 2 ** n_level_hierarchy - 1, action_dim, initial_std, **kw)


def select_elites(scores, n_elite):
    """cem_planner.py:129-130: indices of the n_elite lowest scores (stable, like numpy argsort on ties)."""
    return torch.argsort(scores, stable=True)[:n_elite]


class CEMPlanner:
    """Flat CEM over tree latents with device-resident rollouts and sharded candidates (cem_planner.py:55-135)."""

    def __init__(self, simulator, cost, sampler, n_iters=3, batch_size=512, elite_frac=0.1, max_seq_len=80,
                 goal_in_cost=True, decode_candidates=False):
        # decode_candidates: also decode the images of every candidate while scoring (what the reference's simulator does,
        # cem_simulator.py:29-59).  The learned cost reads latents only, so by default the images are decoded once, for the
        # plan that is returned; scores, elites and the returned plan are the same either way.
        self.decode_candidates = decode_candidates or not hasattr(cost, "sequence_cost_device")     # hand-written costs read the images
        self._sim, self._cost, self._sampler = simulator, cost, sampler
        self.n_iters, self.batch_size, self.elite_frac, self.max_seq_len = n_iters, batch_size, elite_frac, max_seq_len
        self.goal_in_cost = goal_in_cost
        self.logs = []

    def _shard(self, n):
        world = D.dist.get_world_size() if D.dist.is_initialized() else 1
        rank = D.dist.get_rank() if D.dist.is_initialized() else 0
        assert n % world == 0, "candidate population must divide evenly over ranks"
        per = n // world
        return rank * per, per

    def evaluate(self, state, goal_state, samples):
        """costs [n] of all candidates given the WHOLE population: this rank rolls out its slice, one all-gather assembles the vector."""
        lo, per = self._shard(samples.shape[0])
        return self._evaluate_local(state, goal_state, samples[lo:lo + per], lo, samples.shape[0])

    def _evaluate_local(self, state, goal_state, local, lo, n_total):
        per = local.shape[0]
        if getattr(self._sim, "supports_latent_only", False):
            kw = {}
            if getattr(self._sim, "pred_length", False) and hasattr(self._sampler, "sample_uniform"):
                # every rank draws the population's length numbers (n floats) from the shared stream and keeps its slice
                kw["len_u"] = self._sampler.sample_uniform(n_total)[lo:lo + per]
            r = self._sim.rollout_device(state, goal_state, local, self.max_seq_len, decode=self.decode_candidates, **kw)
        else:                                            # any simulator with the reference's interface
            r = self._sim.rollout_device(state, goal_state, local, self.max_seq_len)
        if hasattr(self._cost, "sequence_cost_device"):
            cost = self._cost.sequence_cost_device(r.latents, r.lengths, r.e_goal if self.goal_in_cost else None)
        else:
            # a hand-written CostFcn scores the rollouts' `predictions` against the goal state (cem_planner.py:126): image ++ latent
            # rows of the padded device rollout, one launch
            cost = self._cost.rollout_cost_device(self._sim.predictions_device(r), r.lengths, goal_state)
        return D.all_gather_costs(cost), r

    def iterate(self, state, goal_state):
        """One CEM iteration (cem_planner.py:68-79): draw, roll out, score, pick elites, refit.  Under a process group every rank
        draws and rolls out ONLY its own shard of the population (1 / world of the work), the costs are all-gathered, every rank
        picks the same elites from the full cost vector, and the elite latents — rows of whichever rank drew them — are assembled by
        one all-reduce of the rank's masked contribution (exactly one non-zero contributor per row: x + 0 + ... = x bit-exactly;
        n_elite x N x nz floats, 6.6 MB at 51 x 127 x 256).  Returns (elite samples, elite scores, all scores)."""
        n = self.batch_size
        n_elite = max(int(n * self.elite_frac), 1)
        world = D.dist.get_world_size() if D.dist.is_initialized() else 1
        if world == 1:
            samples = self._sampler.sample(n)
            scores, _ = self._evaluate_local(state, goal_state, samples, 0, n)
            idx = select_elites(scores, n_elite)
            best = samples[idx]
        else:
            assert self._sampler.n_shards() == world, "one shard of the population per rank"
            lo, per = self._shard(n)
            local = self._sampler.sample_shard(n, D.dist.get_rank())
            scores, _ = self._evaluate_local(state, goal_state, local, lo, n)
            idx = select_elites(scores, n_elite)
            mine = (idx >= lo) & (idx < lo + per)
            best = torch.zeros((n_elite,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
            best[mine] = local[idx[mine] - lo]
            D.dist.all_reduce(best)
        best_scores = scores[idx]
        self._sampler.fit(best, best_scores)
        return best, best_scores, scores

    def __call__(self, state, goal_state):
        self._sampler.init()
        n_elite = max(int(self.batch_size * self.elite_frac), 1)
        best_samples = best_scores = None
        self.logs = []
        for _ in range(self.n_iters):
            best_samples, best_scores, scores = self.iterate(state, goal_state)
            self.logs.append(Outputs(elite_scores=best_scores.clone(), mean_score=scores.mean()))
        # final rollout of the best candidate (cem_planner.py:81-96); every rank computes it (tiny batch)
        kw = {}
        if getattr(self._sim, "pred_length", False) and hasattr(self._sampler, "sample_uniform"):
            kw["len_u"] = self._sampler.sample_uniform(1)             # the plan's own length draw, from the rank-shared stream
        final = self._sim.rollout(state, goal_state, best_samples[:1].cpu().numpy(), self.max_seq_len, **kw)
        actions = final.actions[0] if final.actions is not None else None
        return final.predictions[0], actions, final.latents[0], float(best_scores[0])


class ImageHierarchicalTreeCEMSampler:
    """sampler.py:79-143: draws come from the hierarchical latent optimizer, one tree level is fixed per iteration."""

    def __init__(self, clip_val, n_steps, action_dim, initial_std, n_level_hierarchy, sampling_rates_per_layer,
                 subgoal_cost_fcn, ll_cost_fcn, n_ll_samples, device_resident=False, rng=None):
        from .tree_latent_search import ImageHierarchicalTreeLatentOptimizer, DeviceHierarchicalTreeLatentOptimizer
        self._cls = DeviceHierarchicalTreeLatentOptimizer if device_resident else ImageHierarchicalTreeLatentOptimizer
        self.device_resident = device_resident
        self._clip_val, self._action_dim, self._n_levels = clip_val, action_dim, n_level_hierarchy
        self._rates, self._sub_cost, self._ll_cost, self._n_ll = list(sampling_rates_per_layer), subgoal_cost_fcn, ll_cost_fcn, n_ll_samples
        self._rng = rng                  # None: the reference's np.random stream; a np.random.Generator: tree_latent_search._draw
        assert n_level_hierarchy >= len(self._rates)
        self.init()

    def init(self):
        self._optimizer = self._cls(self._action_dim, self._rates.copy(), self._n_levels, self._sub_cost, self._ll_cost, self._n_ll,
                                    rng=self._rng)

    def sample(self, n_samples=None):
        z = self._optimizer.sample()
        if torch.is_tensor(z):                                      # (generator mode on the device)
            return z.clamp(-self._clip_val, self._clip_val) if np.isfinite(self._clip_val) else z
        return np.clip(z, -self._clip_val, self._clip_val)

    def optimize(self, rollouts, goal):
        best_rollout, best_cost = self._optimizer.optimize(rollouts, goal)
        if (best_rollout[-1] != goal[0].transpose(2, 0, 1)).any():      # sampler.py:138-139
            best_rollout = np.concatenate((best_rollout, goal.transpose(0, 3, 1, 2)))
        if not hasattr(best_cost, "__len__"):
            best_cost = [best_cost]
        return [best_rollout], best_cost

    def optimize_device(self, rollout, goal_image=None):
        """device-resident variant (tree_latent_search.DeviceHierarchicalTreeLatentOptimizer): `rollout` is what
        GCPImageSimulator.rollout_device returns; the plan comes back as (row, frame) references + its cost"""
        plan, best_cost = self._optimizer.optimize(rollout)
        return [plan], [best_cost]

    def materialize(self, plan, goal_image=None):
        return self._optimizer.materialize(plan, goal_image)

    def fit(self, *args, **kwargs):
        pass

    append_latent = True

    @property
    def fully_optimized(self):
        return self._optimizer.fully_optimized


class HierarchicalCEMPlanner:
    """HierarchicalImageCEMPlanner (cem_planner.py:166-218 with :55-96): n_iters = len(sampling_rates) + 1 rounds of
    sample -> rollout -> optimise one level.  Populations are tiny (10, 10, 5 in the 25-room config), so ranks run
    replicas; the flat `CEMPlanner` is the sharded one."""

    def __init__(self, simulator, cost, n_level_hierarchy, sampling_rates_per_layer, n_ll_samples=5, action_dim=256,
                 max_seq_len=80, clip_val=float("inf"), device_resident=True, fast_draws=False, seed=0):
        # device_resident: the rollouts of every iteration stay on the device — subgoal pair costs, segment costs and the
        # selections are computed there and only the final plan crosses to the host.  False = the reference's data flow (every
        # rollout, image ++ latent, to numpy: cem_simulator.py:68-70), kept as the checker: both give the same search.
        self._sim, self.max_seq_len = simulator, max_seq_len
        self.n_iters = len(sampling_rates_per_layer) + 1
        self.device_resident = bool(device_resident and hasattr(simulator, "rollout_device") and hasattr(cost, "sequence_cost_device"))
        # fast_draws: the search draws from np.random.default_rng(seed) and only the rows it keeps (tree_latent_search._draw) instead
        # of replaying the reference's np.random call sequence — same distribution of searches, a call is no longer bound by the host's
        # Gaussian generator.  False (default) = the reference's stream, draw for draw.
        self._sampler = ImageHierarchicalTreeCEMSampler(clip_val, max_seq_len, action_dim, 1.0, n_level_hierarchy,
                                                        sampling_rates_per_layer, cost, cost, n_ll_samples,
                                                        device_resident=self.device_resident,
                                                        rng=self._make_rng(fast_draws, seed))
        self.logs = []

    def _make_rng(self, fast_draws, seed):
        """fast_draws: False -> None (the reference's np.random stream); True -> a seeded generator — on the model's device when the
        search is device-resident (the population is then drawn where the rollout reads it), numpy's otherwise"""
        if not fast_draws:
            return None
        dev = getattr(getattr(self._sim, "_model", None), "device", None)
        if self.device_resident and dev is not None and dev.type == "cuda":
            g = torch.Generator(device=dev)
            g.manual_seed(int(seed))
            return g
        return np.random.default_rng(seed)

    def __call__(self, state, goal_state):
        self._sampler.init()
        self.logs = []
        goal = np.asarray(goal_state)
        best_samples = best_scores = None
        for it in range(self.n_iters):
            samples = self._sampler.sample()
            if self.device_resident:
                # the learned cost reads latents only (cost_fcn.py:84-97): the scoring rollouts skip the image decoder
                r = self._sim.rollout_device(state, goal, samples, self.max_seq_len, decode=False)
                best_rollouts, best_scores = self._sampler.optimize_device(r)
            else:
                rollouts = self._sim.rollout(state, goal, samples, self.max_seq_len)
                best_rollouts, best_scores = self._sampler.optimize(rollouts.predictions, goal)     # cem_planner.py:215
            # :216 draws the population again after every round; only the last one (everything optimised: no randomness left in it) is
            # used.  The reference's stream has to consume the others' Gaussians to stay draw-for-draw; the generator mode skips them.
            if self._sampler._rng is None or it == self.n_iters - 1:
                best_samples = self._sampler.sample()
            self.logs.append(Outputs(elite_rollouts=best_rollouts, elite_scores=best_scores))
        final = self._sim.rollout(state, goal, best_samples, self.max_seq_len)
        actions = final.actions[0] if final.actions is not None else None
        return final.predictions[0], actions, final.latents[0], float(np.asarray(best_scores[0]).reshape(-1)[0])

    @property
    def fully_optimized(self):
        return self._sampler.fully_optimized


class ImageCEMPolicy:
    """ImageCEMPolicy / CEMPolicy / PlannerPolicy (planner_policy.py:13-241) without the agent infrastructure: plan with the CEM
    planner when there is no plan, the plan is used up, or the re-planning interval comes round (`act`, :89-113); follow the plan
    open loop through its inverse-model actions or closed loop by re-inferring each action from the current image and the next
    planned latent (`_infer_action`, :222-227: `model.encoder(img)[0][:, :, 0, 0]` -> `model.inv_mdl.run_single`)."""

    def __init__(self, model, planner, replan_interval=1, num_max_replans=10, closed_loop_execution=False):
        self.planner_model, self._cem_planner = model, planner
        self.replan_interval, self.num_max_replans, self.closed_loop_execution = replan_interval, num_max_replans, closed_loop_execution
        self.reset()

    def reset(self):
        self.current_exec_step = None
        self.image_plan = self.action_plan = self.latent_plan = None
        self.plan_cost = None
        self.num_replans = 0

    def act(self, t=None, i_tr=None, state=None, images=None, goal_image=None):
        """images: the frames executed so far [t+1, 1, H, W, 3] or [t+1, H, W, 3] (env format); goal_image [1, H, W, 3]"""
        imgs = np.asarray(images)
        if imgs.ndim == 5:
            imgs = imgs[:, 0]
        if self.image_plan is None or self.image_plan.shape[0] - 1 <= self.current_exec_step \
                or (t % self.replan_interval == 0 and self.num_replans < self.num_max_replans):
            self._plan(imgs[t][None], goal_image, t)
            self.num_replans += 1
        out = Outputs(actions=self.get_action(imgs[t]))
        self.current_exec_step += 1
        return out

    def _plan(self, state, goal, step):
        self.image_plan, self.action_plan, self.latent_plan, self.plan_cost = self._cem_planner(state, goal)
        self.current_exec_step = 0

    def get_action(self, current_image):
        if self.closed_loop_execution:
            return self._infer_action(current_image, self.latent_plan[self.current_exec_step + 1])
        assert self.action_plan is not None          # need to attach inverse model to planner to get actions!
        if self.action_plan.size < 1:
            return 0.05 * np.random.rand(2, )
        return self.action_plan[self.current_exec_step]

    def _infer_action(self, current_img, target_latent):
        m = self.planner_model
        img = env2planner(np.asarray(current_img)[None]).to(m.device)
        enc_img0 = m.encoder(img)[0][:, :, 0, 0]
        tl = torch.as_tensor(np.asarray(target_latent)[None], dtype=torch.float32, device=m.device)
        return m.inv_mdl.run_single(enc_img0, tl)[0].cpu().numpy()
