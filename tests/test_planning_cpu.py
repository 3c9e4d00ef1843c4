"""CPU: CEM bookkeeping (elite selection + refit) against a plain numpy restatement of
gcp/planning/cem/cem_planner.py:124-135 and gcp/planning/cem/sampler.py:44-46 (KAT-8), and the sharded planner's
collective path with a stub simulator under gloo (world_size 2)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kat8_elites_and_refit():
    from video_gcp_amd.planning import select_elites, FlatCEMSampler
    rng = np.random.RandomState(0)
    scores = rng.rand(64)
    scores[[3, 17]] = scores[5]                       # ties
    samples = rng.randn(64, 7, 4)
    n_elite = int(64 * 0.1)
    want_idx = scores.argsort(kind="stable")[:n_elite]              # cem_planner.py:129-130
    got = select_elites(torch.tensor(scores), n_elite).numpy()
    assert np.array_equal(got, want_idx)
    s = FlatCEMSampler(float("inf"), 7, 4, 0.3, device="cpu")
    s.fit(torch.tensor(samples[want_idx]))
    assert np.allclose(s.mean.numpy(), samples[want_idx].mean(0)) and np.allclose(s.std.numpy(), samples[want_idx].std(0))
    # same seed -> same population on every rank
    a = FlatCEMSampler(2.0, 7, 4, 0.3, device="cpu", seed=5).sample(16)
    b = FlatCEMSampler(2.0, 7, 4, 0.3, device="cpu", seed=5).sample(16)
    assert torch.equal(a, b) and float(a.abs().max()) <= 2.0
    # a sharded population is the concatenation of its shards' own streams: a rank that draws only shard k gets rows k of it
    full = FlatCEMSampler(2.0, 7, 4, 0.3, device="cpu", seed=5, n_shards=4).sample(16)
    part = FlatCEMSampler(2.0, 7, 4, 0.3, device="cpu", seed=5, n_shards=4).sample_shard(16, 2)
    assert full.shape[0] == 16 and torch.equal(part, full[8:12]) and not torch.equal(full[:4], full[4:8])


def test_env2planner():
    from video_gcp_amd.planning import env2planner
    img = np.random.RandomState(1).randint(0, 256, size=(1, 8, 8, 3)).astype(np.uint8)
    out = env2planner(img)
    assert out.shape == (1, 3, 8, 8)
    assert np.allclose(out.numpy(), img.transpose(0, 3, 1, 2) / 255.0 * 2 - 1, atol=1e-6)


class _StubSim:
    """cost-relevant outputs of a rollout: latents = f(sample) so that the score depends only on the candidate"""
    def rollout_device(self, state, goal, samples, rollout_len):
        from video_gcp_amd.model import Outputs
        n = samples.shape[0]
        lat = samples[:, :rollout_len].clone()
        return Outputs(latents=lat, lengths=torch.full((n,), rollout_len, dtype=torch.int32), e_goal=torch.zeros(n, lat.shape[-1]))

    def rollout(self, state, goal, samples, rollout_len, prune=False):
        from video_gcp_amd.model import Outputs
        s = torch.as_tensor(samples)
        return Outputs(predictions=[s[0].numpy()], actions=None, latents=[s[0, :rollout_len].numpy()])


class _StubCost:
    def sequence_cost_device(self, lat, lengths, goal=None):
        return (lat ** 2).sum((1, 2))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from video_gcp_amd import dist as D
    from video_gcp_amd.planning import CEMPlanner, FlatCEMSampler
    if world > 1:
        D.init_from_env("gloo")
    # the population = two independently seeded shards: a 2-rank group draws one each, the single process draws both
    sampler = FlatCEMSampler(float("inf"), 7, 4, 1.0, device="cpu", seed=3, n_shards=2)
    planner = CEMPlanner(_StubSim(), _StubCost(), sampler, n_iters=3, batch_size=32, elite_frac=0.25, max_seq_len=7)
    _, _, lat, best = planner(None, None)
    q.put((rank, best, [float(l.elite_scores[0]) for l in planner.logs], float(np.abs(lat).sum())))
    if world > 1:
        torch.distributed.destroy_process_group()


def _run(world):
    """(as tests/test_dist_cpu.py:_run_ranks: a run whose rendezvous itself fails is repeated once on a fresh port)"""
    import queue as _q
    last = None
    for _ in range(2):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = []
        try:
            for _ in procs:
                res.append(q.get(timeout=180))
        except _q.Empty:
            last = "a rank delivered nothing within 180 s"
        for p in procs:
            p.join(60)
            if p.exitcode is None:
                p.kill()
                p.join(10)
        if len(res) == world and all(p.exitcode == 0 for p in procs):
            return sorted(res)
        last = last or f"exit codes {[p.exitcode for p in procs]}"
    raise AssertionError(f"the {world}-rank run failed twice: {last}")


def test_sharded_cem_matches_single_rank():
    """every rank draws and rolls out ONLY its shard of the population (sampler shard = rank), costs all-gathered, the elite rows —
    drawn on whichever rank — assembled by one all-reduce: same elites, same refit, same plan as one process that draws both shards"""
    one = _run(1)[0]
    two = _run(2)
    # both ranks agree with each other and with the unsharded planner (same population, same elites, same refit)
    assert two[0][1:] == two[1][1:] == one[1:]
    assert one[2][-1] <= one[2][0]            # CEM on a convex cost improves its best elite


def test_hand_written_costs_and_pddm_sampler():
    """CostFcn subclasses (cost_fcn.py:10-77) and PDDMSampler (sampler.py:52-71) against their definitions written out with numpy:
    final-step weight, dense vs last-step cost, path length with the goal appended, step count, image L2 on the image columns of an
    (image ++ latent) rollout; correlated noise n_i = BETA u_i + (1 - BETA) n_{i-1} and the exp(-score)-weighted mean refit."""
    from video_gcp_amd import planning as P
    rng = np.random.RandomState(0)
    outs = [rng.randn(l, 5) for l in (3, 6, 1)]
    goal = rng.randn(5)
    per = [np.linalg.norm(o - goal[None], axis=-1) for o in outs]
    assert np.allclose(P.EuclideanDistance(True, 2.0)(outs, goal), [p[:-1].sum() + 2 * p[-1] for p in per])
    assert np.allclose(P.EuclideanDistance(False, 2.0)(outs, goal), [2 * p[-1] for p in per])
    pl = [np.linalg.norm(np.concatenate([o[1:], goal[None]]) - o, axis=-1).sum() for o in outs]
    assert np.allclose(P.EuclideanPathLength(True)(outs, goal), pl)
    assert np.allclose(P.StepPathLength(False, 3.0)(outs, goal), [9.0, 18.0, 3.0]) and np.allclose(P.StepPathLength(True)(outs, goal), [3, 6, 1])
    S, nz = 4, P.L2ImageCost.LATENT_SIZE
    rolls = [rng.rand(l, 3 * S * S + nz) for l in (2, 5)]
    goal_img = rng.rand(1, S, S, 3)
    want = [np.sqrt((((r[:, :3 * S * S].reshape(-1, 3, S, S) - (goal_img.transpose(0, 3, 1, 2) * 2 - 1)) ** 2).sum((1, 2, 3))))[-1] for r in rolls]
    assert np.allclose(P.L2ImageCost(False)(rolls, goal_img), want)
    s = P.PDDMSampler(float("inf"), 6, 3, 0.5, device="cpu", seed=1)
    x = s.sample(4)
    u = 0.5 * torch.randn(4, 6, 3, generator=torch.Generator().manual_seed(1))
    n_i, cor = torch.zeros(4, 3), []
    for i in range(6):
        n_i = s.BETA * u[:, i] + (1 - s.BETA) * n_i
        cor.append(n_i)
    assert torch.allclose(x, torch.stack(cor, 1), atol=1e-6)
    scores = torch.tensor([0.3, 1.2, 0.1, 2.0])
    std0 = s.std.clone()
    s.fit(x, scores)
    w = np.exp(-scores.numpy())
    assert np.allclose(s.mean.numpy(), (x.numpy() * w[:, None, None]).sum(0) / w.sum(), atol=1e-6) and torch.equal(s.std, std0)


def test_cost_scores_keep_the_rollout_dtype():
    """cost_fcn.py:15-22: float32 rollouts give float32 step costs, the final-step weight and the sum stay in float32 and so do the
    scores (the reference never widens them) — equal, bit for bit, to that formula written out."""
    from video_gcp_amd.planning import EuclideanDistance, EuclideanPathLength
    rng = np.random.RandomState(4)
    rollouts = [rng.randn(n, 6).astype(np.float32) for n in (5, 9, 2)]
    goal = rng.randn(6).astype(np.float32)
    for cls, dense, w in [(EuclideanDistance, True, 3.0), (EuclideanDistance, False, 0.5), (EuclideanPathLength, True, 1.0)]:
        fn = cls(dense, w)
        got = fn(rollouts, goal)
        assert got.dtype == np.float32, cls.__name__
        want = []
        for r in rollouts:
            c = np.array(fn.per_step(r, fn._prepare_goal(goal)))
            assert c.dtype == np.float32
            c[-1] *= w
            want.append(np.sum(c) if dense else c[-1])
        assert np.array_equal(got, np.array(want))


def test_costs_and_samplers_match_the_executed_reference():
    """tests/golden/ref_costs_samplers.npz = outputs of the reference's own CostFcn subclasses (cost_fcn.py:8-77) and of its
    FlatCEMSampler / PDDMSampler (sampler.py:33-71) executed in the build container (make_ref_costs_goldens.py).  The host contract of
    the cost classes is float64 numpy like the reference's: equal to the last bit.  The samplers here are float32 torch: the population
    they build from the SAME standard-normal numbers the reference's np.random call consumed, and their refits, to float32 rounding."""
    from video_gcp_amd import planning as P
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_costs_samplers.npz"))
    rolls = [g[f"state_roll{i}"] for i in range(len(g["state_lens"]))]
    for k, case in enumerate(g["state_cases"]):
        name, dense, w = str(case).split("|")
        got = getattr(P, name)(bool(int(dense)), float(w))([r.copy() for r in rolls], g["state_goal"])
        assert np.array_equal(np.asarray(got, dtype=np.float64), g[f"state_cost{k}"]), case
    img_rolls = [g[f"img_roll{i}"] for i in range(len(g["img_lens"]))]
    for k, (dense, w) in enumerate(g["img_cases"]):
        got = P.L2ImageCost(bool(dense), float(w))([r.copy() for r in img_rolls], g["img_goal"])
        assert np.array_equal(np.asarray(got, dtype=np.float64), g[f"img_cost{k}"]), (dense, w)
    n, steps, ad = (int(v) for v in g["sampler_shape"])
    for tag, cls in (("flat", P.FlatCEMSampler), ("pddm", P.PDDMSampler)):
        for ct, clip in (("inf", float("inf")), ("clip", 0.8)):
            s = cls(clip, steps, ad, 0.7, device="cpu", seed=0)
            s.mean, s.std = torch.tensor(g[f"{tag}_{ct}_mean"], dtype=torch.float32), torch.tensor(g[f"{tag}_{ct}_std"], dtype=torch.float32)
            got = s.from_unit_noise(torch.tensor(g[f"{tag}_{ct}_unit_noise"], dtype=torch.float32))
            assert np.allclose(got.numpy(), g[f"{tag}_{ct}_samples"], rtol=0, atol=2e-6), (tag, ct)
            if ct == "clip":
                assert float(got.abs().max()) == np.float32(0.8) and float(np.abs(g[f"{tag}_{ct}_samples"]).max()) == 0.8      # something was clipped
        s = cls(float("inf"), steps, ad, 0.7, device="cpu", seed=0)
        s.fit(torch.tensor(g[f"{tag}_fit_data"], dtype=torch.float32), torch.tensor(g[f"{tag}_fit_scores"], dtype=torch.float32))
        assert np.allclose(s.mean.numpy(), g[f"{tag}_fit_mean"], atol=2e-6), tag
        assert np.allclose(s.std.numpy(), g[f"{tag}_fit_std"], atol=2e-6), tag          # (PDDM never refits its std: equals the initial one)
        if tag == "pddm":
            assert np.array_equal(g["pddm_fit_std"], g["pddm_fit_std_before"])


def test_simulator_input_conventions_match_the_executed_reference():
    """GCPImageSimulator._env2planner on uint8 / unit-range / 5-d inputs and ActCondGCPImageSimulator._postprocess_inputs
    (cem_simulator.py:72-104) executed in the build container: env images -> NCHW in [-1, 1]; the action-conditioned simulator's
    candidates are the planner's samples themselves ([n, T-1, n_actions]) with a pad mask of ones"""
    from video_gcp_amd.planning import env2planner
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_costs_samplers.npz"))
    for tag in ("u8", "unit", "five"):
        got = env2planner(g[f"env_{tag}_in"])
        assert got.dtype == torch.float32 and np.array_equal(got.numpy(), g[f"env_{tag}_out"]), tag
    assert np.array_equal(g["act_actions"], g["act_in"]) and np.array_equal(g["act_pad_mask"], np.ones(g["act_in"].shape[:2], np.float32))
    assert np.array_equal(env2planner(np.repeat(g["env_u8_in"], 3, 0)).numpy(), g["act_I_0"])


# ---- the CEM loop itself against the EXECUTED reference (tests/golden/make_ref_cem_loop_goldens.py -> ref_cem_loop.npz):
# cem_planner.py:55-135 (flat) and :166-218 (hierarchical) on a stub simulator, the draws replayed from the recorded unit numbers ----
def _cem_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_cem_loop.npz"))


class _ReplaySampler:
    """FlatCEMSampler whose Gaussian numbers are the ones the reference's np.random stream produced (one array per sample() call)"""

    def __new__(cls, eps_list, *args, **kw):
        from video_gcp_amd.planning import FlatCEMSampler

        class R(FlatCEMSampler):
            def sample(self, n_samples):
                e = torch.as_tensor(self._eps.pop(0))
                assert e.shape[0] == n_samples
                return self.from_unit_noise(e.to(self.mean.dtype))
        r = R(*args, **kw)
        r._eps = list(eps_list)
        return r


class _FlatStubSim:
    """tests/golden/planner_stubs.flat_stub_rollout behind the planner's simulator interface"""

    def __init__(self, dtype):
        self.dtype = dtype

    def _roll(self, samples, rollout_len):
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        from planner_stubs import flat_stub_rollout
        return flat_stub_rollout(np.asarray(samples), rollout_len, self.dtype)

    def rollout_device(self, state, goal, samples, rollout_len):
        from video_gcp_amd.model import Outputs
        preds, _ = self._roll(samples.numpy(), rollout_len)
        n, D = len(preds), preds[0].shape[1]
        pad = np.zeros((n, rollout_len, D), self.dtype)
        for i, p in enumerate(preds):
            pad[i, :len(p)] = p
        return Outputs(latents=torch.as_tensor(pad), lengths=torch.tensor([len(p) for p in preds], dtype=torch.int32), e_goal=None)

    def rollout(self, state, goal, samples, rollout_len, prune=False):
        from video_gcp_amd.model import Outputs
        preds, lats = self._roll(samples, rollout_len)
        return Outputs(predictions=preds, actions=[p[1:] - p[:-1] for p in preds], latents=lats)


class _HostCostOnPadded:
    """the package's host-side hand-written cost (cost_fcn.py contract) applied to the rows of a padded rollout"""

    def __init__(self, cost, goal):
        self.cost, self.goal = cost, goal

    def sequence_cost_device(self, lat, lengths, goal=None):
        rolls = [lat[i, :int(l)].numpy() for i, l in enumerate(lengths)]
        return torch.as_tensor(np.asarray(self.cost(rolls, self.goal), dtype=np.float64))


def test_flat_cem_loop_matches_executed_reference():
    from video_gcp_amd.planning import CEMPlanner, EuclideanPathLength
    g = _cem_golden()
    steps, ad = (int(v) for v in g["flat_shape"])
    for ci, (batch, efrac, mrb, nbytes, clip, seed) in enumerate(g["flat_cases"]):
        batch = int(batch)
        dt = np.float64 if nbytes == 8 else np.float32
        goal = g[f"flat{ci}_goal"]
        sampler = _ReplaySampler([g[f"flat{ci}_it{it}_eps"] for it in range(3)], float(clip), steps, ad, 0.6, device="cpu", dtype=torch.float64)
        cost = _HostCostOnPadded(EuclideanPathLength(True, 2.0), goal)
        planner = CEMPlanner(_FlatStubSim(dt), cost, sampler, n_iters=3, batch_size=batch, elite_frac=float(efrac), max_seq_len=steps)
        # the loop, iteration by iteration, so that every intermediate is compared
        sampler.init()
        if mrb < batch:
            # The reference's chunked rollout (cem_planner.py:114-121, batch_size > max_rollout_bs; no shipped conf gets there: 10 / 5
            # candidates against 100) PREPENDS every chunk (`_join_dicts(sim_output, output)` = d1 + d2), so its score vector is in
            # reversed chunk order while `samples[elite_idxs]` indexes the draw order: elites are refit from the wrong candidates.  This
            # planner scores the population in draw order; the fixture pins what the reference does, as a permutation of the same scores.
            _, _, scores = planner.iterate(None, goal)
            mrb = int(mrb)
            chunks = [scores.numpy()[i * mrb:(i + 1) * mrb] for i in range(batch // mrb)]
            np.testing.assert_allclose(np.concatenate(chunks[::-1]), g[f"flat{ci}_it0_scores"], rtol=1e-13)
            assert list(g[f"flat{ci}_rollout_calls"][:batch // mrb]) == [mrb] * (batch // mrb)
            continue
        for it in range(3):
            best, best_scores, scores = planner.iterate(None, goal)
            want_scores = g[f"flat{ci}_it{it}_scores"]
            assert scores.shape[0] == want_scores.shape[0] == batch
            np.testing.assert_allclose(scores.numpy(), want_scores, rtol=1e-13, atol=0)
            order = torch.argsort(scores, stable=True)[:len(g[f"flat{ci}_it{it}_elite_idx"])].numpy()
            assert np.array_equal(order, g[f"flat{ci}_it{it}_elite_idx"]), (ci, it)
            np.testing.assert_allclose(best_scores.numpy(), g[f"flat{ci}_it{it}_elite_scores"], rtol=1e-13)
            np.testing.assert_allclose(best.numpy(), g[f"flat{ci}_it{it}_elite_samples"], rtol=1e-13, atol=1e-15)
            np.testing.assert_allclose(sampler.mean.numpy(), g[f"flat{ci}_it{it}_mean"], rtol=1e-13, atol=1e-15)
            np.testing.assert_allclose(sampler.std.numpy(), g[f"flat{ci}_it{it}_std"], rtol=1e-12, atol=1e-15)
        # and the whole call: plan = rollout of the best candidate, score = its cost (cem_planner.py:81-100)
        sampler._eps = [g[f"flat{ci}_it{it}_eps"] for it in range(3)]
        pred, actions, latents, score = planner(None, goal)
        np.testing.assert_allclose(pred, g[f"flat{ci}_plan_pred"], rtol=1e-6 if dt is np.float32 else 1e-13, atol=1e-7 if dt is np.float32 else 1e-15)
        np.testing.assert_allclose(actions, g[f"flat{ci}_plan_actions"], rtol=1e-5 if dt is np.float32 else 1e-12, atol=1e-6 if dt is np.float32 else 1e-14)
        np.testing.assert_allclose(latents, g[f"flat{ci}_plan_latents"], rtol=1e-6 if dt is np.float32 else 1e-13, atol=1e-7 if dt is np.float32 else 1e-15)
        assert abs(score - float(g[f"flat{ci}_plan_score"][0])) <= 1e-12 * abs(score)


def test_hierarchical_cem_call_matches_executed_reference():
    """HierarchicalCEMPlanner.__call__ (cem_planner.py:166-218): the host data flow (device_resident=False) on the reference's np.random
    stream, draw for draw: elite rollouts and scores of every round and the returned plan, bit for bit"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from planner_stubs import StubCost, stub_rollouts, RES
    from video_gcp_amd.planning import HierarchicalCEMPlanner
    from video_gcp_amd.model import Outputs
    g = _cem_golden()

    class TreeStubSim:
        def rollout(self, state, goal, samples, max_seq_len):
            r = stub_rollouts(np.asarray(samples))
            n_img = 3 * RES * RES
            return Outputs(predictions=r, actions=[x[1:, :2] - x[:-1, :2] for x in r], latents=[x[:, n_img:].copy() for x in r])

    for ci, row in enumerate(g["hier_cases"]):
        depth, n_ll, ld, seed, nr = (int(v) for v in row[:5])
        rates = [int(v) for v in row[5:5 + nr]]
        goal = g[f"hier{ci}_goal"]
        planner = HierarchicalCEMPlanner(TreeStubSim(), StubCost(), depth, rates, n_ll_samples=n_ll, action_dim=ld, max_seq_len=2 ** depth - 1,
                                         device_resident=False)
        np.random.seed(seed + 100)
        pred, actions, latents, score = planner(None, goal)
        for it in range(len(rates) + 1):
            assert np.array_equal(np.asarray(planner.logs[it].elite_rollouts[0]), g[f"hier{ci}_it{it}_elite_rollout"]), (ci, it)
            assert np.array_equal(np.asarray(planner.logs[it].elite_scores, dtype=np.float64).reshape(-1), g[f"hier{ci}_it{it}_elite_score"]), (ci, it)
        assert np.array_equal(pred, g[f"hier{ci}_plan_pred"]) and np.array_equal(actions, g[f"hier{ci}_plan_actions"])
        assert np.array_equal(latents, g[f"hier{ci}_plan_latents"])
        assert score == float(g[f"hier{ci}_plan_score"][0])
        assert bool(planner.fully_optimized) == bool(g[f"hier{ci}_fully"][0])
