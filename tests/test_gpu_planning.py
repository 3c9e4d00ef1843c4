"""-m gpu: planning path (simulator rollout, learned cost, CEM loop) through the HIP kernels vs the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import make_inputs, assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import video_gcp_amd as V
    from video_gcp_amd.model import GCPTreeModel
    hp = V.config("c1")
    sd = V.init_params(hp, seed=1, randomize_affine=True)
    model = GCPTreeModel(hp, params=sd, device="cuda")
    model.eval()                                         # planner_policy.py:51
    return hp, sd, model


def _env_images(hp, seed):
    rng = np.random.RandomState(seed)
    return (rng.randint(0, 256, size=(1, hp.img_sz, hp.img_sz, 3)).astype(np.uint8),
            rng.randint(0, 256, size=(1, hp.img_sz, hp.img_sz, 3)).astype(np.uint8))


def test_simulator_rollout_matches_oracle(setup):
    from oracle import gcp_model_oracle as O
    from video_gcp_amd.planning import GCPImageSimulator, env2planner
    hp, sd, model = setup
    state, goal = _env_images(hp, 0)
    n, T = 3, hp.max_seq_len
    z = torch.randn(n, hp.n_nodes, hp.nz_vae, generator=torch.Generator().manual_seed(0))
    sim = GCPImageSimulator(model, append_latent=True, pred_length=False)
    got = sim.rollout(state, goal, z.numpy(), T)
    inp = dict(I_0=env2planner(np.repeat(state, n, 0)), I_g=env2planner(np.repeat(goal, n, 0)), z=z,
               end_ind=torch.full((n,), T - 1, dtype=torch.long), start_ind=torch.zeros(n, dtype=torch.long))
    ref = O.forward(sd, hp, inp, sample_prior=True, training_bn=False)      # the rollout runs under val_mode (cem_simulator.py:29)
    assert len(got.predictions) == n
    for i in range(n):
        want = torch.cat((ref["pruned_prediction"][i].reshape(T, -1), ref["model_enc_seq_list"][i]), -1)   # cem_simulator.py:54-59
        assert got.predictions[i].shape == (T, 3 * hp.img_sz ** 2 + hp.nz_enc)
        assert_close(got.predictions[i], want, 5e-5, 1e-4, "predictions")
        assert_close(got.latents[i], ref["model_enc_seq"][i], 5e-5, 1e-4, "latents")
        assert_close(got.states[i], ref["regressed_state"][i], 5e-5, 1e-4, "states")
        assert_close(got.actions[i], ref["actions"][i], 5e-5, 1e-4, "actions")


def test_learned_cost_matches_oracle(setup):
    from oracle import gcp_model_oracle as O
    from video_gcp_amd.planning import LearnedCostEstimate
    hp, sd, model = setup
    cost = LearnedCostEstimate(model)
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(37, hp.nz_enc, generator=g), torch.randn(37, hp.nz_enc, generator=g)
    want = O.predictor(sd, "cost_mdl.cost_pred", hp, a, b)               # TestTimeCostModel.forward, cost_mdl.py:138-145
    assert_close(cost(a.numpy(), b.numpy()), want, 3e-5, 1e-4, "pair cost")
    # list branch (cost_fcn.py:88-95): sum over pairs of cat(seq, goal)
    seqs = [torch.randn(l, hp.nz_enc, generator=g) for l in (5, 2, 9)]
    goals = [torch.randn(1, hp.nz_enc, generator=g) for _ in seqs]
    want = []
    for s, gl in zip(seqs, goals):
        full = torch.cat((s, gl))
        want.append(float(O.predictor(sd, "cost_mdl.cost_pred", hp, full[:-1], full[1:]).sum()))
    got = cost([s.numpy() for s in seqs], [gl.numpy() for gl in goals])
    assert_close(got, np.array(want, dtype=np.float32), 1e-4, 1e-4, "sequence cost")


def test_cem_planner_runs_and_is_deterministic(setup):
    from video_gcp_amd.planning import GCPImageSimulator, LearnedCostEstimate, SimpleTreeCEMSampler, CEMPlanner
    hp, sd, model = setup
    state, goal = _env_images(hp, 2)
    res = []
    for _ in range(2):
        sampler = SimpleTreeCEMSampler(float("inf"), None, hp.nz_vae, 1.0, n_level_hierarchy=hp.hierarchy_levels, device="cuda", seed=11)
        planner = CEMPlanner(GCPImageSimulator(model), LearnedCostEstimate(model), sampler, n_iters=2, batch_size=16,
                             elite_frac=0.25, max_seq_len=hp.max_seq_len)
        plan, actions, latents, score = planner(state, goal)
        res.append((plan, actions, latents, score, [l.elite_scores.cpu().numpy() for l in planner.logs]))
    # the plan's length is a draw from the length predictor (val_mode(pred_length=True), cem_simulator.py:29): 3 .. T frames
    n = res[0][0].shape[0]
    assert 3 <= n <= hp.max_seq_len and res[0][0].shape == (n, 3 * hp.img_sz ** 2 + hp.nz_enc)
    assert res[0][1].shape == (n - 1, hp.n_actions) and res[0][2].shape == (n, hp.nz_enc)
    assert np.array_equal(res[0][0], res[1][0]) and res[0][3] == res[1][3]
    for a, b in zip(res[0][4], res[1][4]):
        assert np.array_equal(a, b)
    assert np.all(np.diff(res[0][4][0]) >= 0)            # elites come out sorted by cost


def test_cem_scoring_without_decoder_is_identical(setup):
    """The learned cost reads latents only (cost_fcn.py:84-97): scoring the candidates without running the image decoder gives
    bit-identical costs, elites and returned plan as decoding every candidate (what the reference's simulator does)."""
    from video_gcp_amd.planning import GCPImageSimulator, LearnedCostEstimate, SimpleTreeCEMSampler, CEMPlanner
    hp, sd, model = setup
    state, goal = _env_images(hp, 5)
    res = []
    for decode in (True, False):
        sampler = SimpleTreeCEMSampler(float("inf"), None, hp.nz_vae, 1.0, n_level_hierarchy=hp.hierarchy_levels, device="cuda", seed=3)
        planner = CEMPlanner(GCPImageSimulator(model), LearnedCostEstimate(model), sampler, n_iters=2, batch_size=16,
                             elite_frac=0.25, max_seq_len=hp.max_seq_len, decode_candidates=decode)
        s0 = sampler.sample(16)
        scores, r = planner.evaluate(state, goal, s0)
        assert (r.images is not None) == decode
        plan, actions, latents, score = planner(state, goal)
        res.append((scores.cpu().numpy(), plan, latents, score, [l.elite_scores.cpu().numpy() for l in planner.logs]))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2]) and res[0][3] == res[1][3]
    for a, b in zip(res[0][4], res[1][4]):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):                       # decode=False is a planner-only path
        inputs, _, _ = make_inputs(hp, seed=0, variant="A")
        with model.val_mode(decode=False):
            model._sample_prior = False
            model({k: v.cuda() for k, v in inputs.items()}, "train")


def test_hierarchical_cem_planner(setup):
    """HierarchicalImageCEMPlanner flow (cem_planner.py:166-218) on the HIP model: one tree level fixed per iteration.  The
    device-resident search (rollouts stay on the GPU, costs and selections computed there) against the reference's data flow
    (every rollout to numpy, the optimizer class that is pinned bit-exactly to the reference's, tests/test_tree_latent_search.py)
    on identical np.random draws: same selections, same optimised latents, same costs, same final plan."""
    from video_gcp_amd.planning import GCPImageSimulator, LearnedCostEstimate, HierarchicalCEMPlanner
    hp, sd, model = setup
    state, goal = _env_images(hp, 3)
    res = {}
    for dev in (False, True):
        np.random.seed(0)
        planner = HierarchicalCEMPlanner(GCPImageSimulator(model, pred_length=False), LearnedCostEstimate(model), hp.hierarchy_levels, [3, 2],
                                         n_ll_samples=2, action_dim=hp.nz_vae, max_seq_len=hp.max_seq_len, device_resident=dev)
        assert planner.device_resident == dev
        plan, actions, latents, score = planner(state, goal)
        assert planner.fully_optimized
        assert plan.shape == (hp.max_seq_len, 3 * hp.img_sz ** 2 + hp.nz_enc) and latents.shape == (hp.max_seq_len, hp.nz_enc)
        assert np.isfinite(score) and len(planner.logs) == 3
        res[dev] = (plan, latents, score, [np.asarray(l.elite_scores, dtype=np.float64).reshape(-1) for l in planner.logs],
                    planner._sampler.sample(), planner.logs)
    for a, b in zip(res[False][3], res[True][3]):                  # per-iteration best costs
        assert np.array_equal(a.astype(np.float32), b.astype(np.float32)), (a, b)
    assert np.array_equal(res[False][4], res[True][4])              # the fully optimised latent tree
    assert np.array_equal(res[False][0], res[True][0]) and np.array_equal(res[False][1], res[True][1]) and res[False][2] == res[True][2]
    # per-iteration plans grow: start / subgoal / goal, then 5 frames, then the dense sequence (+ appended goal)
    lens_np = [l.elite_rollouts[0].shape[0] for l in res[False][5]]
    lens_dev = [len(l.elite_rollouts[0]) for l in res[True][5]]
    assert lens_np[0] <= lens_np[1] <= lens_np[2]
    assert lens_dev[:2] == lens_np[:2]


def test_hierarchical_planner_fast_draws(setup):
    """fast_draws: the same search on a seeded numpy Generator that draws only the rows the search keeps (tree_latent_search._draw):
    reproducible from the seed, complete after len(rates) + 1 rounds, plan of the model's horizon, np.random untouched"""
    from video_gcp_amd.planning import GCPImageSimulator, LearnedCostEstimate, HierarchicalCEMPlanner
    hp, sd, model = setup
    state, goal = _env_images(hp, 3)

    def run(seed):
        planner = HierarchicalCEMPlanner(GCPImageSimulator(model, pred_length=False), LearnedCostEstimate(model), hp.hierarchy_levels, [3, 2],
                                         n_ll_samples=2, action_dim=hp.nz_vae, max_seq_len=hp.max_seq_len, fast_draws=True, seed=seed)
        plan, actions, latents, score = planner(state, goal)
        assert planner.fully_optimized and len(planner.logs) == 3 and np.isfinite(score)
        assert plan.shape == (hp.max_seq_len, 3 * hp.img_sz ** 2 + hp.nz_enc)
        return plan, score
    np.random.seed(9)
    before = np.random.get_state()[1].copy()
    (p1, s1), (p2, s2), (p3, s3) = run(4), run(4), run(5)
    assert np.array_equal(np.random.get_state()[1], before)
    assert np.array_equal(p1, p2) and s1 == s2
    assert not np.array_equal(p1, p3)


def test_plan_entry_point_cem_and_hierarchical(tmp_path):
    """`python -m video_gcp_amd.plan` counterpart of the planner call behind planning/run.py: plans for seeded start / goal pairs"""
    import numpy as np
    from video_gcp_amd.plan import main
    out = str(tmp_path / "plans")
    res = main(["--config", "c1", "--nstart_goal_pairs", "2", "--candidates", "32", "--iters", "2", "--out", out])
    assert len(res) == 2 and all(np.isfinite(r["cost"]) and r["plan_len"] >= 3 for r in res)
    z = np.load(out + "/plan_0.npz")
    assert z["image_plan"].shape[0] == res[0]["plan_len"] and z["latents"].shape[0] == res[0]["plan_len"]
    res = main(["--config", "c1", "--nstart_goal_pairs", "1", "--planner", "hierarchical"])
    assert len(res) == 1 and np.isfinite(res[0]["cost"])


def test_rollout_with_predicted_lengths_matches_oracle(setup):
    """the reference's rollout runs under val_mode() = pred_length=True (cem_simulator.py:29, base_gcp.py:219-226): every
    candidate's length is a draw from the length predictor; fed draws -> bit-exact lengths, pruned rollouts vs the oracle"""
    from oracle import gcp_model_oracle as O
    from video_gcp_amd.planning import GCPImageSimulator, env2planner
    hp, sd, model = setup
    state, goal = _env_images(hp, 5)
    n, T = 5, hp.max_seq_len
    z = torch.randn(n, hp.n_nodes, hp.nz_vae, generator=torch.Generator().manual_seed(2))
    len_u = torch.tensor([0.05, 0.3, 0.55, 0.8, 0.97])
    sim = GCPImageSimulator(model, append_latent=True)            # pred_length=True is the default, as in the reference
    r = sim.rollout_device(state, goal, z, T, len_u=len_u)
    inp = dict(I_0=env2planner(np.repeat(state, n, 0)), I_g=env2planner(np.repeat(goal, n, 0)), z=z, len_u=len_u,
               end_ind=torch.full((n,), T - 1, dtype=torch.long), start_ind=torch.zeros(n, dtype=torch.long))
    ref = O.forward(sd, hp, inp, sample_prior=True, training_bn=False, use_pred_length=True)
    lens = r.lengths.cpu().long()
    assert torch.equal(lens, ref["end_ind"] + 1) and len(set(lens.tolist())) > 1
    for i in range(n):
        assert_close(r.images[i, :lens[i]], ref["pruned_prediction"][i], 2e-5, 0, "pruned rollout")
        assert_close(r.latents[i, :lens[i]], ref["model_enc_seq_list"][i], 5e-5, 1e-4, "latents")
    # the CEM planner on variable-length rollouts: deterministic (shared length draws), finite scores
    from video_gcp_amd.planning import LearnedCostEstimate, SimpleTreeCEMSampler, CEMPlanner
    res = []
    for _ in range(2):
        sampler = SimpleTreeCEMSampler(float("inf"), None, hp.nz_vae, 1.0, n_level_hierarchy=hp.hierarchy_levels, device="cuda", seed=3)
        planner = CEMPlanner(GCPImageSimulator(model), LearnedCostEstimate(model), sampler, n_iters=2, batch_size=16, elite_frac=0.25,
                             max_seq_len=hp.max_seq_len)
        plan, actions, latents, score = planner(state, goal)
        res.append((plan, score))
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1] and np.isfinite(res[0][1])


def test_image_cem_policy_closed_loop(setup):
    """ImageCEMPolicy.act (planner_policy.py:89-113,216-227): plans on the first call, re-plans on the interval, and in closed-loop
    execution re-infers every action with the inverse model from the current image and the next planned latent"""
    from oracle import gcp_model_oracle as O
    from video_gcp_amd.planning import GCPImageSimulator, LearnedCostEstimate, SimpleTreeCEMSampler, CEMPlanner, ImageCEMPolicy, env2planner
    hp, sd, model = setup
    rng = np.random.RandomState(9)
    frames = rng.randint(0, 256, size=(4, 1, hp.img_sz, hp.img_sz, 3)).astype(np.uint8)
    goal = rng.randint(0, 256, size=(1, hp.img_sz, hp.img_sz, 3)).astype(np.uint8)
    sampler = SimpleTreeCEMSampler(float("inf"), None, hp.nz_vae, 1.0, n_level_hierarchy=hp.hierarchy_levels, device="cuda", seed=5)
    planner = CEMPlanner(GCPImageSimulator(model, pred_length=False), LearnedCostEstimate(model), sampler, n_iters=1, batch_size=8,
                         elite_frac=0.25, max_seq_len=hp.max_seq_len)
    policy = ImageCEMPolicy(model, planner, replan_interval=2, num_max_replans=10, closed_loop_execution=True)
    acts = []
    for t in range(3):
        acts.append(policy.act(t=t, i_tr=0, images=frames[:t + 1], goal_image=goal).actions)
    assert policy.num_replans == 2 and policy.current_exec_step == 1       # planned at t = 0 and t = 2
    assert all(a.shape == (hp.n_actions,) for a in acts)
    # the last action against the oracle: inverse model on (encoder(current frame), planned latent 1)
    e0, _ = O.encoder(sd, hp, env2planner(frames[2]), training=False)
    want = O.predictor(sd, "inv_mdl.action_pred", hp, e0[:, :, 0, 0], torch.as_tensor(policy.latent_plan[1][None]))
    assert_close(acts[2], want[0], 5e-5, 1e-4, "closed-loop action")
    policy.closed_loop_execution = False
    a = policy.act(t=3, i_tr=0, images=frames, goal_image=goal).actions
    assert np.array_equal(a, policy.action_plan[1])


def test_hierarchical_planner_at_the_reference_control_shape():
    """The reference's own control configuration (experiments/control/25room/gcp_tree/mod_hyper.py:34-71): 32 x 32 images,
    max_seq_len 200, hierarchy_levels 8 (255 nodes), balanced binding, HierarchicalImageCEMPlanner with sampling rates [10, 10],
    n_iters 3, batch_size 10, learned cost.  The device-resident search against the reference's data flow (numpy rollouts + the
    optimizer class pinned to the reference's) on identical np.random draws: same per-iteration best costs, same optimised latent
    tree, same plan; sample shapes per iteration as the reference produces them ([10, 255, D], [10, 255, D], [5, 255, D])."""
    import video_gcp_amd as V
    from video_gcp_amd.model import GCPTreeModel
    from video_gcp_amd.planning import GCPImageSimulator, LearnedCostEstimate, HierarchicalCEMPlanner
    hp = V.config("c1", max_seq_len=200, hierarchy_levels=8, batch_size=10)
    assert hp.img_sz == 32 and hp.n_nodes == 255 and hp.nz_vae == 256 and hp.matching_type == "balanced"
    model = GCPTreeModel(hp, params=V.init_params(hp, seed=3, randomize_affine=True), device="cuda")
    model.eval()                                                         # planner_policy.py:51
    state, goal = _env_images(hp, 5)
    res = {}
    for dev in (False, True):
        np.random.seed(1)
        planner = HierarchicalCEMPlanner(GCPImageSimulator(model, pred_length=False), LearnedCostEstimate(model), hp.hierarchy_levels, [10, 10],
                                         action_dim=hp.nz_vae, max_seq_len=hp.max_seq_len, device_resident=dev)
        shapes = []
        sample0 = planner._sampler.sample
        planner._sampler.sample = lambda *a, **k: (lambda s: (shapes.append(tuple(s.shape)), s)[1])(sample0(*a, **k))
        plan, actions, latents, score = planner(state, goal)
        assert planner.fully_optimized and np.isfinite(score)
        # (the planner draws twice per iteration: the population, then the best tree after the level is fixed, cem_planner.py:211-216)
        assert shapes[0::2][:3] == [(10, 255, 256), (10, 255, 256), (5, 255, 256)], shapes
        assert plan.shape[1] == 3 * 32 * 32 + hp.nz_enc and 3 <= plan.shape[0] <= hp.max_seq_len + 1
        res[dev] = (plan, latents, score, [np.asarray(l.elite_scores, dtype=np.float32).reshape(-1) for l in planner.logs])
    for a, b in zip(res[False][3], res[True][3]):
        assert np.array_equal(a, b), (a, b)
    assert np.array_equal(res[False][0], res[True][0]) and np.array_equal(res[False][1], res[True][1]) and res[False][2] == res[True][2]


@pytest.mark.parametrize("cost_name,dense", [("EuclideanDistance", True), ("EuclideanDistance", False), ("EuclideanPathLength", True),
                                             ("StepPathLength", False), ("L2ImageCost", True)])
def test_hand_written_costs_on_device_rollouts(setup, cost_name, dense):
    """the hand-written planner costs (cost_fcn.py:42-77) over the padded device rollout (gcpx_rollout_cost, one launch) equal the
    reference's host computation over the per-candidate numpy lists the simulator returns (cem_simulator.py:14-43); and a CEM planner
    built with such a cost (the reference's default is EuclideanPathLength, cem_planner.py:36) picks its elites from those scores"""
    from video_gcp_amd import planning as P
    hp, sd, model = setup
    state, goal = _env_images(hp, 11)
    sim = P.GCPImageSimulator(model, append_latent=True, pred_length=False)
    n, T = 6, hp.max_seq_len
    z = torch.randn(n, hp.n_nodes, hp.nz_vae, generator=torch.Generator().manual_seed(2))
    cost = getattr(P, cost_name)(dense, 1.5)
    host = sim.rollout(state, goal, z.numpy(), T)
    if cost_name == "L2ImageCost":
        g = goal.astype(np.float32) / 255.0
    else:
        g = np.random.RandomState(3).randn(host.predictions[0].shape[1]).astype(np.float32)
    want = cost([p.copy() for p in host.predictions], g)
    r = sim.rollout_device(state, goal, z, T)
    got = cost.rollout_cost_device(sim.predictions_device(r), r.lengths, g)
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5, atol=1e-5)
    if cost_name == "L2ImageCost":
        sampler = P.SimpleTreeCEMSampler(float("inf"), None, hp.nz_vae, 1.0, n_level_hierarchy=hp.hierarchy_levels, device="cuda", seed=4)
        planner = P.CEMPlanner(sim, cost, sampler, n_iters=2, batch_size=8, elite_frac=0.25, max_seq_len=T)
        plan, actions, latents, score = planner(state, g[None] if g.ndim == 3 else g)
        assert planner.decode_candidates and np.isfinite(score) and plan.shape[1] == 3 * hp.img_sz ** 2 + hp.nz_enc
        assert float(planner.logs[-1].elite_scores[0]) <= float(planner.logs[0].mean_score)


def test_pddm_sampler_in_the_planner(setup):
    """PDDMSampler (sampler.py:52-71) as the planner's sampler: correlated draws on the device, exp(-score)-weighted refit"""
    from video_gcp_amd import planning as P
    hp, sd, model = setup
    state, goal = _env_images(hp, 12)

    class TreePDDM(P.PDDMSampler):
        def __init__(self, *a, n_level_hierarchy, **kw):
            super().__init__(a[0], 2 ** n_level_hierarchy - 1, *a[2:], **kw)
    sampler = TreePDDM(float("inf"), None, hp.nz_vae, 1.0, n_level_hierarchy=hp.hierarchy_levels, device="cuda", seed=5)
    planner = P.CEMPlanner(P.GCPImageSimulator(model, pred_length=False), P.LearnedCostEstimate(model), sampler, n_iters=2, batch_size=16,
                           elite_frac=0.25, max_seq_len=hp.max_seq_len)
    plan, actions, latents, score = planner(state, goal)
    assert np.isfinite(score) and float(sampler.mean.abs().max()) > 0 and torch.equal(sampler.std, torch.ones_like(sampler.std))


def test_action_conditioned_simulator_and_flat_cem():
    """ActCondGCPImageSimulator (cem_simulator.py:99-104; planner_policy.py:231 with act_cond): the candidates are action sequences
    rolled out by an action-conditioned flat predictor (base_configs/vmpc.py).  Rollouts against the oracle, then the flat CEM loop
    (FlatCEMSampler over [T - 1, n_actions]) with the learned cost on the rolled-out latents."""
    import video_gcp_amd as V
    from oracle import gcp_sequential_oracle as S
    from video_gcp_amd.sequential import GCPSequentialModel
    from video_gcp_amd.planning import ActCondGCPImageSimulator, CEMPlanner, FlatCEMSampler, LearnedCostEstimate, env2planner
    hp = V.config("c1", nz_mid_lstm=128, lstm_init="zero", action_conditioned_pred=True, non_goal_conditioned=True, nz_vae=0,
                  var_inf="deterministic")
    sd = V.init_params_sequential(hp, seed=2, randomize_affine=True)
    model = GCPSequentialModel(hp, params=sd, device="cuda")
    model.eval()
    state, goal = _env_images(hp, 4)
    n, T = 4, hp.max_seq_len
    acts = torch.randn(n, T - 1, hp.n_actions, generator=torch.Generator().manual_seed(0))
    sim = ActCondGCPImageSimulator(model)
    got = sim.rollout(state, goal, acts.numpy(), T)
    inp = dict(I_0=env2planner(np.repeat(state, n, 0)), I_g=env2planner(np.repeat(goal, n, 0)), actions=acts,
               end_ind=torch.full((n,), T - 1, dtype=torch.long))
    ref = S.forward(sd, hp, inp, training_bn=False, sample_prior=True)
    for i in range(n):
        want = torch.cat((ref["pruned_prediction"][i].reshape(T, -1), ref["model_enc_seq_list"][i]), -1)
        assert_close(got.predictions[i], want, 5e-5, 1e-3, "predictions")
        assert_close(got.latents[i], ref["model_enc_seq"][i], 1e-4, 1e-3, "latents")
    # the samples may arrive with the image simulator's two unit axes (cem_simulator.py:78,102)
    r5 = sim.rollout_device(state, goal, acts[..., None, None], T)
    assert_close(r5.latents, ref["model_enc_seq"], 1e-4, 1e-3, "latents, 5-d samples")

    def plan(seed):
        sampler = FlatCEMSampler(clip_val=float("inf"), n_steps=T - 1, action_dim=hp.n_actions, initial_std=1.0, device="cuda", seed=seed)
        planner = CEMPlanner(sim, LearnedCostEstimate(model), sampler, n_iters=2, batch_size=16, elite_frac=0.25, max_seq_len=T)
        return planner(state, goal), planner
    (pred, actions, latents, cost), planner = plan(3)
    assert pred.shape == (T, 3 * hp.img_sz ** 2 + hp.nz_enc) and latents.shape == (T, hp.nz_enc) and np.isfinite(cost)
    assert len(planner.logs) == 2 and all(bool(torch.isfinite(l.elite_scores).all()) for l in planner.logs)
    (pred2, _, _, cost2), _ = plan(3)
    assert cost2 == cost and np.array_equal(pred2, pred)


def test_flat_cem_loop_on_device_matches_executed_reference():
    """CEMPlanner.iterate / __call__ on the device (draws replayed from the reference's np.random stream, the stub simulator's padded
    rollouts resident in HBM, scores by gcpx_rollout_cost = EuclideanPathLength, elites by the device argsort, refit on the device)
    against the EXECUTED reference loop's fixtures (tests/golden/ref_cem_loop.npz, cem_planner.py:55-135): the elites are the same
    candidates wherever the reference's own scores separate them by more than float32 resolution, scores / refit within float32"""
    import os
    import sys
    from video_gcp_amd import planning as P
    from video_gcp_amd.model import Outputs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    from planner_stubs import flat_stub_rollout
    g = np.load(os.path.join(root, "tests", "golden", "ref_cem_loop.npz"))
    steps, ad = (int(v) for v in g["flat_shape"])

    class Replay(P.FlatCEMSampler):
        def sample(self, n_samples):
            return self.from_unit_noise(torch.as_tensor(self._eps.pop(0), dtype=torch.float32, device=self.device))

    class Sim:
        def predictions_device(self, r):
            return r.latents

        def rollout_device(self, state, goal, samples, rollout_len):
            preds, _ = flat_stub_rollout(samples.double().cpu().numpy(), rollout_len, np.float32)
            pad = np.zeros((len(preds), rollout_len, preds[0].shape[1]), np.float32)
            for i, p in enumerate(preds):
                pad[i, :len(p)] = p
            return Outputs(latents=torch.as_tensor(pad, device="cuda"), lengths=torch.tensor([len(p) for p in preds], dtype=torch.int32, device="cuda"),
                           e_goal=None)

        def rollout(self, state, goal, samples, rollout_len, prune=False):
            preds, lats = flat_stub_rollout(np.asarray(samples, dtype=np.float64), rollout_len, np.float32)
            return Outputs(predictions=preds, actions=[p[1:] - p[:-1] for p in preds], latents=lats)

    checked = 0
    for ci, (batch, efrac, mrb, nbytes, clip, seed) in enumerate(g["flat_cases"]):
        if mrb < batch:
            continue                                     # (the reference's chunked rollout mis-orders its scores: tests/test_planning_cpu.py)
        batch = int(batch)
        goal = g[f"flat{ci}_goal"].astype(np.float32)
        sampler = Replay(float(clip), steps, ad, 0.6, device="cuda")
        sampler._eps = [g[f"flat{ci}_it{it}_eps"] for it in range(3)]
        planner = P.CEMPlanner(Sim(), P.EuclideanPathLength(True, 2.0), sampler, n_iters=3, batch_size=batch, elite_frac=float(efrac), max_seq_len=steps)
        sampler.init()
        for it in range(3):
            best, best_scores, scores = planner.iterate(None, goal)
            want = g[f"flat{ci}_it{it}_scores"]
            np.testing.assert_allclose(scores.cpu().numpy(), want, rtol=3e-5, atol=1e-5)
            n_el = len(g[f"flat{ci}_it{it}_elite_idx"])
            srt = np.sort(want)
            if srt[n_el] - srt[n_el - 1] > 1e-3 and np.min(np.diff(srt[:n_el + 1])) > 1e-3:
                got_idx = torch.argsort(scores, stable=True)[:n_el].cpu().numpy()
                assert np.array_equal(got_idx, g[f"flat{ci}_it{it}_elite_idx"]), (ci, it)
                np.testing.assert_allclose(sampler.mean.cpu().numpy(), g[f"flat{ci}_it{it}_mean"], rtol=1e-4, atol=1e-5)
                np.testing.assert_allclose(sampler.std.cpu().numpy(), g[f"flat{ci}_it{it}_std"], rtol=1e-4, atol=1e-5)
                checked += 1
            else:
                break                                    # a near-tie the float32 cost may order differently: the chain is not comparable further
    assert checked >= 3
