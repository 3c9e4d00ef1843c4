"""-m gpu: gcp_sequential (flat VRNN baseline, sequential.py:13-131) through the HIP kernels vs its CPU oracle."""
import pytest
import torch

from helpers import make_inputs, assert_close

pytestmark = pytest.mark.gpu


def _run(dist, training_bn, sample_prior=False):
    import video_gcp_amd as V
    from video_gcp_amd.sequential import GCPSequentialModel
    from oracle import gcp_sequential_oracle as S
    hp = V.config("c1", decoder_distribution=dist, nz_mid_lstm=128, lstm_init="zero")
    sd = V.init_params_sequential(hp, seed=1, randomize_affine=True)
    model = GCPSequentialModel(hp, params=sd, device="cuda")
    model.train(training_bn)
    inputs, noise, _ = make_inputs(hp, seed=3, variant="B")
    noise = noise[:, :hp.max_seq_len - 1].contiguous()
    ref = S.forward(sd, hp, inputs, noise=noise, training_bn=training_bn, sample_prior=sample_prior)
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    if sample_prior:
        with model.val_mode():
            out = model(dev_in, "train", noise=noise.cuda())
    else:
        out = model(dev_in, "train", noise=noise.cuda())
    torch.cuda.synchronize()
    return hp, sd, model, inputs, dev_in, ref, out


@pytest.mark.parametrize("dist", ["discrete_logistic_mixture", "gaussian"])
@pytest.mark.parametrize("training_bn", [False, True])
def test_sequential_posterior_forward_and_losses(dist, training_bn):
    from oracle import gcp_sequential_oracle as S
    hp, sd, model, inputs, dev_in, ref, out = _run(dist, training_bn)
    # 19 recurrent steps amplify fp32 rounding: latents to 1e-4, pixels to 5e-5
    assert_close(out.dense_rec.encodings, ref["encodings"], 1e-4, 1e-3, "encodings")
    assert_close(out.dense_rec.p_z, ref["p_z"], 1e-4, 1e-3, "p_z")
    assert_close(out.dense_rec.q_z, ref["q_z"], 1e-4, 1e-3, "q_z")
    assert_close(out.dense_rec.images, ref["images"], 5e-5, 0, "images")
    assert_close(out.seq_len_logits, ref["seq_len_logits"], 5e-5, 1e-4, "seq_len_logits")
    pruned = model.pruned_prediction(out)
    assert [p.shape[0] for p in pruned] == [p.shape[0] for p in ref["pruned_prediction"]]
    aux = model.aux_outputs(out)
    assert_close(aux.model_enc_seq, ref["model_enc_seq"], 1e-4, 1e-3, "model_enc_seq")
    assert_close(aux.regressed_state, ref["regressed_state"], 1e-4, 1e-3, "regressed_state")
    # (train phase with the index draws fed: ONE sampled frame pair per sequence, inverse_mdl.py:136-178, and the cost model's segment)
    assert_close(aux.actions, ref["actions_sampled"], 1e-4, 1e-3, "actions")
    assert_close(aux.cost, ref["cost"], 1e-4, 1e-3, "cost")
    assert_close(aux.cost_target.reshape(-1), ref["cost_target"].float().reshape(-1), 1e-3, 1e-4, "cost_target")
    ref_losses, ref_total = S.losses(sd, hp, inputs, ref)
    losses = model.loss(dev_in, out)
    for name, (val, w) in ref_losses.items():
        got = float(losses[name].value)
        assert abs(got - float(val)) <= 1e-4 * abs(float(val)) + 1e-5, (name, got, float(val))
    assert abs(float(model.get_total_loss(dev_in, losses).value) - float(ref_total)) <= 1e-4 * abs(float(ref_total))


def test_sequential_prior_sampling():
    hp, sd, model, inputs, dev_in, ref, out = _run("discrete_logistic_mixture", False, sample_prior=True)
    assert_close(out.dense_rec.encodings, ref["encodings"], 1e-4, 1e-3, "encodings")
    assert_close(out.dense_rec.images, ref["images"], 5e-5, 0, "images")


def _train_setup(graph=True, **over):
    import video_gcp_amd as V
    from video_gcp_amd.sequential import GCPSequentialModel
    from video_gcp_amd.training_sequential import SequentialTrainStep
    hp = V.config("c1", nz_mid_lstm=128, lstm_init="zero", **over)
    sd = V.init_params_sequential(hp, seed=1, randomize_affine=True)
    model = GCPSequentialModel(hp, params=sd, device="cuda")
    model.use_graph = graph
    return hp, sd, model, SequentialTrainStep(model, lr=1e-3)


@pytest.mark.parametrize("variant", ["B", "A"])
def test_sequential_gradients_match_autograd(variant):
    """Training step of the flat VRNN baseline (train.py:155-163 with configuration['model'] = SequentialModel,
    sequential.py:13-131): every parameter gradient of the explicit backward pass through the T - 1 recurrent steps against torch
    autograd over the oracle (stated tolerance as for the tree model: 1e-3 of the gradient's max-abs + 5e-7)."""
    from oracle import gcp_sequential_oracle as S
    hp, sd, model, tr = _train_setup()
    inputs, noise, _ = make_inputs(hp, seed=7, variant=variant)
    noise = noise[:, :hp.max_seq_len - 1].contiguous()
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    for _ in range(2):                                   # the second call replays the captured forward graph
        out = tr.backward(dev_in, noise.cuda())
    torch.cuda.synchronize()
    gref, res, total, _ = S.gradients(sd, hp, inputs, noise)
    assert abs(float(out.raw["losses"][5]) - float(total)) <= 1e-4 * abs(float(total))
    got = tr.named_grads()
    bad, trained = [], 0
    for k, g in gref.items():
        err, scale = float((got[k].cpu() - g).abs().max()), float(g.abs().max())
        trained += scale > 0
        if err > 1e-3 * scale + 5e-7:
            bad.append((k, err, scale))
    assert not bad, bad[:10]
    # everything the loss terms reach is trained: encoder, decoder, the three recurrent nets, the length predictor, the state regressor
    for pre in ("encoder.", "decoder.", "dense_rec.lstm.cell.prior_lstm.", "dense_rec.lstm.cell.inf_lstm.", "dense_rec.lstm.cell.gen_lstm.",
                "length_pred.", "state_regressor.", "inv_mdl.", "cost_mdl."):
        ks = [k for k in gref if k.startswith(pre)]
        assert ks and any(float(got[k].abs().max()) > 0 for k in ks), pre


def test_sequential_cell_backward_in_the_gemm_epilogue_gives_the_same_gradient():
    """fuse_lstm_bwd (off by default: it does not pay, NOTEBOOK round 4): every LSTM cell backward of the 79-step chains rides in the
    epilogue of the data-gradient GEMM that feeds it (gcpx_gemm_args.lstm_bwd) — the same device function on the same values: the
    gradient of the plain plan, bit for bit."""
    hp, sd, model, tr = _train_setup(False)
    _, _, model2, tr2 = _train_setup(False)
    tr2.fuse_lstm_bwd = True
    inputs, noise, _ = make_inputs(hp, seed=7, variant="B")
    noise = noise[:, :hp.max_seq_len - 1].contiguous()
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    tr.backward(dev_in, noise.cuda())
    tr2.backward(dev_in, noise.cuda())
    torch.cuda.synchronize()
    n1 = sum(1 for op in tr.last_bplan.ops if not op[0].startswith("@"))
    n2 = sum(1 for op in tr2.last_bplan.ops if not op[0].startswith("@"))
    assert n1 - n2 >= 3 * (hp.max_seq_len - 1), (n1, n2)
    assert torch.equal(tr.grad, tr2.grad)


@pytest.mark.parametrize("env", [{"GCPX_SEQ_CHAINS": "serial"}, {"GCPX_SEQ_CHAINS": "overlap"}, {"GCPX_SEQ_CHAINS": "lockstep3"}, {"GCPX_SEQ_LIVE_FOLDS": "0"},
                                 {"GCPX_BWD_SEGMENTS": "3"}, {"GCPX_BWD_SEGMENTS": "-1"}])
def test_sequential_backward_schedules_agree(env, monkeypatch):
    """The default schedule of the flat model's training step (prior chain ahead on its lane as small graphs, generator step t and
    inference step t + 1 in the same grouped launches, folded 3-launch forward steps) against the other ones the switches select: the
    round-5 serial order, three lanes with step-by-step events, the unfolded training forward, every run of launches as a graph / none.
    Same mathematics, different summation order inside the `out` data-gradient GEMM (the tripled weights) and a float64-folded
    forward: 2e-4 of each gradient's max-abs; the graph switches replay the very same launches: bit for bit."""
    hp, sd, model, tr = _train_setup(False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _, _, model2, tr2 = _train_setup(False)
    inputs, noise, _ = make_inputs(hp, seed=9, variant="B")
    noise = noise[:, :hp.max_seq_len - 1].contiguous()
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    for _ in range(3):                                    # (the third call of a trainer with segment graphs replays them)
        tr.backward(dev_in, noise.cuda())
        tr2.backward(dev_in, noise.cuda())
    torch.cuda.synchronize()
    if "GCPX_BWD_SEGMENTS" in env:
        assert any(op[0] == "@graph" for op in (tr2.last_bplan.rec.get("_segments") or [])) == (env["GCPX_BWD_SEGMENTS"] != "-1")
        assert torch.equal(tr.grad, tr2.grad)
        return
    g1, g2 = tr.named_grads(), tr2.named_grads()
    bad = [(k, float((g1[k] - g2[k]).abs().max()), float(g1[k].abs().max())) for k in g1
           if float((g1[k] - g2[k]).abs().max()) > 2e-4 * float(g1[k].abs().max()) + 1e-7]
    assert not bad, bad[:8]


def test_sequential_two_trainers_keep_identical_parameters():
    """Two trainers of the flat model fed the same minibatches hold bit-identical parameters, optimizer moments and folded packs after three
    steps: no float atomics in the backward, and the folded layer-0 packs of the training forward — products formed by a library GEMM behind
    every optimizer step — are formed under torch's deterministic switch (rocBLAS without atomics)."""
    hp, sd, m1, tr1 = _train_setup()
    _, _, m2, tr2 = _train_setup()
    for step in range(3):
        inputs, noise, _ = make_inputs(hp, seed=40 + step, variant="B")
        noise = noise[:, :hp.max_seq_len - 1].contiguous().cuda()
        dev_in = {k: v.cuda() for k, v in inputs.items()}
        tr1.step(dev_in, noise)
        tr2.step(dev_in, noise)
    torch.cuda.synchronize()
    assert torch.equal(m1.theta, m2.theta) and torch.equal(tr1.exp_avg, tr2.exp_avg) and torch.equal(tr1.exp_avg_sq, tr2.exp_avg_sq)
    for net in m1._nets:
        for key in ("lstm0f.w", "lstm0f.b") + (("lstm0ff.w", "lstm0ff.b") if net == "gen_lstm" else ()):
            assert torch.equal(m1.pk[net][key], m2.pk[net][key]), (net, key)
    assert not torch.are_deterministic_algorithms_enabled()        # the switch is restored behind every re-fold


def test_sequential_two_training_steps():
    """losses of two consecutive optimisation steps and the updated parameters against the oracle loop (RAdam)"""
    from oracle import gcp_sequential_oracle as S
    from oracle.radam_oracle import RAdamOracle
    hp, sd, model, tr = _train_setup()
    ref_sd = {k: v.clone() for k, v in sd.items()}
    opt = RAdamOracle(lr=1e-3)
    for step in range(2):
        inputs, noise, _ = make_inputs(hp, seed=20 + step, variant="B")
        noise = noise[:, :hp.max_seq_len - 1].contiguous()
        out = tr.step({k: v.cuda() for k, v in inputs.items()}, noise.cuda())
        torch.cuda.synchronize()
        gref, res, total, _ = S.gradients(ref_sd, hp, inputs, noise)
        assert abs(float(out.raw["losses"][5]) - float(total)) <= 1e-4 * abs(float(total)), step
        opt.step(ref_sd, gref)
        worst = max(float((model.sd[k].cpu() - ref_sd[k]).abs().max()) for k in gref)
        assert worst <= 2e-3 * 1e-3 * (step + 1) + 1e-7, (step, worst)


def test_sequential_training_c2_shapes_decreases_loss():
    """full-size shapes of the 25-room gcp_sequential configuration (64x64, T=80; batch 4, nz_mid_lstm 512): finite gradients and a
    falling loss on a fixed batch"""
    import video_gcp_amd as V
    from video_gcp_amd.sequential import GCPSequentialModel
    from video_gcp_amd.training_sequential import SequentialTrainStep
    hp = V.config("c2", batch_size=4, lstm_init="zero")
    model = GCPSequentialModel(hp, params=V.init_params_sequential(hp, seed=2), device="cuda")
    tr = SequentialTrainStep(model, lr=2e-3)
    inputs, noise, _ = make_inputs(hp, seed=5, variant="B")
    noise = noise[:, :hp.max_seq_len - 1].contiguous().cuda()
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    losses = []
    for _ in range(6):
        out = tr.step(dev_in, noise)
        losses.append(float(out.raw["losses"][5]))
    assert all(torch.isfinite(torch.tensor(losses))) and torch.isfinite(tr.grad).all()
    assert losses[-1] < losses[0], losses


# ---- flat-predictor variants of experiments/prediction/base_configs/vmpc.py:11-16 ----
_VARIANTS = {
    # the visual-MPC style predictor: action-conditioned, deterministic (no latent), not goal-conditioned
    "vmpc": dict(action_conditioned_pred=True, non_goal_conditioned=True, nz_vae=0, var_inf="deterministic"),
    # action conditioning alone: the VRNN keeps its latent, all three nets read the encoded action
    "act_vrnn": dict(action_conditioned_pred=True),
    # deterministic but goal-conditioned, no actions
    "det": dict(nz_vae=0, var_inf="deterministic"),
}


def _variant_setup(name, **over):
    import video_gcp_amd as V
    from video_gcp_amd.sequential import GCPSequentialModel
    hp = V.config("c1", nz_mid_lstm=128, lstm_init="zero", **_VARIANTS[name], **over)
    sd = V.init_params_sequential(hp, seed=1, randomize_affine=True)
    model = GCPSequentialModel(hp, params=sd, device="cuda")
    inputs, noise, _ = make_inputs(hp, seed=5, variant="B")
    noise = noise[:, :hp.max_seq_len - 1].contiguous() if hp.nz_vae else None
    return hp, sd, model, inputs, noise


@pytest.mark.parametrize("name", sorted(_VARIANTS))
def test_sequential_variants_forward_and_losses(name):
    """action_conditioned_pred (sequential.py:24-25,45-49; base_gcp.py:211-213), var_inf='deterministic' and non_goal_conditioned
    (base_gcp.py:163-170) against the oracle: rollout, images, every loss term"""
    from oracle import gcp_sequential_oracle as S
    hp, sd, model, inputs, noise = _variant_setup(name)
    ref = S.forward(sd, hp, inputs, noise=noise, training_bn=True)
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    keep = {k: v.clone() for k, v in dev_in.items()}
    out = model(dev_in, "train", noise=(noise.cuda() if noise is not None else None))
    torch.cuda.synchronize()
    assert all(torch.equal(dev_in[k], keep[k]) for k in keep), "the caller's tensors are left alone"
    assert_close(out.dense_rec.encodings, ref["encodings"], 1e-4, 1e-3, "encodings")
    assert_close(out.dense_rec.images, ref["images"], 5e-5, 0, "images")
    if hp.nz_vae:
        assert_close(out.dense_rec.p_z, ref["p_z"], 1e-4, 1e-3, "p_z")
        assert_close(out.dense_rec.q_z, ref["q_z"], 1e-4, 1e-3, "q_z")
    ref_losses, ref_total = S.losses(sd, hp, inputs, ref)
    losses = model.loss(dev_in, out)
    for key, (val, w) in ref_losses.items():
        got = float(losses[key].value)
        assert abs(got - float(val)) <= 1e-4 * abs(float(val)) + 1e-5, (key, got, float(val))
    assert abs(float(model.get_total_loss(dev_in, losses).value) - float(ref_total)) <= 1e-4 * abs(float(ref_total))
    if hp.deterministic:
        assert float(losses["kl"].value) == 0.0


@pytest.mark.parametrize("name", ["vmpc", "act_vrnn"])
def test_sequential_variants_rollout_from_actions(name):
    """the planner's call (cem_simulator.py:99-104): start / goal image and an action sequence, no ground-truth frames; running-stat
    BatchNorm, prior samples where the model has a latent"""
    from oracle import gcp_sequential_oracle as S
    hp, sd, model, inputs, noise = _variant_setup(name)
    model.train(False)
    feed = {k: inputs[k] for k in ("I_0", "I_g", "actions", "end_ind")}
    ref = S.forward(sd, hp, feed, noise=noise, training_bn=False, sample_prior=True)
    with model.val_mode():
        out = model({k: v.cuda() for k, v in feed.items()}, "train", noise=(noise.cuda() if noise is not None else None))
    torch.cuda.synchronize()
    assert_close(out.dense_rec.encodings, ref["encodings"], 1e-4, 1e-3, "encodings")
    assert_close(out.dense_rec.images, ref["images"], 5e-5, 0, "images")
    # a different action sequence gives a different rollout
    feed2 = dict(feed, actions=-feed["actions"])
    with model.val_mode():
        out2 = model({k: v.cuda() for k, v in feed2.items()}, "train", noise=(noise.cuda() if noise is not None else None))
    assert float((out2.dense_rec.encodings - torch.as_tensor(ref["encodings"]).cuda()).abs().max()) > 1e-3


@pytest.mark.parametrize("name", sorted(_VARIANTS))
def test_sequential_variants_gradients_match_autograd(name):
    """explicit backward of the variants against torch autograd over the oracle: the action encoder is trained through every net that
    reads the encoded action; a deterministic predictor has neither prior nor inference net (same tolerance as the base model)"""
    from oracle import gcp_sequential_oracle as S
    from video_gcp_amd.training_sequential import SequentialTrainStep
    hp, sd, model, inputs, noise = _variant_setup(name)
    tr = SequentialTrainStep(model, lr=1e-3)
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    for _ in range(2):
        out = tr.backward(dev_in, noise.cuda() if noise is not None else None)
    torch.cuda.synchronize()
    gref, res, total, _ = S.gradients(sd, hp, inputs, noise)
    assert abs(float(out.raw["losses"][5]) - float(total)) <= 1e-4 * abs(float(total))
    got = tr.named_grads()
    assert set(gref) <= set(got)
    bad = []
    for k, g in gref.items():
        err, scale = float((got[k].cpu() - g).abs().max()), float(g.abs().max())
        if err > 1e-3 * scale + 5e-7:
            bad.append((k, err, scale))
    assert not bad, bad[:10]
    pres = ["encoder.", "decoder.", "dense_rec.lstm.cell.gen_lstm."] + (["action_encoder."] if hp.action_conditioned_pred else []) + \
        ([] if hp.deterministic else ["dense_rec.lstm.cell.prior_lstm.", "dense_rec.lstm.cell.inf_lstm."])
    for pre in pres:
        ks = [k for k in gref if k.startswith(pre)]
        assert ks and any(float(got[k].abs().max()) > 0 for k in ks), pre
    assert hp.deterministic == (not any("prior_lstm" in k for k in gref))
