"""-m gpu: the HIP forward (through the C-ABI, as the model API drives it) against the CPU oracle on identical
weights, inputs and noise.  Integer bookkeeping must be bit-exact; floating point within the tolerances written
below (fp32 everywhere; differences are summation order only).  BASELINE.json: pixel MSE within 1e-5."""
import numpy as np
import pytest
import torch

from helpers import make_inputs, assert_close

pytestmark = pytest.mark.gpu

LAT_ATOL, LAT_RTOL = 5e-5, 1e-4      # latents / hidden states / distribution parameters
PIX_ATOL = 2e-5                      # decoded pixels in [-1, 1]


def _build(cfg, **over):
    import video_gcp_amd as V
    from video_gcp_amd.model import GCPTreeModel
    hp = V.config(cfg, **over)
    sd = V.init_params(hp, seed=1, randomize_affine=True)
    model = GCPTreeModel(hp, params=sd, device="cuda", materialize_distr=True)
    return hp, sd, model


def _check_common(hp, model, out, ref, posterior):
    from oracle import tree_index_oracle as TI
    bf = ref["tree_bf"]
    tree = out.tree
    assert_close(tree.bf.e_g_prime, bf["e_g_prime"], LAT_ATOL, LAT_RTOL, "e_g_prime")
    if "hidden" in bf:                                         # (the non-LSTM subgoal predictor carries no hidden state)
        assert_close(tree.bf.hidden_state, bf["hidden"], LAT_ATOL, LAT_RTOL, "hidden_state")
    assert_close(tree.bf.z, bf["z"], LAT_ATOL, LAT_RTOL, "z")
    assert_close(tree.bf.p_z_mu, bf["p_z_mu"], LAT_ATOL, LAT_RTOL, "p_z.mu")
    assert_close(tree.bf.p_z_log_sigma, bf["p_z_log_sigma"], LAT_ATOL, LAT_RTOL, "p_z.log_sigma")
    if posterior:
        assert_close(tree.bf.q_z_mu, bf["q_z_mu"], LAT_ATOL, LAT_RTOL, "q_z.mu")
        assert_close(tree.bf.q_z_log_sigma, bf["q_z_log_sigma"], LAT_ATOL, LAT_RTOL, "q_z.log_sigma")
    assert_close(tree.bf.images, bf["images"], PIX_ATOL, 0, "images")
    if hp.decoder_distribution == "discrete_logistic_mixture":
        assert_close(tree.bf.distr, bf["distr"], LAT_ATOL, LAT_RTOL, "distr")
    mse = float(((tree.bf.images.cpu() - bf["images"]) ** 2).mean())
    assert mse < 1e-9, mse
    assert_close(out.seq_len_logits, ref["seq_len_logits"], LAT_ATOL, LAT_RTOL, "seq_len_logits")
    assert_close(out.existence_predictor.existence, ref["existence"], LAT_ATOL, LAT_RTOL, "existence")
    # integer bookkeeping: bit-exact
    raw = out.raw
    assert np.array_equal(raw["leave"].cpu().numpy().astype(bool), ref["leave_df"].numpy())
    assert np.array_equal(raw["node_t"].cpu().numpy(), TI.balanced_timesteps_bf(ref["end_ind"].numpy(), hp.hierarchy_levels,
                          hp.max_seq_len)[:, np.argsort(TI.bf2df_perm(hp.hierarchy_levels))])
    pruned = model.pruned_prediction(out)
    assert [p.shape[0] for p in pruned] == [p.shape[0] for p in ref["pruned_prediction"]]
    for a, b in zip(pruned, ref["pruned_prediction"]):
        assert_close(a, b, PIX_ATOL, 0, "pruned_prediction")
    aux = model.aux_outputs(out)
    assert_close(aux.model_enc_seq, ref["model_enc_seq"], LAT_ATOL, LAT_RTOL, "model_enc_seq")
    assert_close(aux.regressed_state, ref["regressed_state"], LAT_ATOL, LAT_RTOL, "regressed_state")
    if "actions" in ref:
        # posterior path: ONE sampled frame pair per sequence (inverse_mdl.py:136-178); val_mode: the full sequence (:110-134)
        assert_close(aux.actions, ref["actions"], LAT_ATOL, LAT_RTOL, "actions")
    if "cost" in ref:
        assert_close(aux.cost, ref["cost"], LAT_ATOL, LAT_RTOL, "cost")
        assert_close(aux.cost_target, ref["cost_target"], 0, 2e-6, "cost_target")      # float32 sum of ~1e4 row norms


@pytest.mark.parametrize("training_bn", [False, True])
@pytest.mark.parametrize("variant", ["A", "B"])
def test_posterior_forward_c1(training_bn, variant):
    from oracle import gcp_model_oracle as O
    hp, sd, model = _build("c1")
    model.train(training_bn)
    inputs, noise, _ = make_inputs(hp, seed=2, variant=variant)
    ref = O.forward(sd, hp, inputs, noise=noise, training_bn=training_bn)
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    out = model(dev_in, "train", noise=noise.cuda())
    torch.cuda.synchronize()
    _check_common(hp, model, out, ref, posterior=True)
    assert_close(out.soft_matched_estimates, ref["soft_matched_estimates"], PIX_ATOL, 0, "soft_matched_estimates")
    from oracle import tree_index_oracle as TI
    want_f2n = TI.bf2df_perm(hp.hierarchy_levels)[ref["matched_idx"].numpy()]
    assert np.array_equal(out.raw["frame2node"].cpu().numpy(), want_f2n)
    # graph replay gives the same bits as the first (eager + captured) run
    img1 = out.images_df.clone()
    out2 = model(dev_in, "train", noise=noise.cuda())
    torch.cuda.synchronize()
    assert torch.equal(out2.images_df, img1)


def test_prior_and_given_z_c1():
    """planning paths: val_mode() prior sampling, and candidate latents z fed in depth-first order
    (cem_simulator.py:19-31), both with eval-mode BatchNorm (planner_policy.py:51)."""
    from oracle import gcp_model_oracle as O
    hp, sd, model = _build("c1")
    model.eval()
    inputs, noise, z = make_inputs(hp, seed=3, variant="A")
    plan_in = {k: inputs[k] for k in ("I_0", "I_g", "end_ind", "start_ind")}
    ref = O.forward(sd, hp, plan_in, noise=noise, sample_prior=True, training_bn=False)
    with model.val_mode(pred_length=False):
        out = model({k: v.cuda() for k, v in plan_in.items()}, "train", noise=noise.cuda())
    torch.cuda.synchronize()
    _check_common(hp, model, out, ref, posterior=False)
    assert "actions" in ref and ref["actions"].dim() == 3          # val_mode: inverse model over the full sequence
    zin = dict(plan_in, z=z)
    ref = O.forward(sd, hp, zin, sample_prior=True, training_bn=False)
    with model.val_mode(pred_length=False):
        out = model({k: v.cuda() for k, v in zin.items()}, "train")
    torch.cuda.synchronize()
    _check_common(hp, model, out, ref, posterior=False)


@pytest.mark.parametrize("feed_end_ind", [True, False])
def test_sampled_sequence_length_c1(feed_end_ind):
    """val_mode(pred_length=True) (base_gcp.py:219-226, what the planner's rollout runs under, cem_simulator.py:29-31): the sequence
    length is a draw from the length predictor (clamped to >= 2) and REPLACES a fed end_ind; everything downstream — balanced
    binding, pruning, the latent-space heads — follows the drawn length.  The categorical draw is fed as one uniform number per
    sequence; lengths and the integer bookkeeping must be bit-exact."""
    from oracle import gcp_model_oracle as O
    hp, sd, model = _build("c1", batch_size=6)
    model.eval()
    inputs, noise, z = make_inputs(hp, seed=11, variant="A")
    plan_in = {k: inputs[k] for k in ("I_0", "I_g", "start_ind")}
    if feed_end_ind:
        plan_in["end_ind"] = inputs["end_ind"]
    plan_in["z"] = z
    len_u = torch.tensor([0.03, 0.21, 0.48, 0.62, 0.87, 0.995])
    ref = O.forward(sd, hp, dict(plan_in, len_u=len_u), sample_prior=True, training_bn=False, use_pred_length=True)
    with model.val_mode():                                         # pred_length=True is the default, as in the reference
        out = model({k: v.cuda() for k, v in dict(plan_in, len_u=len_u).items()}, "train")
    torch.cuda.synchronize()
    assert torch.equal(out.end_ind.cpu(), ref["end_ind"])
    assert int(ref["end_ind"].min()) >= 2 and len(set(ref["end_ind"].tolist())) > 1
    assert torch.equal(out.raw["seq_len"].cpu().long(), ref["end_ind"] + 1)
    _check_common(hp, model, out, ref, posterior=False)
    # without a fed draw the lengths come from the device RNG: still valid lengths
    with model.val_mode():
        out = model({k: v.cuda() for k, v in plan_in.items()}, "train")
    torch.cuda.synchronize()
    e = out.end_ind.cpu()
    assert int(e.min()) >= 2 and int(e.max()) <= hp.max_seq_len - 1
    # a loader that filled input_buffer('end_ind') in place (the zero-copy hand-over) keeps its ground-truth lengths: the draw goes
    # to a buffer of its own (the reference's outputs.end_ind), it does not overwrite the input
    gt = model.input_buffer("end_ind", (hp.batch_size,))
    gt.copy_(inputs["end_ind"])
    with model.val_mode():
        out = model(dict({k: v.cuda() for k, v in plan_in.items()}, end_ind=gt, len_u=len_u.cuda()), "train")
    torch.cuda.synchronize()
    assert torch.equal(out.end_ind.cpu(), ref["end_ind"]) and out.end_ind.data_ptr() != gt.data_ptr()
    assert torch.equal(gt.cpu(), inputs["end_ind"])


def test_gaussian_decoder_c1():
    from oracle import gcp_model_oracle as O
    hp, sd, model = _build("c1", decoder_distribution="gaussian")
    model.train(True)
    inputs, noise, _ = make_inputs(hp, seed=4, variant="B")
    ref = O.forward(sd, hp, inputs, noise=noise, training_bn=True)
    out = model({k: v.cuda() for k, v in inputs.items()}, "train", noise=noise.cuda())
    torch.cuda.synchronize()
    assert_close(out.tree.bf.images, ref["tree_bf"]["images"], PIX_ATOL, 0, "images")


def test_posterior_forward_c2_full_size():
    """BASELINE.json configs[1]: 64x64, T=80, B=16 (L=7, 127 nodes) against the oracle run on the host cores."""
    from oracle import gcp_model_oracle as O
    hp, sd, model = _build("c2", batch_size=4)
    model.train(True)
    inputs, noise, _ = make_inputs(hp, seed=5, variant="B")
    with torch.no_grad():
        ref = O.forward(sd, hp, inputs, noise=noise, training_bn=True)
    out = model({k: v.cuda() for k, v in inputs.items()}, "train", noise=noise.cuda())
    torch.cuda.synchronize()
    _check_common(hp, model, out, ref, posterior=True)
    # size-independent properties at full size
    lens = out.raw["seq_len"].cpu()
    assert torch.equal(lens, inputs["end_ind"].int() + 1)
    kept = out.raw["kept_idx"].cpu()
    f2n = out.raw["frame2node"].cpu()
    for b in range(hp.batch_size):
        n = int(lens[b])
        assert torch.equal(kept[b, :n], f2n[b, :n])          # k-th kept node (temporal order) is matched to frame k
        assert torch.all(kept[b, :n][1:] > kept[b, :n][:-1])   # strictly increasing depth-first positions
        assert torch.all(kept[b, n:] == -1)


@pytest.mark.parametrize("materialize", [False, True])
@pytest.mark.parametrize("dist", ["discrete_logistic_mixture", "gaussian"])
def test_losses_c1(dist, materialize):
    """ELBO terms and the normalised total (base_gcp.py:264-304) against the oracle.  Stated tolerance: relative 2e-5
    on every loss value (fp32 sums of ~1e5 per-pixel terms), absolute 1e-6 on the per-pixel total."""
    from oracle import gcp_model_oracle as O
    import video_gcp_amd as V
    from video_gcp_amd.model import GCPTreeModel
    hp = V.config("c1", decoder_distribution=dist)
    sd = V.init_params(hp, seed=1, randomize_affine=True)
    model = GCPTreeModel(hp, params=sd, device="cuda", materialize_distr=materialize)
    model.train(True)
    inputs, noise, _ = make_inputs(hp, seed=7, variant="B")
    ref_out = O.forward(sd, hp, inputs, noise=noise, training_bn=True)
    ref_losses, ref_total = O.losses(sd, hp, inputs, ref_out)
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    out = model(dev_in, "train", noise=noise.cuda())
    losses = model.loss(dev_in, out)
    total = model.get_total_loss(dev_in, losses)
    torch.cuda.synchronize()
    for name, (val, w) in ref_losses.items():
        got = float(losses[name].value)
        assert abs(got - float(val)) <= 2e-5 * abs(float(val)) + 1e-6, (name, got, float(val))
        assert losses[name].weight == w
    assert abs(float(total.value) - float(ref_total)) <= 2e-5 * abs(float(ref_total)) + 1e-6
    assert abs(float(losses["nll"].value) - float(ref_losses["dense_img_rec"][0] + ref_losses["kl"][0])) <= 1e-4 * abs(float(losses["nll"].value))


@pytest.mark.parametrize("cfg", [dict(img_sz=64, batch_size=1, max_seq_len=3), dict(img_sz=32, batch_size=3, max_seq_len=33)])
def test_edge_shapes(cfg):
    """smallest tree (T=3: L=2, 3 nodes, one sequence) and a sequence length just past a power of two (T=33: L=6, 63 nodes for
    33 frames, ragged batch): decoded nodes and the total loss against the oracle"""
    from oracle import gcp_model_oracle as O
    hp, sd, model = _build("c1", **cfg)
    model.train(True)
    inputs, noise, _ = make_inputs(hp, seed=2, variant="B")
    ref = O.forward(sd, hp, inputs, noise=noise, training_bn=True)
    _, ref_total = O.losses(sd, hp, inputs, ref)
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    out = model(dev_in, "train", noise=noise.cuda())
    torch.cuda.synchronize()
    _check_common(hp, model, out, ref, posterior=True)
    total = float(model.get_total_loss(dev_in, model.loss(dev_in, out)).value)
    assert abs(total - float(ref_total)) <= 2e-5 * abs(float(ref_total)) + 1e-6


@pytest.mark.parametrize("tree_lstm,lstm_init", [("sum", "mlp"), ("linear", "mlp"), ("split_linear", "zero"), ("sum", "zero"), ("", "mlp")])
def test_tree_lstm_variants_c1(tree_lstm, lstm_init):
    """the other TreeLSTM morphologies and the parameter-free initialiser (tree_lstm.py:11-27,52-74): SumTree adds the parents' hidden
    states, LinTree projects their concatenation with one Linear, ZeroLSTMCellInitializer starts the root's parents from zero states —
    forward and losses against the oracle"""
    from oracle import gcp_model_oracle as O
    hp, sd, model = _build("c1", tree_lstm=tree_lstm, lstm_init=lstm_init)
    # ('' = the non-LSTM subgoal predictor, tree_module.py:45-46,109-110: one Predictor + tanh, no hidden state, no initialiser)
    assert (f"tree_module.tree_modules.0.lstm_initializer.net.input.linear.weight" in sd) == (lstm_init == "mlp" and tree_lstm != "")
    assert ("tree_module.tree_modules.1.subgoal_pred.projection.weight" in sd) == (tree_lstm == "linear")
    assert ("tree_module.tree_modules.1.subgoal_pred.net.head.linear.weight" in sd) == (tree_lstm == "")
    model.train(True)
    inputs, noise, _ = make_inputs(hp, seed=8, variant="B")
    ref = O.forward(sd, hp, inputs, noise=noise, training_bn=True)
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    out = model(dev_in, "train", noise=noise.cuda())
    torch.cuda.synchronize()
    _check_common(hp, model, out, ref, posterior=True)
    ref_losses, ref_total = O.losses(sd, hp, inputs, ref)
    losses = model.loss(dev_in, out)
    assert abs(float(losses["_total"]) - float(ref_total)) <= 3e-5 * abs(float(ref_total))


def test_head32_opt_in_matches_the_default_head(monkeypatch):
    """GCPX_HEAD32=1 routes the mean-only / fused-likelihood head modes to the 32x32x16-tile kernel (csrc/conv3x3_head32.hip, opt-in:
    profiles/r06_head32_study.txt): the same forward, images within f32 rounding, every loss term within 2e-5"""
    import video_gcp_amd as V
    from video_gcp_amd.model import GCPTreeModel
    hp = V.config("c1")
    sd = V.init_params(hp, seed=3, randomize_affine=True)
    inputs, noise, _ = make_inputs(hp, seed=1, variant="B")
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("GCPX_HEAD32", flag)
        m = GCPTreeModel(hp, params=sd, device="cuda")
        assert m.head32 == (flag == "1") and ("dec.head32" in m.pk_split) == (flag == "1")
        m.train(True)
        dev_in = {k: v.cuda() for k, v in inputs.items()}
        out = m(dev_in, "train", noise=noise.cuda())
        losses = m.loss(dev_in, out)
        total = m.get_total_loss(dev_in, losses)
        torch.cuda.synchronize()
        res[flag] = (out.tree.bf.images.float().cpu().clone(), {k: float(losses[k].value) for k in ("dense_img_rec", "kl")}, float(total.value))
    assert_close(res["1"][0], res["0"][0], atol=1e-5, name="images, 32x32 head vs default head")
    for k, v in res["0"][1].items():
        assert abs(res["1"][1][k] - v) <= 2e-5 * max(1.0, abs(v)), (k, res["1"][1][k], v)
    assert abs(res["1"][2] - res["0"][2]) <= 2e-5 * max(1.0, abs(res["0"][2]))
