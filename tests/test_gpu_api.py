"""-m gpu: the model's sub-module handles under the reference's attribute names, the evaluation bindings and the top-of-N
evaluation harness, against the CPU oracle.  Reference call sites: planner_policy.py:225-227 (encoder / inv_mdl.run_single),
tree_dense_rec.py:13-44 (dense_rec.get_sample_with_len, decoder.decode_seq), evaluation_matching.py:123-221,
compute_metrics.py:89-141, train.py:163 (model.step)."""
import numpy as np
import pytest
import torch

from helpers import make_inputs, assert_close

pytestmark = pytest.mark.gpu

LAT_ATOL, LAT_RTOL, PIX_ATOL = 5e-5, 1e-4, 2e-5


@pytest.fixture(scope="module")
def setup():
    import video_gcp_amd as V
    from video_gcp_amd.model import GCPTreeModel
    hp = V.config("c1", batch_size=3)
    sd = V.init_params(hp, seed=1, randomize_affine=True)
    model = GCPTreeModel(hp, params=sd, device="cuda")
    model.eval()
    return hp, sd, model


def test_encoder_inverse_model_and_cost_handles(setup):
    from oracle import gcp_model_oracle as O
    hp, sd, model = setup
    inputs, _, _ = make_inputs(hp, seed=4, variant="A")
    img = inputs["traj_seq"][:, 3]
    enc, skips = model.encoder(img.cuda())
    ref_enc, ref_skips = O.encoder(sd, hp, img, training=False)
    assert enc.shape == ref_enc.shape == (hp.batch_size, hp.nz_enc, 1, 1)
    assert_close(enc[:, :, 0, 0], ref_enc[:, :, 0, 0], LAT_ATOL, LAT_RTOL, "encoder")
    lat = torch.randn(hp.batch_size, hp.nz_enc, generator=torch.Generator().manual_seed(0))
    act = model.inv_mdl.run_single(enc[:, :, 0, 0], lat.cuda())                     # planner_policy.py:225-227
    assert_close(act, O.predictor(sd, "inv_mdl.action_pred", hp, ref_enc[:, :, 0, 0], lat), LAT_ATOL, LAT_RTOL, "run_single")
    c = model.cost_mdl.cost_pred(lat.cuda(), enc[:, :, 0, 0])
    assert_close(c, O.predictor(sd, "cost_mdl.cost_pred", hp, lat, ref_enc[:, :, 0, 0]), LAT_ATOL, LAT_RTOL, "cost_pred")
    model.step()
    model.step()
    assert model.n_steps == 2


def test_decoder_decode_seq_handle(setup):
    from oracle import gcp_model_oracle as O
    hp, sd, model = setup
    inputs, _, _ = make_inputs(hp, seed=5, variant="A")
    B, N = hp.batch_size, 5
    enc = torch.randn(B, N, hp.nz_enc, generator=torch.Generator().manual_seed(1))
    e0, skips = O.encoder(sd, hp, inputs["I_0"], training=False)
    ref = O.decode_seq(sd, hp, dict(skips=skips), enc[..., None, None], training=False)
    got = model.decoder.decode_seq({"I_0": inputs["I_0"].cuda()}, enc.cuda()[..., None, None])
    assert_close(got.images, ref["images"], PIX_ATOL, 0, "decode_seq from I_0")
    _, sk = model.encoder(inputs["I_0"].cuda())
    got2 = model.decoder.decode_seq({"skips": sk}, enc.cuda())
    assert torch.equal(got2.images, got.images)


def test_output_attributes_are_the_reference_names(setup):
    hp, sd, model = setup
    inputs, noise, _ = make_inputs(hp, seed=6, variant="B")
    plan_in = {k: inputs[k].cuda() for k in ("I_0", "I_g", "end_ind", "start_ind")}
    with model.val_mode(pred_length=False):
        out = model(plan_in, "train", noise=noise.cuda())
    pp = out.pruned_prediction                                   # tree.py:62-65: a list of [len_b, 3, H, W]
    assert [p.shape[0] for p in pp] == (inputs["end_ind"] + 1).tolist()
    assert all(torch.equal(a, b) for a, b in zip(pp, model.pruned_prediction(out)))
    aux = model.aux_outputs(out)
    assert torch.equal(out.actions, aux.actions) and torch.equal(out.regressed_state, aux.regressed_state)
    assert torch.equal(out.model_enc_seq, aux.model_enc_seq)
    assert not hasattr(out, "no_such_output")


def test_dense_rec_eval_bindings_match_oracle(setup):
    """get_sample_with_len under the three pruning schemes the reference offers (tree_dense_rec.py:32-40)"""
    from oracle import gcp_model_oracle as O, adaptive_oracle as A
    from video_gcp_amd.handles import DenseRecHandle
    hp, sd, model = setup
    inputs, noise, _ = make_inputs(hp, seed=7, variant="B")
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    ref = O.forward(sd, hp, inputs, noise=noise, training_bn=False)
    out = model(dev_in, "train", noise=noise.cuda())
    torch.cuda.synchronize()
    img_df = out.images_df.cpu()
    B = hp.batch_size
    for b in range(B):
        L = int(inputs["end_ind"][b]) + 1
        # 'basic': the balanced binding's kept nodes
        seq, _ = DenseRecHandle(model).get_sample_with_len(b, L, out, dev_in, "basic")
        assert_close(seq, ref["pruned_prediction"][b], PIX_ATOL, 0, "basic")
        lat, _ = DenseRecHandle(model).get_sample_with_len(b, L, out, dev_in, "basic", name="e_g_prime")
        assert_close(lat, ref["model_enc_seq_list"][b], LAT_ATOL, LAT_RTOL, "basic latents")
        # 'dtw': every node warped onto the ground truth; chosen node per frame bit-exact vs the oracle's get_single_matches
        tgt = inputs["traj_seq"][b, :L]
        cost = ((img_df[b][:, None] - tgt[None]) ** 2).mean(dim=(2, 3, 4)).numpy()
        _, _, _, inds = A.dtw_matches(cost)
        seq, info = DenseRecHandle(model).get_sample_with_len(b, L, out, dev_in, "dtw")
        assert np.array_equal(info.inds[b, :L].cpu().numpy(), inds)
        assert torch.equal(seq.cpu(), img_df[b][inds])
        # 'pruned_dtw': prune, then warp the pruned sequence
        pr = ref["pruned_prediction"][b]
        cost = ((pr[:, None] - tgt[None]) ** 2).mean(dim=(2, 3, 4)).numpy()
        _, _, _, inds = A.dtw_matches(cost)
        seq, info = DenseRecHandle(model).get_sample_with_len(b, L, out, dev_in, "pruned_dtw")
        assert np.array_equal(info.inds[b, :L].cpu().numpy(), inds)
        assert_close(seq, pr[inds], PIX_ATOL, 0, "pruned_dtw")


def test_image_metrics_kernel_matches_oracle():
    from oracle import metrics_oracle as MO
    import video_gcp_amd as V
    from video_gcp_amd.model import GCPTreeModel
    from video_gcp_amd.evaluation import image_metrics
    hp = V.config("c1")
    model = GCPTreeModel(hp, device="cuda")
    g = torch.Generator().manual_seed(3)
    for (B, T, S) in ((2, 6, 32), (1, 4, 64)):
        tgt = torch.rand(B, T, 3, S, S, generator=g) * 2 - 1
        pool = torch.clamp(tgt.reshape(B * T, 3, S, S) + 0.2 * torch.randn(B * T, 3, S, S, generator=g), -1, 1)
        perm = torch.stack([torch.randperm(T, generator=g) + b * T for b in range(B)]).to(torch.int32)
        perm[0, T - 1] = -1                                   # an unmatched frame outside the scored range
        first = torch.ones(B, dtype=torch.int32)
        last = torch.full((B,), T - 1, dtype=torch.int32)
        got = image_metrics(model, pool.cuda(), perm.cuda(), tgt.cuda(), first.cuda(), last.cuda()).cpu()
        for b in range(B):
            idx = perm[b, 1:T - 1].long()
            mse, psnr, ssim = MO.sequence_metrics(pool[idx].numpy(), tgt[b, 1:T - 1].numpy())
            assert abs(float(got[b, 0]) - mse) <= 1e-5 * mse
            assert abs(float(got[b, 1]) - psnr) <= 1e-4
            assert abs(float(got[b, 2]) - ssim) <= 2e-5


def test_top_of_n_evaluation_matches_oracle(setup):
    """Evaluator.eval with top_of_100_eval (compute_metrics.py:132-141), here the best of 3 prior samples with fed noise:
    every sample's (mse, psnr, ssim) and the selected sample against the oracle pipeline (forward -> DTW -> crop -> metrics)"""
    from oracle import gcp_model_oracle as O, adaptive_oracle as A, metrics_oracle as MO
    from video_gcp_amd.evaluation import Evaluator
    hp, sd, model = setup
    inputs, _, _ = make_inputs(hp, seed=8, variant="B")
    inputs["end_ind"] = torch.tensor([hp.max_seq_len - 1, 6, 11])          # sequences long enough to have frames left after the crop
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    n = 3
    noises = torch.randn(n, hp.batch_size, hp.n_nodes, hp.nz_vae, generator=torch.Generator().manual_seed(9))
    ev = Evaluator(model, pruning_scheme="dtw", top_of_100=True, top_of=n)
    with model.val_mode(pred_length=False):
        res = ev.eval(dev_in, noises=noises.cuda())
    want = np.zeros((hp.batch_size, n, 3))
    plan_in = {k: inputs[k] for k in ("I_0", "I_g", "end_ind", "start_ind")}
    for s in range(n):
        ref = O.forward(sd, hp, plan_in, noise=noises[s], sample_prior=True, training_bn=False)
        img_bf = ref["tree_bf"]["images"]
        img_df = O._bf_to_df(img_bf, hp.hierarchy_levels)
        for b in range(hp.batch_size):
            L = int(inputs["end_ind"][b]) + 1
            tgt = inputs["traj_seq"][b, :L]
            cost = ((img_df[b][:, None] - tgt[None]) ** 2).mean(dim=(2, 3, 4)).numpy()
            _, _, _, inds = A.dtw_matches(cost)
            gen = img_df[b][inds]
            want[b, s] = MO.sequence_metrics(gen[1:-1].numpy(), tgt[1:-1].numpy())
    np.testing.assert_allclose(res["mse"], want[..., 0], rtol=2e-4)
    np.testing.assert_allclose(res["psnr"], want[..., 1], atol=2e-3)
    np.testing.assert_allclose(res["ssim"], want[..., 2], atol=1e-4)
    assert np.array_equal(res["best"], np.argmin(want[..., 0], 1))
    summary = ev.dump_metrics()
    assert set(summary) == {"mse", "psnr", "ssim"} and all(len(v) == 3 for v in summary.values())


def test_decoder_nll_and_binding_handles(setup):
    """`self.decoder.nll(estimates, targets, weights)` as BalancedBinding.reconstruction_loss calls it (frame_binding.py:88-99) and the
    binding's integer helpers (frame_binding.py:52-65)"""
    hp, sd, model = setup
    model.train(True)
    inputs, noise, _ = make_inputs(hp, seed=12, variant="B")
    dev_in = {k: v.cuda() for k, v in inputs.items()}
    out = model(dev_in, "train", noise=noise.cuda())
    fused = float(model.loss(dev_in, out).dense_img_rec.value)          # likelihood evaluated inside the head kernel
    assert out.raw["matched_distr_kernel_order"] is None                 # ... so the matched parameters are not kept
    model.fused_head_nll = False                                          # the stored-parameters forward (what the trainer runs)
    model._clear_plans()
    try:
        out = model(dev_in, "train", noise=noise.cuda())
        losses = model.loss(dev_in, out)
        got = model.decoder.nll(out.raw["matched_distr_kernel_order"], dev_in["traj_seq"], dev_in["pad_mask"], log_error_arr=True)
        torch.cuda.synchronize()
    finally:
        model.fused_head_nll = True
        model._clear_plans()
    assert abs(float(got.dense_img_rec.value) - float(losses.dense_img_rec.value)) <= 1e-6 * abs(float(losses.dense_img_rec.value))
    assert abs(fused - float(losses.dense_img_rec.value)) <= 1e-5 * abs(fused)
    assert got.dense_img_rec.error_mat.shape == (hp.batch_size, hp.max_seq_len)
    model.eval()
    lo, hi = model.tree_module.binding.get_init_inds(out)
    assert torch.equal(lo.cpu(), torch.full((hp.batch_size, 1), -1)) and torch.equal(hi.cpu(), inputs["end_ind"][:, None] + 1)
    t = model.tree_module.binding.comp_timestep(torch.tensor([-1, -1, 3]), torch.tensor([0, 4, 8]))
    assert t.tolist() == [0, 1, 5]                                  # (-1 + 0) / 2 truncates to 0 under torch 1.3, not to -1
