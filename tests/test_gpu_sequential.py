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
    assert_close(aux.actions, ref["actions"], 1e-4, 1e-3, "actions")
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
