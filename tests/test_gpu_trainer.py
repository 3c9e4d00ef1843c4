"""-m gpu: the training entry point (video-gcp_amd/train.py, counterpart of gcp/prediction/train.py): epoch loop, validation
(prior-sampled prediction + training-mode NLL), checkpoint save in the reference's format, resume."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_trainer_epochs_checkpoint_and_resume(tmp_path):
    from video_gcp_amd.train import ModelTrainer, get_cmd_args
    exp = str(tmp_path / "exp")
    argv = ["--path", exp, "--config", "c1", "--num_epochs", "2", "--batches_per_epoch", "3", "--log_outputs_interval", "1"]
    tr = ModelTrainer(get_cmd_args(argv))
    tr.run()
    assert tr.global_step == 6 and len(tr.log) == 6
    assert all(torch.isfinite(torch.tensor([l for _, l in tr.log])))
    assert os.path.exists(os.path.join(exp, "weights", "weights_ep0.pth")) and os.path.exists(os.path.join(exp, "weights", "weights_ep1.pth"))
    ck = torch.load(os.path.join(exp, "weights", "weights_ep1.pth"), map_location="cpu")
    assert set(ck) == {"epoch", "global_step", "state_dict", "optimizer"} and ck["global_step"] == 6   # train.py:106-115
    theta_end, val_end = tr.model.theta.clone(), tr.last_val
    # resume 'latest' restores weights and optimizer state exactly and continues from the next epoch
    tr2 = ModelTrainer(get_cmd_args(argv + ["--resume", "latest", "--train", "0"]))
    assert tr2.resume("latest") == 2
    assert torch.equal(tr2.model.theta, theta_end)
    assert torch.equal(tr2.trainer.exp_avg, tr.trainer.exp_avg) and float(tr2.trainer.opt_state[0]) == 6.0
    assert abs(tr2.val() - val_end) <= 5e-3 * abs(val_end)      # same weights, fresh Gaussian draws for the latents


def test_trainer_runs_the_flat_vrnn_baseline(tmp_path):
    """configuration['model'] = SequentialModel (gcp_builder.py:75, experiments/prediction/base_configs/gcp_sequential.py): the same
    entry point trains gcp_sequential — model class and training step are chosen from the configuration, checkpoints round-trip"""
    import json
    from video_gcp_amd.sequential import GCPSequentialModel
    from video_gcp_amd.train import ModelTrainer, get_cmd_args
    from video_gcp_amd.training_sequential import SequentialTrainStep
    exp = tmp_path / "seq"
    exp.mkdir()
    (exp / "conf.json").write_text(json.dumps({"config": "c1", "model": "SequentialModel", "lr": 1e-3,
                                               "overrides": {"nz_mid_lstm": 128, "lstm_init": "zero"}}))
    argv = ["--path", str(exp), "--num_epochs", "1", "--batches_per_epoch", "4", "--log_outputs_interval", "1"]
    tr = ModelTrainer(get_cmd_args(argv))
    assert isinstance(tr.model, GCPSequentialModel) and isinstance(tr.trainer, SequentialTrainStep)
    tr.run()
    losses = [l for _, l in tr.log]
    assert tr.global_step == 4 and all(torch.isfinite(torch.tensor(losses)))
    assert os.path.exists(os.path.join(str(exp), "weights", "weights_ep0.pth"))
    tr2 = ModelTrainer(get_cmd_args(argv + ["--resume", "latest", "--train", "0"]))
    assert tr2.resume("latest") == 1 and torch.equal(tr2.model.theta, tr.model.theta)
