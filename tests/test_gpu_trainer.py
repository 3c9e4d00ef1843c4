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
