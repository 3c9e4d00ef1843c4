"""Checkpoint save / load in the reference's on-disk format.

Mirrors /root/reference/gcp/prediction/train.py:113-122 (save: {'epoch', 'global_step', 'state_dict', 'optimizer'} to
weights/weights_ep{N}.pth), gcp/prediction/training/checkpoint_handler.py:31-42 (resume 'latest' / epoch / path) and
:45-74,133-143 (load with optional sub-module filter by key prefix — how TestTimeCostModel pulls `cost_mdl.*`,
cost_mdl.py:123-136).  Top-level parameter prefixes equal the reference's; leaf names follow this build's spec
(params.py), so checkpoints trained with the original blox modules need a key map (not derivable: blox is absent).
"""
import glob
import os
import re

import torch


def checkpoint_name(epoch):
    return f"weights_ep{epoch}.pth"


def save_checkpoint(model, folder, epoch, global_step=0, optimizer_state=None):
    os.makedirs(folder, exist_ok=True)
    state = {"epoch": epoch, "global_step": global_step,
             "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
             "optimizer": optimizer_state}
    path = os.path.join(folder, checkpoint_name(epoch))
    torch.save(state, path)
    return path


def get_resume_ckpt_file(ckpt, path):
    """'latest' -> highest epoch in `path`; int / digit string -> that epoch; otherwise a file path."""
    if ckpt == "latest":
        files = glob.glob(os.path.join(path, "weights_ep*.pth"))
        if not files:
            raise FileNotFoundError(f"no checkpoints in {path}")
        return max(files, key=lambda f: int(re.search(r"weights_ep(\d+)\.pth", f).group(1)))
    if isinstance(ckpt, int) or str(ckpt).isdigit():
        return os.path.join(path, checkpoint_name(int(ckpt)))
    return ckpt


def load_weights(weights_file, model, submodule_name=None, strict=True):
    """Returns (global_step, epoch, optimizer_state).  With `submodule_name` only keys under that prefix are loaded
    (prefix kept, since the model holds the sub-module under the same name)."""
    ckpt = torch.load(weights_file, map_location="cpu")
    sd = ckpt["state_dict"]
    if submodule_name is not None:
        sd = {k: v for k, v in sd.items() if k.startswith(submodule_name + ".")}
        if not sd:
            raise ValueError(f"No variable with scope '{submodule_name}' found in checkpoint '{weights_file}'!")
        strict = False
    model.load_state_dict(sd, strict=strict)
    return ckpt.get("global_step", 0), ckpt.get("epoch", 0), ckpt.get("optimizer")
