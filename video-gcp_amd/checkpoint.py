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
    # (beyond the reference's four keys: where the model's in-plan noise stream stands, so that a resumed run continues the sequence)
    rng = model.rng_state() if hasattr(model, "rng_state") else None
    if rng is not None:
        state["rng_state"] = rng
    path = os.path.join(folder, checkpoint_name(epoch))
    torch.save(state, path)
    return path


class NoCheckpointsException(Exception):
    """checkpoint_handler.py:10-11: the folder holds no checkpoint (ModelTrainer.resume then starts from epoch 0)"""


def get_resume_ckpt_file(ckpt, path):
    """checkpoint_handler.py:31-42: 'latest' -> highest epoch in `path` (NoCheckpointsException when there is none); an epoch
    number -> weights_ep{N}.pth; any other name gets '.pth' appended when missing; the result is always joined with `path`."""
    if ckpt == "latest":
        files = glob.glob(os.path.join(os.path.abspath(path), "*.pth"))
        epochs = [int(m.group(1)) for m in (re.fullmatch(r"weights_ep(\d+)\.pth", os.path.basename(f)) for f in files) if m]
        if not epochs:
            print(f"Warning: No checkpoints found at {path}!")
            raise NoCheckpointsException
        name = checkpoint_name(max(epochs))
    elif isinstance(ckpt, int) or str(ckpt).isdigit():
        name = checkpoint_name(int(ckpt))
    elif ".pth" not in str(ckpt):
        name = str(ckpt) + ".pth"
    else:
        name = str(ckpt)
    return os.path.join(path, name)


def resumed_rng_state(stored, own_key, world):
    """(key, offset) a model continues its in-plan noise stream from after loading a checkpoint that stored `stored` = (key, offset).
    One process: the stored stream, as it was.  A process group: the checkpoint is rank 0's (train.py saves on rank 0 only), so every
    rank keeps its own key — per-rank seeds give per-rank keys — and takes the offset, the position in the sequence."""
    key, offset = int(stored[0]), int(stored[1])
    return (key, offset) if world <= 1 else (int(own_key), offset)


def load_weights(weights_file, model, submodule_name=None, strict=True):
    """Returns (global_step, epoch, optimizer_state).  checkpoint_handler.py:45-74,133-143: with `submodule_name` only keys under
    that prefix are loaded.  (The reference strips the prefix because it loads INTO the sub-module; this model holds the
    sub-module under the same name, so the prefix stays.)  strict: every parameter of the model — of that sub-module, when one is
    named — must be in the checkpoint, and no key of the (filtered) checkpoint may be unknown to the model."""
    if not os.path.isfile(weights_file):
        raise ValueError("Could not find checkpoint file in {}!".format(weights_file))
    ckpt = torch.load(weights_file, map_location="cpu")
    sd = ckpt["state_dict"]
    own = set(model.state_dict())
    if submodule_name is not None:
        sd = {k: v for k, v in sd.items() if k.startswith(submodule_name + ".")}
        if not sd:
            raise ValueError("Did not find submodule {} in checkpoint!".format(submodule_name))
        own = {k for k in own if k.startswith(submodule_name + ".")}
    if strict:
        missing, unexpected = sorted(own - set(sd)), sorted(set(sd) - set(model.state_dict()))
        if missing or unexpected:
            raise KeyError(f"checkpoint {weights_file}: missing keys {missing[:5]}{'...' if len(missing) > 5 else ''}, "
                           f"unexpected keys {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
    model.load_state_dict(sd, strict=False)
    if submodule_name is None and ckpt.get("rng_state") is not None and hasattr(model, "set_rng_state"):
        model.set_rng_state(ckpt["rng_state"])
    return ckpt.get("global_step", 0), ckpt.get("epoch", 0), ckpt.get("optimizer")
