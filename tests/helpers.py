"""Shared helpers for the parity tests (seeded synthetic inputs per BASELINE.md §2 / SURVEY.md §8d)."""
import numpy as np
import torch


def make_inputs(hp, seed=0, variant="B", device="cpu"):
    """traj_seq ~ U(-1,1); variant A: end_ind = T-1 for every sample, variant B: ragged lengths with padded
    frames zeroed (data_loader.py:239-248); I_0 = first frame, I_g = frame end_ind (data_loader.py:77-78)."""
    g = torch.Generator().manual_seed(seed)
    B, T, S = hp.batch_size, hp.max_seq_len, hp.img_sz
    traj = torch.rand(B, T, 3, S, S, generator=g) * 2 - 1
    if variant == "A":
        end_ind = torch.full((B,), T - 1, dtype=torch.long)
    else:
        end_ind = torch.randint(2, T, (B,), generator=g)
        end_ind[0] = T - 1
        if B > 1:
            end_ind[1] = 2
    pad_mask = (torch.arange(T)[None] <= end_ind[:, None]).float()
    traj = traj * pad_mask[:, :, None, None, None]
    inputs = dict(traj_seq=traj, pad_mask=pad_mask, I_0=traj[:, 0].clone(), I_g=traj[torch.arange(B), end_ind].clone(),
                  end_ind=end_ind, start_ind=torch.zeros(B, dtype=torch.long),
                  traj_seq_states=torch.randn(B, T, hp.state_dim, generator=g),
                  actions=torch.randn(B, T - 1, hp.n_actions, generator=g))
    noise = torch.randn(B, hp.n_nodes, hp.nz_vae, generator=g)
    z = torch.randn(B, hp.n_nodes, hp.nz_vae, generator=g)
    if device != "cpu":
        inputs = {k: v.to(device) for k, v in inputs.items()}
    return inputs, noise, z


def maxdiff(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    d = (a - b).abs()
    return float(d.max()) if d.numel() else 0.0


def assert_close(a, b, atol, rtol=0.0, name=""):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.nonzero(bad.flatten())[0])
        raise AssertionError(f"{name}: {int(bad.sum())}/{bad.numel()} elements differ; max err {float(err.max()):.3e} "
                             f"(atol {atol}, rtol {rtol}); first bad flat index {i}: got {float(a.flatten()[i])} "
                             f"want {float(b.flatten()[i])}")
