"""Seeded synthetic batches in the reference's input-dict format (README.md:83-95, data_loader.py:56-81,220-248):
traj_seq ~ U(-1, 1) with frames after end_ind zeroed, pad_mask, I_0 = first frame, I_g = frame end_ind, states / actions
~ N(0, 1).  Used by bench.py, the trainer's --feed_random_data mode and the parity tests (same numbers on CPU and GPU)."""
import torch


def make_inputs(hp, seed=0, variant="B", device="cpu"):
    """variant A: end_ind = T-1 for every sample (the planner's shape, cem_simulator.py:22); B: ragged lengths."""
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


def make_inputs_device(hp, seed, variant, device):
    """Same distribution and structure as make_inputs, drawn on the device with a device generator (a different random stream):
    for the trainer's --feed_random_data style loader, where generating the 63 MB batch on the host and uploading it from pageable
    memory cost 190 ms per batch against a 22 ms training step."""
    g = torch.Generator(device=device).manual_seed(seed)
    B, T, S = hp.batch_size, hp.max_seq_len, hp.img_sz
    traj = torch.rand(B, T, 3, S, S, generator=g, device=device) * 2 - 1
    if variant == "A":
        end_ind = torch.full((B,), T - 1, dtype=torch.long, device=device)
    else:
        end_ind = torch.randint(2, T, (B,), generator=g, device=device)
        end_ind[0] = T - 1
        if B > 1:
            end_ind[1] = 2
    pad_mask = (torch.arange(T, device=device)[None] <= end_ind[:, None]).float()
    traj = traj * pad_mask[:, :, None, None, None]
    return dict(traj_seq=traj, pad_mask=pad_mask, I_0=traj[:, 0].clone(), I_g=traj[torch.arange(B, device=device), end_ind].clone(),
                end_ind=end_ind, start_ind=torch.zeros(B, dtype=torch.long, device=device),
                traj_seq_states=torch.randn(B, T, hp.state_dim, generator=g, device=device),
                actions=torch.randn(B, T - 1, hp.n_actions, generator=g, device=device))
