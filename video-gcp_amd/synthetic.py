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
    inputs.update(aux_indices(end_ind, torch.rand(4, B, generator=g), hp.inv_mdl_temp_dist))
    if device != "cpu":
        inputs = {k: v.to(device) for k, v in inputs.items()}
    return inputs, noise, z


def aux_indices(end_ind, u, temp_dist=1):
    """Index inputs of the auxiliary models' training paths from four uniform draws per sequence (u [4, B] in [0, 1)):
    inv_t0 ~ U{0 .. end_ind - temp_dist}, inv_t1 = inv_t0 + U{1 .. temp_dist} (inverse_mdl.py:84-104);
    cost_start_idx ~ U{0 .. end_ind - 1}, cost_end_idx ~ U{start + 1 .. end_ind} (cost_mdl.py:105-107).
    Same distributions as the reference's np.random draws (a different random stream)."""
    e = end_ind.to(torch.float64)
    u = u.to(torch.float64)
    fl = lambda x: torch.floor(x).long()
    # (the reference asserts end_ind >= temp_dist, inverse_mdl.py:93; shorter sequences are clamped into [0, end_ind] like the kernel does)
    t0 = torch.clamp(torch.minimum(fl(u[0] * (e - temp_dist + 1)), end_ind - temp_dist), min=0)
    t1 = torch.minimum(t0 + 1 + torch.clamp(fl(u[1] * temp_dist), max=temp_dist - 1), torch.clamp(end_ind, min=0))
    s = torch.minimum(fl(u[2] * e), end_ind - 1)
    en = s + 1 + torch.minimum(fl(u[3] * (e - s.to(torch.float64))), end_ind - s - 1)
    return dict(inv_t0=t0, inv_t1=t1, cost_start_idx=s, cost_end_idx=en)


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
    inputs = dict(traj_seq=traj, pad_mask=pad_mask, I_0=traj[:, 0].clone(), I_g=traj[torch.arange(B, device=device), end_ind].clone(),
                  end_ind=end_ind, start_ind=torch.zeros(B, dtype=torch.long, device=device),
                  traj_seq_states=torch.randn(B, T, hp.state_dim, generator=g, device=device),
                  actions=torch.randn(B, T - 1, hp.n_actions, generator=g, device=device))
    inputs.update(aux_indices(end_ind, torch.rand(4, B, generator=g, device=device), hp.inv_mdl_temp_dist))
    return inputs
