"""ORACLE (test infrastructure only).  CPU PyTorch fp32 restatement of the flat VRNN baseline `gcp_sequential`:

  SequentialModel.predict_sequence / SequentialRecModule.forward / loss    /root/reference/gcp/prediction/models/sequential.py:33-68,112-114
  BaseGCPModel.forward / run_encoder / run_auxilliary_models               /root/reference/gcp/prediction/models/base_gcp.py:140-262

PARITY UNPINNED: `blox.torch.models.vrnn.VRNNCell` is absent (empty submodule).  This build's spec of the cell
(DESIGN.md): three recurrent nets of the form embed Linear -> n LSTMCells -> out Linear with zero initial state
(lstm_init default 'zero', hyperparameters.py:96):
    p(z_t | x_t, ctx)      = prior_lstm([x_t, e_0, e_g])
    q(z_t | x_{t+1}, ctx)  = inf_lstm([enc(traj_{t+1}), e_0, e_g])
    x_{t+1}                = gen_lstm([x_t, z_t, e_0, e_g])
run for T-1 steps from x_0 = e_0 (sequential.py:49-54); images = cat(I_0, decode(x_1..x_{T-1})) (:56-57).
Variants (experiments/prediction/base_configs/vmpc.py:11-16):
  action_conditioned_pred   a_t = action_encoder(actions[:, t]) (base_gcp.py:211-213) is `more_context` of the cell
                            (sequential.py:45-49): appended to the input of every net at step t
  var_inf = 'deterministic' nz_vae = 0: no prior / inference net, x_{t+1} = gen_lstm([x_t, e_0, e_g, a_t]), KL = 0
  non_goal_conditioned      I_g and the sequence's end frame are zeroed before anything is encoded (base_gcp.py:163-170)
"""
import torch
import torch.nn.functional as F

from . import gcp_model_oracle as O


def _hsp_step(sd, p, hp, state, x):
    nl = hp.n_lstm_layers
    x = F.linear(x, sd[f"{p}.embed.weight"], sd[f"{p}.embed.bias"])
    new = []
    for i in range(nl):
        h, c = state[i]
        gates = F.linear(x, sd[f"{p}.lstm.{i}.weight_ih"], sd[f"{p}.lstm.{i}.bias_ih"]) + \
            F.linear(h, sd[f"{p}.lstm.{i}.weight_hh"], sd[f"{p}.lstm.{i}.bias_hh"])
        gi, gf, gg, go = torch.chunk(gates, 4, 1)
        c = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
        h = torch.sigmoid(go) * torch.tanh(c)
        new.append((h, c))
        x = h
    return new, F.linear(x, sd[f"{p}.out.weight"], sd[f"{p}.out.bias"])


def preprocess(hp, inputs):
    """optional_preprocessing (base_gcp.py:163-170; the reference edits `inputs` in place, so its losses see the same tensors)."""
    inp = dict(inputs)
    if hp.non_goal_conditioned:
        if "traj_seq" in inp:
            inp["traj_seq"] = inp["traj_seq"].clone()
            inp["traj_seq"][torch.arange(inp["traj_seq"].shape[0]), inp["end_ind"]] = 0.0
        inp["I_g"] = torch.zeros_like(inp["I_g"])
    return inp


def forward(sd, hp, inputs, noise=None, sample_prior=False, training_bn=False, phase="train"):
    """noise: eps [B, T-1, nz_vae]; inputs may carry z [B, T-1, nz_vae] (used as the latent directly)."""
    inp = dict(inputs)
    B = inp["I_0"].shape[0]
    T, H, nv = hp.max_seq_len, hp.nz_mid_lstm, hp.nz_vae
    out = {}
    inp = preprocess(hp, inp)
    if "traj_seq" in inp:
        ts = inp["traj_seq"]
        enc, _ = O.encoder(sd, hp, ts.reshape(B * T, *ts.shape[2:]), training_bn)
        inp["enc_traj_seq"] = enc.reshape(B, T, -1)
    e0, skips = O.encoder(sd, hp, inp["I_0"], training_bn)
    eg, _ = O.encoder(sd, hp, inp["I_g"], training_bn)
    e0, eg = e0[:, :, 0, 0], eg[:, :, 0, 0]
    inp["e_0"], inp["e_g"], inp["skips"] = e0, eg, skips
    if hp.regress_length:
        out["seq_len_logits"] = O.predictor(sd, "length_pred.p", hp, e0, eg)
    ctx = [e0, eg] if hp.context_every_step else []
    if hp.action_conditioned_pred:                                    # base_gcp.py:211-213: batch_apply(action_encoder, actions)
        a = inp["actions"]
        enc_act = O.predictor(sd, "action_encoder", hp, a.reshape(-1, a.shape[-1])).reshape(B, a.shape[1], -1)
    zero = lambda: [(torch.zeros(B, H), torch.zeros(B, H)) for _ in range(hp.n_lstm_layers)]
    sp, sq, sg = zero(), zero(), zero()
    p = "dense_rec.lstm.cell"
    x = e0
    xs, pzs, qzs, zs = [], [], [], []
    for t in range(T - 1):
        if hp.action_conditioned_pred:
            ctx = ctx[:2 if hp.context_every_step else 0] + [enc_act[:, t]]
        if hp.deterministic:
            sg, x = _hsp_step(sd, f"{p}.gen_lstm", hp, sg, torch.cat([x] + ctx, 1))
            xs.append(x)
            continue
        sp, pz = _hsp_step(sd, f"{p}.prior_lstm", hp, sp, torch.cat([x] + ctx, 1))
        if "enc_traj_seq" in inp:
            sq, qz = _hsp_step(sd, f"{p}.inf_lstm", hp, sq, torch.cat([inp["enc_traj_seq"][:, t + 1]] + ctx, 1))
        else:
            qz = torch.zeros_like(pz)
        if "z" in inp:
            z = inp["z"][:, t]
        elif sample_prior or "enc_traj_seq" not in inp:
            z = pz[:, :nv] + torch.exp(pz[:, nv:]) * noise[:, t]
        else:
            z = qz[:, :nv] + torch.exp(qz[:, nv:]) * noise[:, t]
        sg, x = _hsp_step(sd, f"{p}.gen_lstm", hp, sg, torch.cat([x, z] + ctx, 1))
        xs.append(x); pzs.append(pz); qzs.append(qz); zs.append(z)
    enc = torch.stack(xs, 1)                                          # encodings [B, T-1, nz]
    out["encodings"] = enc
    if not hp.deterministic:
        out["p_z"], out["q_z"], out["z"] = torch.stack(pzs, 1), torch.stack(qzs, 1), torch.stack(zs, 1)
    dec = O.decode_seq(sd, hp, inp, enc, training_bn)                 # sequential.py:56
    out["distr"] = dec["distr"]
    out["images"] = torch.cat([inp["I_0"][:, None], dec["images"]], 1)   # :57
    end_ind = inp["end_ind"]
    out["pruned_prediction"] = [out["images"][b, :int(end_ind[b]) + 1] for b in range(B)]       # :88-89 ('basic')
    # get_predicted_pruned_seqs (:130-131) / name='encodings' branch (:90-93): e_0 prepended
    full = torch.cat([e0[:, None], enc], 1)
    out["model_enc_seq_list"] = [full[b, :int(end_ind[b]) + 1] for b in range(B)]
    mes = torch.nn.utils.rnn.pad_sequence(out["model_enc_seq_list"], batch_first=True)
    out["model_enc_seq"] = mes
    if hp.run_state_regressor:
        reg_in = mes.detach()                                         # base_gcp.py:253-255 (supervised_decoder=False)
        out["regressed_state"] = O.predictor(sd, "state_regressor", hp, reg_in.reshape(-1, mes.shape[-1])).reshape(B, mes.shape[1], -1)
    # run_auxilliary_models (base_gcp.py:234-262) is BaseGCPModel's: the same branches as in gcp_model_oracle.forward
    if hp.attach_inv_mdl and phase == "train":
        if sample_prior or hp.train_inv_mdl_full_seq or "inv_t0" not in inp:      # base_gcp.py:250 (val_mode sets _inv_mdl_full_seq)
            e1 = mes[:, 1:]
            e0s = inp["enc_traj_seq"][:, :-1][:, :e1.shape[1]] if "enc_traj_seq" in inp else mes[:, :-1]
            a = O.predictor(sd, "inv_mdl.action_pred", hp, torch.cat([e0s, e1], 2).detach().reshape(-1, 2 * hp.nz_enc))
            out["actions"] = a.reshape(B, e1.shape[1], -1)
        else:
            # InverseModel.forward, sampled pair (inverse_mdl.py:136-178); the np.random draws of sample_offsets are inputs
            ar = torch.arange(B)
            t0, t1 = inp["inv_t0"], inp["inv_t1"]
            enc_im0, enc_im1 = inp["enc_traj_seq"][ar, t0].detach(), mes[ar, t1].detach()
            out["actions_sampled"] = O.predictor(sd, "inv_mdl.action_pred", hp, enc_im0, enc_im1)      # [B, n_actions]
            if "actions" in inp:
                out["action_targets"] = inp["actions"][ar, t0]
    if hp.attach_cost_mdl and hp.run_cost_mdl and phase == "train" and "cost_start_idx" in inp and "traj_seq" in inp:
        # CostModel.forward (cost_mdl.py:42-57) with _general_cost's np.random draws fed as inputs (:101-117)
        from . import aux_models_oracle as AX
        ar = torch.arange(B)
        s_idx, e_idx = inp["cost_start_idx"], inp["cost_end_idx"]
        start, end = mes[ar, s_idx].detach(), mes[ar, e_idx].detach()
        out["cost"] = O.predictor(sd, "cost_mdl.cost_pred", hp, torch.cat([start, end], dim=-1))
        out["cost_target"] = torch.as_tensor(AX.euclidean_path_cost(inp["traj_seq"].detach().numpy(), s_idx.numpy(), e_idx.numpy()))
    return out


def losses(sd, hp, inputs, out):
    """decoder.loss on frames 1..T-1 + KL weighted by pad_mask[:, 1:] (sequential.py:60-68)."""
    inputs = preprocess(hp, inputs)
    B, T = inputs["traj_seq"].shape[:2]
    pm = inputs["pad_mask"]
    tgt = inputs["traj_seq"][:, 1:]
    d = out["distr"]
    if hp.decoder_distribution == "gaussian":
        import math
        ls = sd["decoder.log_sigma"]
        err = 0.5 * ((tgt - d) / torch.exp(ls)) ** 2 + ls + 0.5 * math.log(2 * math.pi)
        nll = (err.sum((2, 3, 4)) * pm[:, 1:]).sum() / B
    else:
        nllpp = O.dlm_nll(d.reshape(B * (T - 1), *d.shape[2:]), tgt.reshape(B * (T - 1), *tgt.shape[2:]), hp).reshape(B, T - 1, -1)
        nll = (nllpp.sum(2) * pm[:, 1:]).sum() / B
    nv = hp.nz_vae
    if hp.deterministic:
        kl = torch.zeros(B, 1, 1)                                     # no latent: both distributions are empty
    else:
        mq, lq, mp, lp = out["q_z"][..., :nv], out["q_z"][..., nv:], out["p_z"][..., :nv], out["p_z"][..., nv:]
        kl = lp - lq + (torch.exp(2 * lq) + (mq - mp) ** 2) / (2 * torch.exp(2 * lp)) - 0.5
        kl = torch.clamp(kl, min=hp.free_nats) * pm[:, 1:, None]
    res = {"dense_img_rec": (nll, hp.dense_img_rec_weight), "kl": (kl.sum() / B, hp.kl_weight)}
    if hp.regress_length:
        res["len_pred"] = (F.cross_entropy(out["seq_len_logits"], inputs["end_ind"]), hp.length_pred_weight)
    if "regressed_state" in out and "traj_seq_states" in inputs:      # base_gcp.py:281-286
        rl = out["regressed_state"].shape[1]
        e = (out["regressed_state"] - inputs["traj_seq_states"][:, :rl]) ** 2 * pm[:, :rl, None]
        res["state_regression"] = (e.mean(), 1.0)
    if hp.attach_inv_mdl and "action_targets" in out:                 # base_gcp.py:275-276, inverse_mdl.py:181-191
        from . import aux_models_oracle as AX
        res["action_reconst"] = (AX.l2_loss(out["actions_sampled"], out["action_targets"], 1.0), hp.action_rec_weight)
    if hp.attach_cost_mdl and hp.run_cost_mdl and "cost" in out:      # base_gcp.py:279-280, cost_mdl.py:59-62
        from . import aux_models_oracle as AX
        res["cost_estimation"] = (AX.l2_loss(out["cost"], out["cost_target"]), 1.0)
    total = sum(v * w for v, w in res.values() if w > 0) / float(torch.tensor(inputs["traj_seq"].shape[1:]).prod())
    return res, total


def gradients(sd, hp, inputs, noise):
    """d total_loss / d parameter by torch autograd over this oracle's forward + losses (what `losses.total.value.backward()` of
    train.py:159-161 produces for configuration['model'] = SequentialModel).  Returns ({name: grad}, loss dict, total, out)."""
    names = [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var"))]
    leaf = {k: (sd[k].detach().clone().requires_grad_(True) if k in names else sd[k]) for k in sd}
    out = forward(leaf, hp, inputs, noise=noise, training_bn=True, phase="train")
    res, total = losses(leaf, hp, inputs, out)
    grads = torch.autograd.grad(total, [leaf[k] for k in names], allow_unused=True)
    g = {k: (torch.zeros_like(leaf[k]) if gr is None else gr) for k, gr in zip(names, grads)}
    return g, res, total, out
