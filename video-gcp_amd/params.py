"""Parameter table (names, shapes, initialisers) of the gcp_tree model.

Names mirror the module tree the reference builds in
  gcp/prediction/models/base_gcp.py:76-138   (encoder, decoder, inf_encoder, length_pred, inv_mdl, cost_mdl,
                                               state_regressor)
  gcp/prediction/models/tree/tree.py:15-24, untied_layers_tree.py:9-12, tree_module.py:28-52
  gcp/prediction/models/tree/tree_lstm.py:30-41, 52-74
so that a state_dict has the reference's top-level prefixes (`encoder.*`, `decoder.*`,
`tree_module.tree_modules.{i}.*`, `cost_mdl.*`, `inv_mdl.*`, checkpoint_handler.py:133-143).
Leaf names below those prefixes are this build's spec (blox is absent, see hparams.py).

Reference defect D6 (untied_layers_tree.py:17-21): only tree_modules[0].binding is ever used and only
tree_modules[0].lstm_initializer is ever called, so only those are allocated.
"""
import math
from collections import OrderedDict

import torch


def _predictor(tab, prefix, in_dim, out_dim, mid, n_layers):
    """blox-style Predictor / BaseProcessingNet: input(in->mid, act), n x (mid->mid, GroupNorm, act), head(mid->out)."""
    tab[f"{prefix}.input.linear.weight"] = ((mid, in_dim), "xavier")
    tab[f"{prefix}.input.linear.bias"] = ((mid,), "zeros")
    for i in range(n_layers):
        tab[f"{prefix}.pyramid-{i}.linear.weight"] = ((mid, mid), "xavier")
        tab[f"{prefix}.pyramid-{i}.linear.bias"] = ((mid,), "zeros")
        tab[f"{prefix}.pyramid-{i}.norm.weight"] = ((mid,), "ones")
        tab[f"{prefix}.pyramid-{i}.norm.bias"] = ((mid,), "zeros")
    tab[f"{prefix}.head.linear.weight"] = ((out_dim, mid), "xavier")
    tab[f"{prefix}.head.linear.bias"] = ((out_dim,), "zeros")


def _bn(tab, prefix, c):
    tab[f"{prefix}.weight"] = ((c,), "ones")
    tab[f"{prefix}.bias"] = ((c,), "zeros")
    tab[f"{prefix}.running_mean"] = ((c,), "zeros")
    tab[f"{prefix}.running_var"] = ((c,), "ones")


def encoder_layers(hp):
    """[(name, cin, cout, has_norm)] for the stride-2 4x4 convs, then the 4x4 valid head."""
    n = hp.n_conv_layers
    layers = [("input", hp.input_nc, hp.ngf, False)]
    for i in range(n - 3):
        c = hp.ngf * 2 ** i
        layers.append((f"pyramid-{i}", c, 2 * c, True))
    return layers, hp.ngf * 2 ** (n - 3)


def encoder_skip_layers(hp):
    """Indices (into the module list input, pyramid-*, head) whose outputs are skips: every
    `skips_stride`-th module, the last one (head) excluded."""
    n_modules = hp.n_conv_layers - 3 + 2
    if not hp.use_skips:
        return []
    return [i for i in range(n_modules - 1) if i % hp.skips_stride == 0]


def decoder_layers(hp):
    """[(name, c_prev, c_skip, skip_idx, cout)] for the upsample+3x3 blocks (pyramid-*, additional_conv_layer)."""
    n = hp.n_conv_layers
    out = []
    c_prev = hp.ngf * 2 ** (n - 3)
    for i in reversed(range(n - 3)):
        f_out = hp.ngf * 2 ** i
        has_skip = hp.use_skips and (i + 1) % hp.skips_stride == 0
        c_skip = c_prev if has_skip else 0
        assert c_prev == 2 * f_out
        out.append((f"pyramid-{i}", c_prev, c_skip, (i + 1) if has_skip else -1, f_out))
        c_prev = f_out
    has_skip = hp.use_skips
    out.append(("additional_conv_layer", c_prev, c_prev if has_skip else 0, 0 if has_skip else -1, hp.ngf))
    return out


def param_table(hp):
    tab = OrderedDict()
    # ---- encoder -------------------------------------------------------------------------------
    layers, c_top = encoder_layers(hp)
    for name, cin, cout, norm in layers:
        tab[f"encoder.net.{name}.conv.weight"] = ((cout, cin, 4, 4), "xavier")
        tab[f"encoder.net.{name}.conv.bias"] = ((cout,), "zeros")
        if norm:
            _bn(tab, f"encoder.net.{name}.norm", cout)
    tab["encoder.net.head.weight"] = ((hp.nz_enc, c_top, 4, 4), "xavier")
    tab["encoder.net.head.bias"] = ((hp.nz_enc,), "zeros")
    # ---- decoder -------------------------------------------------------------------------------
    tab["decoder.net.input.conv.weight"] = ((hp.nz_enc, c_top, 4, 4), "xavier_t")   # ConvTranspose2d [in,out,kh,kw]
    tab["decoder.net.input.conv.bias"] = ((c_top,), "zeros")
    _bn(tab, "decoder.net.input.norm", c_top)
    for name, c_prev, c_skip, _, cout in decoder_layers(hp):
        tab[f"decoder.net.{name}.conv.weight"] = ((cout, c_prev + c_skip, 3, 3), "xavier")
        tab[f"decoder.net.{name}.conv.bias"] = ((cout,), "zeros")
        _bn(tab, f"decoder.net.{name}.norm", cout)
    tab["decoder.gen_head.conv.weight"] = ((hp.head_channels, hp.ngf, 3, 3), "xavier")
    tab["decoder.gen_head.conv.bias"] = ((hp.head_channels,), "zeros")
    if hp.decoder_distribution == "gaussian" or hp.adaptive:
        tab["decoder.log_sigma"] = ((1,), "zeros")           # adaptive loss reads decoder.log_sigma (adaptive.py:133)
    # ---- temporal inference encoder (ConvSeqEncodingModule, base_gcp.py:130-134) -----------------
    k = hp.conv_inf_enc_kernel_size
    # seq_enc = 'none' (base_gcp.py:131-132): build_temporal_inf_encoder returns Identity — no parameters
    if hp.seq_enc == "conv":
        tab["inf_encoder.net.input.conv.weight"] = ((hp.nz_mid, hp.nz_enc, k), "xavier")
        tab["inf_encoder.net.input.conv.bias"] = ((hp.nz_mid,), "zeros")
        for i in range(hp.conv_inf_enc_layers):
            tab[f"inf_encoder.net.pyramid-{i}.conv.weight"] = ((hp.nz_mid, hp.nz_mid, k), "xavier")
            tab[f"inf_encoder.net.pyramid-{i}.conv.bias"] = ((hp.nz_mid,), "zeros")
            _bn(tab, f"inf_encoder.net.pyramid-{i}.norm", hp.nz_mid)
        tab["inf_encoder.net.head.conv.weight"] = ((hp.nz_enc, hp.nz_mid, k), "xavier")
        tab["inf_encoder.net.head.conv.bias"] = ((hp.nz_enc,), "zeros")
    if hp.attentive_inference:
        # inf_key_encoder = Sequential(ConvSeqEncodingModule, AttnKeyEncodingModule) (base_gcp.py:122-123); it is only
        # read by the attentive posterior, so the balanced model does not allocate it
        q = "inf_key_encoder.0.net"
        if hp.seq_enc == "conv":
            tab[f"{q}.input.conv.weight"] = ((hp.nz_mid, hp.nz_enc, k), "xavier")
            tab[f"{q}.input.conv.bias"] = ((hp.nz_mid,), "zeros")
            for i in range(hp.conv_inf_enc_layers):
                tab[f"{q}.pyramid-{i}.conv.weight"] = ((hp.nz_mid, hp.nz_mid, k), "xavier")
                tab[f"{q}.pyramid-{i}.conv.bias"] = ((hp.nz_mid,), "zeros")
                _bn(tab, f"{q}.pyramid-{i}.norm", hp.nz_mid)
            tab[f"{q}.head.conv.weight"] = ((hp.nz_enc, hp.nz_mid, k), "xavier")
            tab[f"{q}.head.conv.bias"] = ((hp.nz_enc,), "zeros")
        tab["inf_key_encoder.1.linear.weight"] = ((hp.nz_attn_key, hp.nz_enc), "xavier")
        tab["inf_key_encoder.1.linear.bias"] = ((hp.nz_attn_key,), "zeros")
    # ---- heads ---------------------------------------------------------------------------------
    npl = hp.n_processing_layers
    if hp.regress_length:
        _predictor(tab, "length_pred.p", 2 * hp.nz_enc, hp.max_seq_len, hp.nz_mid, npl)
    if hp.attach_state_regressor:
        _predictor(tab, "state_regressor", hp.nz_enc, hp.state_dim, hp.nz_mid, npl)
    if hp.attach_inv_mdl:
        _predictor(tab, "inv_mdl.action_pred", 2 * hp.nz_enc, hp.n_actions, hp.nz_mid, npl)
    if hp.attach_cost_mdl:
        _predictor(tab, "cost_mdl.cost_pred", 2 * hp.nz_enc, 1, hp.nz_mid, npl)
    # ---- tree modules (untied: one per level) -----------------------------------------------------
    H = hp.nz_mid_lstm
    n_mod = hp.hierarchy_levels if hp.untied_layers else 1
    for l in range(n_mod):
        p = f"tree_module.tree_modules.{l}"
        _predictor(tab, f"{p}.prior", 2 * hp.nz_enc, 2 * hp.nz_vae, hp.nz_mid, npl)
        _predictor(tab, f"{p}.inference.q", 3 * hp.nz_enc, 2 * hp.nz_vae, hp.nz_mid, npl)
        if not hp.tree_lstm:
            # tree_lstm = '' (tree_module.py:45-46,109-110): GeneralizedPredictorModel with one head (blox, absent) — this build's spec:
            # a Predictor over [e_l, e_r, z (, e_0, e_g)] whose output goes through tanh; no hidden states, no initialiser
            _predictor(tab, f"{p}.subgoal_pred.net", hp.pred_inp_dim, hp.nz_enc, hp.nz_mid, npl)
        lstm_layers = hp.n_lstm_layers if hp.tree_lstm else 0
        if hp.tree_lstm:
            tab[f"{p}.subgoal_pred.embed.weight"] = ((H, hp.pred_inp_dim), "xavier")
            tab[f"{p}.subgoal_pred.embed.bias"] = ((H,), "zeros")
        for i in range(lstm_layers):
            tab[f"{p}.subgoal_pred.lstm.{i}.weight_ih"] = ((4 * H, H), "lstm")
            tab[f"{p}.subgoal_pred.lstm.{i}.weight_hh"] = ((4 * H, H), "lstm")
            tab[f"{p}.subgoal_pred.lstm.{i}.bias_ih"] = ((4 * H,), "lstm")
            tab[f"{p}.subgoal_pred.lstm.{i}.bias_hh"] = ((4 * H,), "lstm")
        if hp.tree_lstm:
            tab[f"{p}.subgoal_pred.out.weight"] = ((hp.nz_enc, H), "xavier")
            tab[f"{p}.subgoal_pred.out.bias"] = ((hp.nz_enc,), "zeros")
        if hp.tree_lstm == "split_linear":                       # tree_lstm.py:30-41: one Linear(2H -> H) per (layer, h / c) chunk
            for j in range(2 * hp.n_lstm_layers):
                tab[f"{p}.subgoal_pred.projections.{j}.weight"] = ((H, 2 * H), "xavier")
                tab[f"{p}.subgoal_pred.projections.{j}.bias"] = ((H,), "zeros")
        elif hp.tree_lstm == "linear":                           # tree_lstm.py:19-27: one Linear over both parents' whole states
            tab[f"{p}.subgoal_pred.projection.weight"] = ((hp.lstm_state_dim, 2 * hp.lstm_state_dim), "xavier")
            tab[f"{p}.subgoal_pred.projection.bias"] = ((hp.lstm_state_dim,), "zeros")
        if hp.attentive_inference:
            # AttentiveInference.attention (attentive_inference.py:38-45)
            a = f"{p}.inference.attention"
            dk = hp.nz_attn_key
            _predictor(tab, f"{a}.query_net", 2 * hp.nz_enc, dk, hp.nz_mid, npl)
            for i in range(hp.n_attention_layers):
                m = f"{a}.attention_layers.{i}"
                for nm, (o, ii) in dict(q_proj=(dk, dk), k_proj=(dk, dk), v_proj=(hp.nz_enc, hp.nz_enc),
                                        out_proj=(hp.nz_enc, hp.nz_enc)).items():
                    tab[f"{m}.{nm}.weight"] = ((o, ii), "xavier")
                    tab[f"{m}.{nm}.bias"] = ((o,), "zeros")
                tab[f"{m}.temperature"] = ((1,), ("const", hp.attention_temperature))
                _predictor(tab, f"{a}.predictor_layers.{i}", hp.nz_enc, dk, hp.nz_mid, 2)
            tab[f"{a}.out.weight"] = ((hp.nz_enc, hp.nz_enc), "xavier")
            tab[f"{a}.out.bias"] = ((hp.nz_enc,), "zeros")
        if l == 0:
            if hp.tree_lstm and hp.lstm_init == "mlp":           # 'zero' (ZeroLSTMCellInitializer, tree_lstm.py:68-70) has no parameters
                _predictor(tab, f"{p}.lstm_initializer.net", 2 * hp.nz_enc + hp.nz_vae, 2 * hp.lstm_state_dim,
                           hp.init_mlp_mid_sz, hp.init_mlp_layers)
            if hp.adaptive:
                # AdaptiveBinding.build_network (adaptive.py:18-30)
                _predictor(tab, f"{p}.binding.distance_predictor", 2 * hp.nz_enc, 1, hp.nz_mid, npl)
                tab[f"{p}.binding.temp"] = ((1,), ("const", hp.matching_temp))
            else:
                _predictor(tab, f"{p}.binding.existence_predictor", hp.nz_enc, 1, hp.nz_mid, npl)
    return tab


def init_params(hp, seed=0, device="cpu", randomize_affine=False):
    """Random-init weights of the architecture (there are no checkpoints to load here).

    xavier: U(-a, a), a = sqrt(6 / (fan_in + fan_out)); lstm: U(-1/sqrt(H), 1/sqrt(H)) (torch LSTMCell default).
    `randomize_affine=True` perturbs biases / norm affine / running stats so parity tests exercise them.
    """
    return _init_from_table(param_table(hp), hp, seed, device, randomize_affine)


def _init_from_table(table, hp, seed, device, randomize_affine):
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = OrderedDict()
    H = hp.nz_mid_lstm
    for name, (shape, kind) in table.items():
        if kind in ("xavier", "xavier_t"):
            rf = 1
            for s in shape[2:]:
                rf *= s
            a, b = (shape[1], shape[0]) if kind == "xavier" else (shape[0], shape[1])
            bound = math.sqrt(6.0 / ((a + b) * rf))
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif kind == "lstm":
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(H)
        elif kind == "zeros":
            t = torch.zeros(shape)
            if randomize_affine:
                t = (torch.rand(shape, generator=g) * 2 - 1) * 0.1
        elif kind == "ones":
            t = torch.ones(shape)
            if randomize_affine:
                t = 1.0 + (torch.rand(shape, generator=g) * 2 - 1) * 0.25
        elif isinstance(kind, tuple) and kind[0] == "const":
            t = torch.full(shape, float(kind[1]))
        else:
            raise ValueError(kind)
        out[name] = t.to(device=device, dtype=torch.float32).contiguous()
    return out


def n_parameters(hp):
    n = 0
    for name, (shape, _) in param_table(hp).items():
        if name.endswith("running_mean") or name.endswith("running_var"):
            continue
        k = 1
        for s in shape:
            k *= s
        n += k
    return n


# ---------------------------------------------------------------------------------------------------
# gcp_sequential (flat VRNN baseline): /root/reference/gcp/prediction/models/sequential.py:13-131
# ---------------------------------------------------------------------------------------------------
def _hsp(tab, prefix, in_dim, out_dim, H, n_layers):
    """HiddenStatePredictorModel-style cell: embed Linear, n LSTMCells, out Linear (build spec of the absent blox
    VRNNCell's three recurrent nets)."""
    tab[f"{prefix}.embed.weight"] = ((H, in_dim), "xavier")
    tab[f"{prefix}.embed.bias"] = ((H,), "zeros")
    for i in range(n_layers):
        tab[f"{prefix}.lstm.{i}.weight_ih"] = ((4 * H, H), "lstm")
        tab[f"{prefix}.lstm.{i}.weight_hh"] = ((4 * H, H), "lstm")
        tab[f"{prefix}.lstm.{i}.bias_ih"] = ((4 * H,), "lstm")
        tab[f"{prefix}.lstm.{i}.bias_hh"] = ((4 * H,), "lstm")
    tab[f"{prefix}.out.weight"] = ((out_dim, H), "xavier")
    tab[f"{prefix}.out.bias"] = ((out_dim,), "zeros")


def param_table_sequential(hp):
    """Same encoder / decoder / heads as the tree model; the tree modules are replaced by the VRNN cell
    (sequential.py:19-28: VRNNCell(hp, nz_enc, context = 2*nz_enc (+ nz_enc of the encoded action when action_conditioned_pred), ...)).
    var_inf = 'deterministic' (vmpc.py:14-15): no latent, so only the generator net exists.  action_conditioned_pred adds the action
    encoder (sequential.py:108-110: BaseProcessingNet(n_actions, nz_mid, nz_enc, n_processing_layers))."""
    tab = OrderedDict((k, v) for k, v in param_table(hp).items() if not k.startswith("tree_module."))
    ctx = (2 * hp.nz_enc if hp.context_every_step else 0) + (hp.nz_enc if hp.action_conditioned_pred else 0)
    p = "dense_rec.lstm.cell"
    if not hp.deterministic:
        _hsp(tab, f"{p}.prior_lstm", hp.nz_enc + ctx, 2 * hp.nz_vae, hp.nz_mid_lstm, hp.n_lstm_layers)
        _hsp(tab, f"{p}.inf_lstm", hp.nz_enc + ctx, 2 * hp.nz_vae, hp.nz_mid_lstm, hp.n_lstm_layers)
    _hsp(tab, f"{p}.gen_lstm", hp.nz_enc + hp.nz_vae + ctx, hp.nz_enc, hp.nz_mid_lstm, hp.n_lstm_layers)
    if hp.action_conditioned_pred:
        _predictor(tab, "action_encoder", hp.n_actions, hp.nz_enc, hp.nz_mid, hp.n_processing_layers)
    return tab


def init_params_sequential(hp, seed=0, device="cpu", randomize_affine=False):
    return _init_from_table(param_table_sequential(hp), hp, seed, device, randomize_affine)
