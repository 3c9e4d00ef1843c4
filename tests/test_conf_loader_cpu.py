"""Experiment configuration in the reference's format (gcp_builder.py:112-147): a conf.py with `configuration` / `model_config`
that imports blox / gcp / experiments names, loaded without those packages."""
import os
import textwrap

import pytest

from video_gcp_amd import conf_loader as CL

CONF = textwrap.dedent('''
    import os
    from blox import AttrDict
    from gcp.datasets.data_loader import MazeTopRenderedGlobalSplitVarLenVideoDataset
    from gcp.planning.cem.cost_fcn import EuclideanPathLength
    current_dir = os.path.dirname(os.path.realpath(__file__))
    from experiments.prediction.base_configs import gcp_tree as base_conf

    configuration = AttrDict(base_conf.configuration)
    configuration.update({'dataset_name': 'nav_25rooms', 'batch_size': 16, 'lr': 2e-4, 'epoch_cycles_train': 2, 'n_rooms': 25})
    model_config = AttrDict(base_conf.model_config)
    model_config.update({'untied_layers': True, 'hierarchy_levels': 8, 'ngf': 16, 'nz_mid_lstm': 512, 'n_lstm_layers': 3,
                         'nz_mid': 128, 'nz_enc': 128, 'nz_vae': 256, 'regress_length': True, 'attach_state_regressor': True,
                         'attach_cost_mdl': True, 'cost_mdl_params': AttrDict(cost_fcn=EuclideanPathLength),
                         'attach_inv_mdl': True, 'inv_mdl_params': AttrDict(n_actions=2, use_convs=False, build_encoder=False),
                         'decoder_distribution': 'discrete_logistic_mixture'})
    model_config.pop("add_weighted_pixel_copy")
''')


def test_reference_style_conf_py(tmp_path):
    (tmp_path / "conf.py").write_text(CONF)
    hp, trainer, ignored = CL.load_conf(str(tmp_path), max_seq_len=200)
    assert trainer["lr"] == 2e-4 and trainer["batch_size"] == 16 and trainer["metric_pruning_scheme"] == "pruned_dtw"
    assert hp.batch_size == 16 and hp.hierarchy_levels == 8 and hp.n_nodes == 255 and hp.matching_type == "balanced"
    assert hp.nz_vae == 256 and hp.attach_cost_mdl and hp.attach_inv_mdl and hp.n_actions == 2
    assert hp.decoder_distribution == "discrete_logistic_mixture" and hp.untied_layers
    assert "dataset_name" in ignored and trainer["model"] == "tree"          # configuration['model'] = TreeModel (gcp_builder.py:75)
    # nothing leaks into the interpreter's module table
    import sys
    assert "blox" not in sys.modules and "experiments" not in sys.modules


def test_sequential_conf_selects_the_flat_model(tmp_path):
    """experiments/prediction/25room/gcp_sequential/conf.py: base config gcp_sequential (configuration['model'] = SequentialModel,
    hierarchy_levels 0, add_weighted_pixel_copy popped), 1024-wide LSTMs, free_nats 1, KL burn-in (not built: reported)"""
    (tmp_path / "conf.py").write_text(textwrap.dedent('''
        from blox import AttrDict
        from gcp.planning.cem.cost_fcn import EuclideanPathLength
        from experiments.prediction.base_configs import gcp_sequential as base_conf
        configuration = AttrDict(base_conf.configuration)
        configuration.update({'dataset_name': 'nav_25rooms', 'batch_size': 16, 'lr': 2e-4, 'metric_pruning_scheme': 'basic'})
        model_config = AttrDict(base_conf.model_config)
        model_config.update({'kl_weight_burn_in': 1e4, 'free_nats': 1, 'ngf': 16, 'nz_mid_lstm': 1024, 'n_lstm_layers': 3, 'nz_mid': 128,
                             'nz_enc': 128, 'nz_vae': 256, 'regress_length': True, 'attach_state_regressor': True,
                             'decoder_distribution': 'discrete_logistic_mixture'})
        model_config.pop("add_weighted_pixel_copy")
    '''))
    hp, trainer, ignored = CL.load_conf(str(tmp_path), max_seq_len=80, img_sz=64)
    assert trainer["model"] == "sequential" and trainer["lr"] == 2e-4
    assert hp.nz_mid_lstm == 1024 and hp.free_nats == 1 and hp.hierarchy_levels == 7 and hp.max_seq_len == 80
    assert hp.kl_weight_burn_in == 1e4 and "kl_weight_burn_in" not in ignored      # base_gcp.py:121-128


def test_vmpc_base_config_selects_the_action_conditioned_flat_model(tmp_path):
    """experiments/prediction/base_configs/vmpc.py:11-16 on top of gcp_sequential: action-conditioned, not goal-conditioned, no latent"""
    (tmp_path / "conf.py").write_text(textwrap.dedent('''
        from blox import AttrDict
        from experiments.prediction.base_configs import vmpc as base_conf
        configuration = AttrDict(base_conf.configuration)
        configuration.update({'batch_size': 16, 'lr': 2e-4})
        model_config = AttrDict(base_conf.model_config)
        model_config.update({'nz_mid_lstm': 512, 'inv_mdl_params': AttrDict(n_actions=2)})
        model_config.pop("add_weighted_pixel_copy")
    '''))
    hp, trainer, ignored = CL.load_conf(str(tmp_path), max_seq_len=40, img_sz=32)
    assert trainer["model"] == "sequential"
    assert hp.action_conditioned_pred and hp.non_goal_conditioned and hp.deterministic and hp.nz_vae == 0 and hp.n_actions == 2
    # the sequential parameter table of such a model: one recurrent net and the action encoder
    from video_gcp_amd.params import param_table_sequential
    tab = param_table_sequential(hp)
    assert not any("prior_lstm" in k or "inf_lstm" in k for k in tab)
    assert tab["action_encoder.input.linear.weight"][0] == (hp.nz_mid, hp.n_actions)
    assert tab["dense_rec.lstm.cell.gen_lstm.embed.weight"][0] == (hp.nz_mid_lstm, 4 * hp.nz_enc)      # x, e_0, e_g, encoded action
    import pytest
    with pytest.raises(AssertionError):
        CL.GCPHParams(var_inf="deterministic")            # a deterministic predictor has no latent (vmpc.py:14-15)


def test_adaptive_base_config_and_errors(tmp_path):
    (tmp_path / "conf.py").write_text(textwrap.dedent('''
        from blox import AttrDict
        from experiments.prediction.base_configs import gcp_adaptive as base_conf
        configuration = AttrDict(base_conf.configuration)
        configuration.update({'batch_size': 8, 'lr': 1e-3})
        model_config = AttrDict(base_conf.model_config)
        model_config.update({'hierarchy_levels': 8})
        model_config.pop("add_weighted_pixel_copy")
    '''))
    hp, trainer, _ = CL.load_conf(str(tmp_path), max_seq_len=200)
    assert hp.adaptive and hp.attentive_inference and hp.batch_size == 8
    assert hp.learn_matching_temp is False                              # base_configs/gcp_adaptive.py:9
    # a DTW conf that does not mention it gets hyperparameters.py:132's default, one that sets it keeps its value
    (tmp_path / "conf.py").write_text("configuration = {}\nmodel_config = {'matching_type': 'dtw_image', 'attentive_inference': True}\n")
    assert CL.load_conf(str(tmp_path), max_seq_len=12, img_sz=32)[0].learn_matching_temp is True
    (tmp_path / "conf.py").write_text("configuration = {}\nmodel_config = {'matching_type': 'dtw_image', 'learn_matching_temp': False}\n")
    assert CL.load_conf(str(tmp_path), max_seq_len=12, img_sz=32)[0].learn_matching_temp is False
    (tmp_path / "conf.py").write_text("configuration = {}\nmodel_config = {'tree_lstm': 'sum', 'lstm_init': 'zero'}\n")
    hp, _, _ = CL.load_conf(str(tmp_path))
    assert hp.tree_lstm == "sum" and hp.lstm_init == "zero"             # tree_lstm.py:52-74: sum / linear / split_linear, zero / mlp
    (tmp_path / "conf.py").write_text("configuration = {}\nmodel_config = {'tree_lstm': ''}\n")     # the non-LSTM subgoal predictor
    hp, _, _ = CL.load_conf(str(tmp_path))                              # (tree_module.py:45-46)
    assert hp.tree_lstm == ""
    (tmp_path / "conf.py").write_text("configuration = {}\nmodel_config = {'tree_lstm': 'gru'}\n")
    with pytest.raises(ValueError):
        CL.load_conf(str(tmp_path))
    # seq_enc (base_gcp.py:130-138): 'conv' and 'none' (Identity) load; the parameter table of 'none' has no temporal encoder
    (tmp_path / "conf.py").write_text("configuration = {}\nmodel_config = {'seq_enc': 'none'}\n")
    hp, _, _ = CL.load_conf(str(tmp_path))
    assert hp.seq_enc == "none"
    from video_gcp_amd.params import param_table
    assert not any(k.startswith("inf_encoder.") for k in param_table(hp)) and any(k.startswith("inf_encoder.") for k in param_table(CL.load_conf(None, default="c1")[0]))
    (tmp_path / "conf.py").write_text("x = 1\n")
    with pytest.raises(ValueError):
        CL.load_conf(str(tmp_path))


def test_conf_json_and_default(tmp_path):
    (tmp_path / "conf.json").write_text('{"config": "c1", "overrides": {"batch_size": 3}, "lr": 0.01}')
    hp, trainer, _ = CL.load_conf(str(tmp_path))
    assert hp.batch_size == 3 and hp.img_sz == 32 and trainer["lr"] == 0.01
    hp, trainer, _ = CL.load_conf(None, default="c2")
    assert hp.max_seq_len == 80 and trainer == {}


@pytest.mark.skipif(not os.path.isdir("/root/reference/experiments/prediction/25room/gcp_tree"), reason="reference tree not present")
def test_the_reference_25room_conf_itself():
    hp, trainer, ignored = CL.load_conf("/root/reference/experiments/prediction/25room/gcp_tree", max_seq_len=200)
    assert trainer["lr"] == 2e-4 and hp.batch_size == 16 and hp.hierarchy_levels == 8 and hp.attach_cost_mdl


def test_a_conf_that_keeps_the_pixel_copy_stream_is_refused(tmp_path):
    """base_configs/base_tree.py sets add_weighted_pixel_copy=True (hyperparameters.py:54) and the room confs pop it; a conf that
    keeps it asks for a decoder this build does not have and must not train a different one silently"""
    (tmp_path / "conf.py").write_text(CONF.replace('model_config.pop("add_weighted_pixel_copy")', ''))
    with pytest.raises(ValueError, match="add_weighted_pixel_copy"):
        CL.load_conf(str(tmp_path), max_seq_len=200)
