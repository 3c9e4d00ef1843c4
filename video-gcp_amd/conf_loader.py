"""Experiment configuration in the reference's own format.

The reference's trainer loads `<exp_dir>/conf.py` with `imp.load_source` and reads two module-level dicts, `configuration`
(trainer: batch_size, lr, num_epochs, dataset ...) and `model_config` (model hyper-parameters)
(/root/reference/gcp/prediction/training/gcp_builder.py:112-147; e.g. experiments/prediction/25room/gcp_tree/conf.py).  Those files
import `blox.AttrDict`, dataset / logger / cost-function classes and `experiments.prediction.base_configs.*` — none of which is
importable here — but only use them as VALUES of the two dicts.  `load_conf` therefore executes the file with

  * `blox.AttrDict` provided (a dict with attribute access),
  * `experiments.prediction.base_configs.{base_tree, gcp_tree, gcp_adaptive, gcp_sequential, vmpc}` provided with the same entries the
    reference's base configs hold (restated below as data: base_configs/*.py),
  * every other `blox.*` / `gcp.*` / `experiments.*` name resolved to an inert named placeholder,

and maps what it finds onto this build's `GCPHParams` / trainer settings.  Keys without a counterpart on the device path (loggers,
dataset classes, visualisation switches) are returned in `ignored` rather than silently dropped.  A `conf.json`
({"config": "c2", "overrides": {...}, "lr": ...}) is accepted as well."""
import glob
import importlib
import importlib.abc
import importlib.machinery
import json
import os
import sys
import types

from .hparams import GCPHParams, config as named_config


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class _Named(type):
    """placeholder class: `repr` is the dotted name it was imported under; any attribute is another placeholder"""

    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Named(f"{cls.__name__}.{name}", (), {})

    def __repr__(cls):
        return f"<ref {cls.__name__}>"


# experiments/prediction/base_configs/*.py of the reference, as data
_BASE_TREE = dict(one_step_planner="sh_pred", hierarchy_levels=7, binding="loss", seq_enc="conv", tree_lstm="split_linear",
                  lstm_init="mlp", add_weighted_pixel_copy=True, dense_rec_type="node_prob")
_BASE_CONFIGS = {
    "base_tree": (dict(model="TreeModel", logger="HierarchyLogger"), dict(_BASE_TREE)),
    "gcp_tree": (dict(model="TreeModel", logger="HierarchyLogger", metric_pruning_scheme="pruned_dtw"),
                 dict(_BASE_TREE, matching_type="balanced")),
    "gcp_adaptive": (dict(model="TreeModel", logger="HierarchyLogger"),
                     dict(_BASE_TREE, matching_type="dtw_image", learn_matching_temp=False, attentive_inference=True)),
    "gcp_sequential": (dict(model="SequentialModel", logger="HierarchyLogger"),
                       dict(one_step_planner="continuous", dense_rec_type="svg", hierarchy_levels=0, add_weighted_pixel_copy=True)),
    # base_configs/vmpc.py:11-16: gcp_sequential made action-conditioned, deterministic and blind to the goal
    "vmpc": (dict(model="SequentialModel", logger="HierarchyLogger"),
             dict(one_step_planner="continuous", dense_rec_type="svg", hierarchy_levels=0, add_weighted_pixel_copy=True,
                  action_conditioned_pred=True, non_goal_conditioned=True, nz_vae=0, var_inf="deterministic")),
}


class _Module(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name == "AttrDict":
            return AttrDict
        if self.__name__ == "experiments.prediction.base_configs" and name in _BASE_CONFIGS:
            return importlib.import_module(f"{self.__name__}.{name}")     # `from ...base_configs import gcp_tree as base_conf`
        v = _Named(name, (), {})
        setattr(self, name, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("blox", "gcp", "experiments")

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Module(spec.name)
        parts = spec.name.split(".")
        if parts[:3] == ["experiments", "prediction", "base_configs"] and len(parts) == 4 and parts[3] in _BASE_CONFIGS:
            c, mc = _BASE_CONFIGS[parts[3]]
            m.configuration, m.model_config = AttrDict(c), AttrDict(mc)
        return m

    def exec_module(self, module):
        pass


def exec_conf_py(path):
    """Run a reference-style conf.py; returns (configuration, model_config) as plain dicts."""
    finder = _Finder()
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in _Finder.ROOTS}
    for k in saved:
        del sys.modules[k]
    sys.meta_path.insert(0, finder)
    try:
        ns = {"__file__": os.path.abspath(path), "__name__": "conf"}
        with open(path) as f:
            exec(compile(f.read(), path, "exec"), ns)
    finally:
        sys.meta_path.remove(finder)
        for k in [k for k in sys.modules if k.split(".")[0] in _Finder.ROOTS]:
            del sys.modules[k]
        sys.modules.update(saved)
    if "configuration" not in ns or "model_config" not in ns:
        raise ValueError(f"{path}: a configuration file defines `configuration` and `model_config` (gcp_builder.py:136-143)")
    return dict(ns["configuration"]), dict(ns["model_config"])


# trainer keys (gcp_builder.py:_default_hparams) that the device trainer reads
_TRAINER_KEYS = ("batch_size", "lr", "num_epochs", "adam_beta", "epoch_cycles_train", "seed", "gradient_clip", "top_of_100_eval",
                 "metric_pruning_scheme", "optimizer", "momentum")


def model_kind(configuration):
    """configuration['model'] (gcp_builder.py:75: the class the trainer instantiates) -> 'sequential' (SequentialModel, the flat
    VRNN baseline) or 'tree' (TreeModel).  The class arrives as a placeholder named like the reference's class, or as a string."""
    v = configuration.get("model", "TreeModel")
    name = v if isinstance(v, str) else getattr(v, "__name__", repr(v))
    return "sequential" if "sequential" in name.lower() else "tree"
# model_config keys that are settled by what this build implements (checked, not mapped)
_FIXED = {"one_step_planner": ("sh_pred", "continuous"), "binding": ("loss",),
          "dense_rec_type": ("node_prob", "svg", "none", None)}


def hparams_from_conf(configuration, model_config, **over):
    """(GCPHParams, trainer settings, ignored keys).  max_seq_len / img_sz come from the dataset spec in the reference
    (data_loader.py); here they are overrides or the c2 defaults (T=80, 64x64)."""
    fields = set(GCPHParams.__dataclass_fields__)
    kw, ignored = {}, []
    mc = dict(model_config)
    for k, allowed in _FIXED.items():
        if k in mc and mc[k] not in allowed:
            raise ValueError(f"model_config[{k!r}] = {mc[k]!r}: only {allowed} is built")
        mc.pop(k, None)
    if (mc.get("tree_lstm", "split_linear") or "") not in ("split_linear", "linear", "sum", ""):      # tree_lstm.py:52-60; '' / None = the
        raise ValueError(f"model_config['tree_lstm'] = {mc['tree_lstm']!r}: split_linear, linear, sum and '' are built")   # non-LSTM predictor
    if mc.pop("add_weighted_pixel_copy", False):
        # hyperparameters.py:54; base_configs/base_tree.py and gcp_sequential.py set it, every room conf pops it again
        # (25room/gcp_tree/conf.py:43).  A conf that keeps it asks for a decoder with a pixel-copy stream, which is not built: training
        # a different decoder silently is worse than stopping
        raise ValueError("model_config['add_weighted_pixel_copy'] = True: the weighted pixel-copy decoder stream is not built "
                         "(the 25-room / 9-room confs pop the key: `model_config.pop('add_weighted_pixel_copy')`)")
    inv = mc.pop("inv_mdl_params", None) or {}
    if "n_actions" in inv:
        kw["n_actions"] = int(inv["n_actions"])
    if "temp_dist" in inv:
        kw["inv_mdl_temp_dist"] = int(inv["temp_dist"])
    cost = mc.pop("cost_mdl_params", None) or {}
    if cost.get("use_path_dist_cost"):
        raise ValueError("cost_mdl_params.use_path_dist_cost: the fast path needs state sequences; image trajectories use the "
                         "generic cost (cost_mdl.py:81-117)")
    for k, v in mc.items():
        if k in fields:
            kw[k] = v
        else:
            ignored.append(k)
    if "batch_size" in configuration:
        kw["batch_size"] = int(configuration["batch_size"])
    kw.update(over)
    if "hierarchy_levels" in kw and "max_seq_len" in kw and 2 ** int(kw["hierarchy_levels"]) - 1 < int(kw["max_seq_len"]):
        kw.pop("hierarchy_levels")        # a tree too small for the sequence length: fall back to ceil(log2(T)) (train.py:80-81)
    if kw.get("hierarchy_levels", 1) == 0:
        kw.pop("hierarchy_levels")        # the flat model's base config (base_configs/gcp_sequential.py) has no tree
    hp = GCPHParams(**kw)
    trainer = {k: configuration[k] for k in _TRAINER_KEYS if k in configuration}
    trainer["model"] = model_kind(configuration)
    ignored += [k for k in configuration if k not in _TRAINER_KEYS and k != "model"]
    return hp, trainer, ignored


def load_conf(exp_dir, default="c2", name=None, **over):
    """`<exp_dir>/*.py` (the reference's rule, gcp_builder.py:112-118: the one configuration file in the directory) or
    `<exp_dir>/conf.json`.  Returns (GCPHParams, trainer settings dict, ignored keys).  `name`: a named configuration given
    explicitly (the --config flag): it wins over conf.json's "config" entry; `default` is used only when neither names one.
    A reference-style conf.py carries no dataset spec: `max_seq_len` / `img_sz` not given as overrides fall back to the dataclass
    defaults and are reported among the ignored keys as "defaulted:<key>"."""
    if exp_dir and os.path.isdir(exp_dir):
        py = sorted(glob.glob(os.path.join(os.path.abspath(exp_dir), "*.py")))
        if len(py) > 1:
            raise ValueError(f"Multiple configuration files found at {exp_dir}!")
        if py:
            c, mc = exec_conf_py(py[0])
            hp, trainer, ignored = hparams_from_conf(c, mc, **over)
            # the reference takes these from the dataset spec (data_loader.py), which a conf.py does not hold
            ignored += [f"defaulted:{k}" for k in ("max_seq_len", "img_sz") if k not in over and k not in mc]
            return hp, trainer, ignored
        js = os.path.join(exp_dir, "conf.json")
        if os.path.exists(js):
            conf = json.load(open(js))
            ov = dict(conf.get("overrides", {}), **over)
            hp = named_config(name or conf.get("config", default), **ov)
            tr = {k: conf[k] for k in conf if k not in ("config", "overrides")}
            tr["model"] = model_kind(tr)
            return hp, tr, []
    return named_config(name or default, **over), {}, []
