"""Golden vectors for the auxiliary-model training paths, produced by EXECUTING the reference's own functions here.

Run from the repo root:  python tests/golden/make_ref_aux_goldens.py  ->  tests/golden/ref_aux_models.npz
(inputs + expected outputs only — no reference source is copied).

Functions executed (paths relative to /root/reference):
    gcp/prediction/models/auxilliary_models/inverse_mdl.py:  InverseModel.sample_offsets (:84-104), .index_input (:106-114)
    gcp/prediction/models/auxilliary_models/cost_mdl.py:     CostModel._general_cost (:101-117), ._fast_path_dist_cost (:81-99)
    gcp/planning/cem/cost_fcn.py:                            CostFcn.__call__ (:14-21), EuclideanPathLength._compute (:49-54)
The modules import the un-vendored `blox` package and TF-1 `HParams` at module level; an import hook below fabricates
EMPTY placeholder modules / classes for those names so that the files can be loaded — none of the placeholders is executed
except `batchwise_index(t, idx) -> t[arange(B), idx]` (the one-line gather the same helper had in make_ref_dtw_goldens.py).
The methods are called unbound on a plain namespace carrying only the hyper-parameters they read.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_aux_models.npz")


class _Meta(type):
    """placeholder classes answer any attribute with another placeholder class (`from blox.torch.models import base as m; m.BaseModel`)"""

    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Meta(name, (), {})


class _Placeholder(types.ModuleType):
    """module whose every attribute is an empty class (names that are imported but never executed)"""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name == "AttrDict":
            class AttrDict(dict):
                __getattr__ = dict.__getitem__
                __setattr__ = dict.__setitem__
            return AttrDict
        if name == "batchwise_index":
            return lambda t, idx: t[torch.arange(t.shape[0]), idx]
        cls = _Meta(name, (), {})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("blox", "tensorflow")

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Placeholder(spec.name)

    def exec_module(self, module):
        pass


def main():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REF)
    from gcp.prediction.models.auxilliary_models import inverse_mdl, cost_mdl
    from gcp.planning.cem import cost_fcn
    out = {}
    g = torch.Generator().manual_seed(0)

    # ---- InverseModel.sample_offsets / index_input: np.random draw order and the gathers ----
    for case, (end_ind, temp_dist, seed) in enumerate([([5, 2, 3, 7], 1, 0), ([9, 4, 6], 2, 1), ([19, 2, 11, 7, 3, 15], 1, 2)]):
        B = len(end_ind)
        e = torch.tensor(end_ind, dtype=torch.long)
        self_ = types.SimpleNamespace(_hp=types.SimpleNamespace(take_first_tstep=False, temp_dist=temp_dist, device="cpu"))
        np.random.seed(seed)
        t0, t1 = inverse_mdl.InverseModel.sample_offsets(self_, e)
        T = max(end_ind) + 1
        actions = torch.randn(B, T - 1, 2, generator=g)
        enc = torch.randn(B, T, 4, generator=g)
        sel_a = inverse_mdl.InverseModel.index_input(self_, actions, t0)
        sel_agg = inverse_mdl.InverseModel.index_input(self_, actions, t0, aggregate=True, t1=t1)
        sel_e = inverse_mdl.InverseModel.index_input(self_, enc, t1)
        out.update({f"inv{case}_end_ind": e.numpy(), f"inv{case}_cfg": np.array([temp_dist, seed]), f"inv{case}_t0": t0.numpy(),
                    f"inv{case}_t1": t1.numpy(), f"inv{case}_actions": actions.numpy(), f"inv{case}_enc": enc.numpy(),
                    f"inv{case}_sel_actions": sel_a.numpy(), f"inv{case}_sel_agg": sel_agg.numpy(), f"inv{case}_sel_enc": sel_e.numpy()})

    # ---- CostModel._general_cost with the 25-room ground-truth cost (conf.py:35-37: cost_fcn=EuclideanPathLength) ----
    AttrDict = sys.modules["blox"].AttrDict
    for case, (end_ind, shape, seed) in enumerate([([5, 2, 3, 7], (3, 8, 8), 3), ([9, 4, 6], (3, 4, 4), 4), ([6, 3], (2,), 5)]):
        B, T = len(end_ind), max(end_ind) + 1
        e = torch.tensor(end_ind, dtype=torch.long)
        traj = torch.rand(B, T, *shape, generator=g) * 2 - 1
        mes = torch.randn(B, T, 4, generator=g)
        self_ = types.SimpleNamespace(_gt_cost_fcn=cost_fcn.EuclideanPathLength(True))
        np.random.seed(seed)
        # replay the draws to record the indices the call will use (the function does not return them)
        st = np.random.get_state()
        idx = []
        for b in range(B):
            s = np.random.randint(0, end_ind[b], 1)[0]
            idx.append((s, np.random.randint(s + 1, end_ind[b] + 1, 1)[0]))
        np.random.set_state(st)
        start, end, gt = cost_mdl.CostModel._general_cost(self_, AttrDict(end_ind=e, model_enc_seq=mes, traj_seq=traj))
        idx = np.array(idx)
        assert np.array_equal(start.numpy(), mes[torch.arange(B), idx[:, 0]].numpy())
        assert np.array_equal(end.numpy(), mes[torch.arange(B), idx[:, 1]].numpy())
        out.update({f"cost{case}_end_ind": e.numpy(), f"cost{case}_seed": np.array([seed]), f"cost{case}_traj": traj.numpy(),
                    f"cost{case}_mes": mes.numpy(), f"cost{case}_start_idx": idx[:, 0], f"cost{case}_end_idx": idx[:, 1],
                    f"cost{case}_start": start.numpy(), f"cost{case}_end": end.numpy(), f"cost{case}_gt": gt.numpy()})

    # ---- CostModel._fast_path_dist_cost (state sequences [B, T, D]; torch RNG) ----
    for case, (end_ind, D, seed) in enumerate([([5, 2, 3, 7], 2, 6), ([9, 4, 6], 3, 7)]):
        B, T = len(end_ind), max(end_ind) + 1
        e = torch.tensor(end_ind, dtype=torch.long)
        traj = torch.randn(B, T, D, generator=g)
        mes = torch.randn(B, T, 4, generator=g)
        torch.manual_seed(seed)
        u0, u1 = torch.rand((B,)), torch.rand((B,))          # the two draws the function makes, in its order
        torch.manual_seed(seed)
        start, end, gt = cost_mdl.CostModel._fast_path_dist_cost(None, AttrDict(end_ind=e, model_enc_seq=mes, traj_seq=traj))
        out.update({f"fast{case}_end_ind": e.numpy(), f"fast{case}_traj": traj.numpy(), f"fast{case}_mes": mes.numpy(),
                    f"fast{case}_u0": u0.numpy(), f"fast{case}_u1": u1.numpy(), f"fast{case}_start": start.numpy(),
                    f"fast{case}_end": end.numpy(), f"fast{case}_gt": gt.numpy()})
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
