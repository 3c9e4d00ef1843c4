"""video-gcp_amd: MI355X-native hot path of the goal-conditioned hierarchical predictor (gcp_tree).

Host side in Python (mirrors the reference's gcp.prediction model API); compute in hand-written HIP kernels
for gfx950 behind a C-ABI shared library (include/gcpx.h), loaded with ctypes.  PyTorch-ROCm is used only for
device memory, streams and torch.distributed.
"""
from .hparams import GCPHParams, config  # noqa: F401
from .params import param_table, init_params, n_parameters, param_table_sequential, init_params_sequential  # noqa: F401
