"""-m gpu, needs >= 2 GPUs (skipped on the 1-GPU test box): the bucketed gradient exchange of the training step over a real 2-rank RCCL
group, on the attentive adaptive model (c5s) whose attention key / value projections share a bucket with their tree level and on c1."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["c5s", "c1"])
def test_two_rank_rccl_gradient_exchange(name):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    import socket
    with socket.socket() as s:                      # a free port per case (a fixed one can still be in TIME_WAIT from the previous case)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "rccl_two_rank_check.py"), name], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok:" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
