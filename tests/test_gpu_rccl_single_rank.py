"""-m gpu: the multi-GPU code paths of bench.py / training / planning on a 1-rank RCCL ("nccl") process group — real RCCL API usage
(init with device_id, barrier, MAX reduction of the step time, gradient all-reduce ordered against the model's streams, cost
all-gather) on the one GPU the test box has.  The data-parallel arithmetic itself is covered by the gloo world-size-2 CPU tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_paths_with_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_single_rank_check.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()               # (RCCL prints its version banner after the script's last line)
    assert "max |theta diff| = 0.0" in r.stdout and "ok" in lines, r.stdout[-2000:]
