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


def test_comm_exports_with_one_rank():
    """gcpx_comm_* (the collectives behind the C boundary, include/gcpx.h): a one-rank communicator on the box's GPU — unique id, init,
    an in-place all-reduce and an all-gather on a stream of the caller, destroy.  (In a subprocess: RCCL initialises its own state.)"""
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
import torch
from video_gcp_amd import runtime as rt
lib = rt.load_library()
torch.cuda.set_device(0)
uid = C.create_string_buffer(128)
rt.check(lib.gcpx_comm_unique_id(uid), "unique_id")
comm = C.c_void_p()
rt.check(lib.gcpx_comm_init(C.byref(comm), 0, 1, uid), "comm_init")
st = torch.cuda.Stream()
x = torch.arange(1000, dtype=torch.float32, device="cuda")
want = x.clone()
y = torch.zeros(1000, device="cuda")
st.wait_stream(torch.cuda.current_stream())
rt.check(lib.gcpx_comm_allreduce(comm, x.data_ptr(), x.numel(), st.cuda_stream), "allreduce")
rt.check(lib.gcpx_comm_allgather(comm, x.data_ptr(), y.data_ptr(), x.numel(), st.cuda_stream), "allgather")
st.synchronize()
assert torch.equal(x, want) and torch.equal(y, want)
rt.check(lib.gcpx_comm_destroy(comm), "destroy")
print("comm ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "comm ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
