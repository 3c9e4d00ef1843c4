"""gcp_sequential training step at the c2 shapes: `GCPX_SEQ_CHAINS=serial|overlap python tools/r06/time_seq_train.py [width]`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.sequential import GCPSequentialModel
from video_gcp_amd.training_sequential import SequentialTrainStep
from helpers import make_inputs

over = {"nz_mid_lstm": int(sys.argv[1])} if len(sys.argv) > 1 else {}
hp = V.config("c2", **over)
tr = SequentialTrainStep(GCPSequentialModel(hp, device="cuda"))
inputs, noise, _ = make_inputs(hp, seed=3, variant="A")
dev = {k: v.cuda() for k, v in inputs.items()}
for _ in range(3):
    tr.step(dev)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); th = 0.0
    for _ in range(10):
        h0 = time.perf_counter(); tr.step(dev); th += time.perf_counter() - h0
    torch.cuda.synchronize()
    print(f"GCPX_SEQ_CHAINS={os.environ.get('GCPX_SEQ_CHAINS', 'lockstep')} width {hp.nz_mid_lstm}: {(time.perf_counter() - t0) * 100:.2f} ms / step (host {th * 100:.2f})", flush=True)
