"""c5 training step with and without the learned matching temperature (gcpx_soft_dtw_dtemp on a side lane)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import video_gcp_amd as V
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs

for learn in (False, True, False, True):
    hp = V.config("c5", learn_matching_temp=learn)
    model = GCPTreeModel(hp, params=V.init_params(hp, seed=1), device="cuda")
    tr = GCPTrainStep(model)
    inputs, noise, _ = make_inputs(hp, seed=2, variant="A")
    dev = {k: v.cuda() for k, v in inputs.items()}
    for _ in range(3):
        tr.step(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step(dev)
    torch.cuda.synchronize()
    print(f"learn_matching_temp={learn}: {(time.perf_counter() - t0) * 100:.3f} ms / step", flush=True)
    del tr, model
