"""Two ranks on two GPUs: the bucketed RCCL gradient exchange of GCPTrainStep on a real process group (tests/test_gpu_rccl_two_ranks.py
starts it with torch.distributed.run when the box has >= 2 GPUs).  Every rank trains the same weights on its own shard of sequences;
after each backward every named gradient must equal the cross-rank SUM of the ranks' local gradients (computed by a second, un-grouped
trainer on each rank and one plain all-reduce), and after two optimizer steps theta must be bit-identical on both ranks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
import video_gcp_amd as V
from video_gcp_amd import dist as D
from video_gcp_amd.model import GCPTreeModel
from video_gcp_amd.training import GCPTrainStep
from helpers import make_inputs

rank, local_rank, world = D.init_from_env("nccl", timeout_s=300)
assert world == 2
dev = torch.device("cuda", local_rank)
name = sys.argv[1] if len(sys.argv) > 1 else "c5s"
hp = V.config(name)
mk = lambda: GCPTreeModel(hp, params=V.init_params(hp, seed=1, randomize_affine=True), device=dev)
tr = GCPTrainStep(mk(), process_group=dist.group.WORLD)
ref = GCPTrainStep(mk())                                   # same weights, no exchange: the local gradient
for step in range(2):
    inputs, noise, _ = make_inputs(hp, seed=50 + 10 * step + rank, variant="B")
    d = {k: v.to(dev) for k, v in inputs.items()}
    ref.backward(d, noise.to(dev))
    want = ref.grad.clone()
    dist.all_reduce(want)                                  # plain sum over the ranks
    tr.backward(d, noise.to(dev))
    scale = tr.buckets.finish()                            # what optimizer_step does first (idempotent within a backward pass:
                                                           # optimizer_step's own finish() below does not reduce a second time)
    torch.cuda.synchronize()
    numel = {k: v.numel() for k, v in ref.named_grads().items()}
    bad = [k for k, (o, shp) in tr.m._poff.items() if not torch.equal(tr.grad[o:o + numel[k]], want[o:o + numel[k]])]
    assert not bad and scale == 0.5, (step, bad[:8], scale)
    tr.optimizer_step(); ref.grad.copy_(want); ref_scale = 0.5
    # the reference trainer applies the same reduced gradient
    from video_gcp_amd import runtime as rt
    m = ref.m
    st = torch.cuda.current_stream(dev).cuda_stream
    rt.check(m.lib.gcpx_radam_step(m.theta.data_ptr(), ref.grad.data_ptr(), ref.exp_avg.data_ptr(), ref.exp_avg_sq.data_ptr(),
                                   ref.opt_state.data_ptr(), m.theta.numel(), ref.lr, ref.betas[0], ref.betas[1], ref.eps, 0.5, st), "radam")
    m.repack(st)
    torch.cuda.synchronize()
    assert torch.equal(tr.m.theta, ref.m.theta), step
both = [torch.empty_like(tr.m.theta) for _ in range(2)]
dist.all_gather(both, tr.m.theta)
assert torch.equal(both[0], both[1])
if rank == 0:
    print("ok: gradients equal the cross-rank sum, theta identical on both ranks after 2 steps")
dist.destroy_process_group()
