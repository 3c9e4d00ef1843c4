"""CPU, world_size 2, gloo: the multi-process path bench.py uses (sharded data, barrier, max-over-ranks timing,
whole-job throughput) and the CEM cost all-gather, without a GPU."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import video_gcp_amd as V
    from video_gcp_amd import dist as D
    from helpers import make_inputs
    r, lr, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    hp = V.config("c1")
    inputs, _, _ = make_inputs(hp, seed=D.shard_seed(100, r))
    D.barrier()
    elapsed = D.max_over_ranks(1.0 + r)                       # rank 1 is the slow one
    costs = D.all_gather_costs(torch.arange(4, dtype=torch.float32) + 10 * r)
    elites = torch.argsort(costs)[:3].tolist()                # every rank must pick the same elites
    value = D.aggregate_throughput(hp.batch_size * hp.max_seq_len, 5, w, elapsed)
    q.put((r, float(inputs["traj_seq"].sum()), elapsed, costs.tolist(), elites, value))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_path():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, e0, c0, el0, v0), (r1, s1, e1, c1, el1, v1) = res
    assert s0 != s1                                  # different data shards
    assert e0 == e1 == 2.0                           # max over ranks
    assert c0 == c1 == [0.0, 1.0, 2.0, 3.0, 10.0, 11.0, 12.0, 13.0]
    assert el0 == el1 == [0, 1, 2]
    assert v0 == v1 == 2 * 2 * 20 * 5 / 2.0          # whole-job frames / slowest-rank time


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from video_gcp_amd import dist as D
    from oracle.radam_oracle import RAdamOracle
    D.init_from_env("gloo")
    g = torch.Generator().manual_seed(0)
    theta = {"p": torch.randn(1000, generator=g)}                 # same initial weights on every rank
    opt = RAdamOracle(lr=1e-2)
    for step in range(3):
        gl = torch.Generator().manual_seed(100 * step + rank)    # every rank: gradient of its own shard
        grad = torch.randn(1000, generator=gl)
        scale = D.all_reduce_sum_(grad)
        opt.step(theta, {"p": grad * scale})
    q.put((rank, theta["p"].clone()))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_all_reduce_keeps_replicas_identical():
    """the training step's only collective: sum all-reduce of the flat gradient, 1/world folded into the update"""
    from oracle.radam_oracle import RAdamOracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert torch.equal(res[0], res[1])
    g = torch.Generator().manual_seed(0)
    theta = {"p": torch.randn(1000, generator=g)}
    opt = RAdamOracle(lr=1e-2)
    for step in range(3):
        gs = [torch.randn(1000, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(2)]
        opt.step(theta, {"p": (gs[0] + gs[1]) * 0.5})
    assert torch.allclose(res[0], theta["p"], atol=1e-6)
