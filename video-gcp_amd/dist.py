"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).

The forward shards by sequence: every rank owns `batch_per_gpu` independent sequences (training minibatch shard or CEM
candidate shard) and no collective sits on the data path.  The only exchanges are the timing reduction of
bench.py and (rows "next") the gradient all-reduce / CEM cost all-gather.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Returns (rank, local_rank, world).  Initialises the default process group when WORLD_SIZE > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_seed(base_seed, rank):
    """Every rank draws its own sequences: same weights (same init seed), different data."""
    return base_seed + rank


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds, device="cpu"):
    """Wall time of the slowest rank (the contract's timing rule)."""
    if not dist.is_initialized():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_costs(local_costs):
    """CEM: every rank scores its candidate shard; all ranks need the full cost vector to pick the same elites
    (cem_planner.py:124-135).  local_costs: [n_local] tensor -> [world * n_local]."""
    if not dist.is_initialized():
        return local_costs
    out = [torch.empty_like(local_costs) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local_costs)
    return torch.cat(out)


def aggregate_throughput(units_per_rank_step, steps, world, elapsed_max):
    return world * units_per_rank_step * steps / elapsed_max


def all_reduce_sum_(flat, group=None):
    """Data-parallel gradient exchange: ONE all-reduce (sum) over the flat fp32 gradient vector, in
    place.  Returns the factor 1 / world the optimizer kernel folds into its update, so the mean costs no extra pass."""
    if not dist.is_initialized():
        return 1.0
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / dist.get_world_size(group)
