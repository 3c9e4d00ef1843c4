"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU tests).

The forward shards by sequence: every rank owns `batch_per_gpu` independent sequences (training minibatch shard or CEM
candidate shard) and no collective sits on the data path.  The only exchanges are the timing reduction of
bench.py and (rows "next") the gradient all-reduce / CEM cost all-gather.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, timeout_s=None):
    """Returns (rank, local_rank, world).  Initialises the default process group when WORLD_SIZE > 1.  timeout_s: collective
    timeout (a rank that dies mid-collective then fails the others after that long instead of after the 10-minute default)."""
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
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=timeout_s)
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


def gradient_bucket_ranges(poff, hierarchy_levels, untied_layers):
    """Contiguous [lo, hi) ranges of the flat gradient in the order they become FINAL during the backward pass
    (training.py walks the tree leaves-first): one bucket per untied tree level L-1 .. 0 (about 41 MB each at c2; level 0 also owns
    the existence / distance predictor and the LSTM initialiser, whose gradients are out before its mark), then ONE bucket with
    everything below `tree_modules.0` — conv encoder (shared by the three encoder passes, final last), decoder, temporal encoder,
    latent-space heads: 5 MB at c2, the only part of the exchange (and of the optimizer step) that cannot start before the backward
    ends.  `poff` is the model's {name: (offset, shape)} table; offsets are in floats.  Tied levels accumulate into one weight set: a
    single bucket."""
    end = 0
    starts, spans = {}, []
    for k, (o, shp) in poff.items():
        n = 1
        for d in shp:
            n *= d
        hi = o + (n + 3) // 4 * 4
        end = max(end, hi)
        lvl = int(k.split(".")[2]) if k.startswith("tree_module.tree_modules.") else -1
        spans.append((o, hi, lvl, k))
        if lvl >= 0:
            starts[lvl] = min(starts.get(lvl, o), o)
    if not untied_layers or hierarchy_levels < 2 or len(starts) < hierarchy_levels:
        return [("all", 0, end)]
    out = []
    for l in reversed(range(0, hierarchy_levels)):
        hi = starts[l + 1] if l + 1 in starts else end
        out.append((f"tree{l}", starts[l], hi))
    if starts[0] > 0:
        out.append(("rest", 0, starts[0]))
    # The ranges are also the slices of the early optimizer (training.py marks a slice final at its tree level and updates + re-packs its
    # parameters before the encoder's gradients exist): that is only right if every parameter of level l lies inside [starts[l],
    # starts[l + 1]) and nothing else does — true for the model's own table (params.param_table order), not for an arbitrary order of
    # the `params` dict (a converted reference state_dict).  Anything else gets the single bucket, which is always right.
    level_range = {int(n[4:]): (lo, hi) for n, lo, hi in out if n.startswith("tree")}
    for lo, hi, lvl, k in spans:
        # (a level the model does not have — a converted state_dict of a deeper tree — has no slice: not ok, one bucket)
        rng = level_range.get(lvl)
        ok = (rng is not None and rng[0] <= lo and hi <= rng[1]) if lvl >= 0 else hi <= starts[0]
        if not ok:
            import warnings
            warnings.warn(f"gradient buckets: parameter {k} at [{lo}, {hi}) is outside the range of its slice; using one bucket "
                          "(pass the parameters in params.param_table order to get the per-level exchange)")
            return [("all", 0, end)]
    # the table is laid out in parameter order: the ranges must tile [0, end) exactly
    cover = sorted((lo, hi) for _, lo, hi in out)
    assert cover[0][0] == 0 and cover[-1][1] == end and all(a[1] == b[0] for a, b in zip(cover, cover[1:])), cover
    return out


class GradBuckets:
    """Bucketed data-parallel gradient exchange overlapped with the rest of the backward pass: bucket i is
    all-reduced (sum, in place on its slice of the flat gradient) as soon as the backward reports it final, on a communication
    stream of its own, while the remaining levels are still being differentiated; `finish()` reduces whatever is left, waits for
    everything and returns the 1 / world factor the optimizer kernel folds into its update.  xGMI is point-to-point, so a ring
    all-reduce is per-link bound: a handful of ~40 MB buckets (not hundreds of small ones) keeps each at its bandwidth plateau.
    Works on CPU tensors with gloo (tests) exactly as on the device with RCCL."""

    def __init__(self, flat, ranges, group=None):
        self.flat, self.ranges, self.group = flat, list(ranges), group
        self.works = {}
        self.reduced = set()           # buckets whose all-reduce has been issued in THIS backward pass (reset by begin())
        self.comm_stream = torch.cuda.Stream(device=flat.device) if flat.is_cuda else None
        self.timing = False            # record an event pair on the communication stream around every bucket's collective
        self._events = {}

    def begin(self):
        """Start of a backward pass: nothing of a previous pass may still be pending.  A backward that was not followed by
        `finish()` (gradient inspection, an exception between backward and step, a manual accumulate loop) leaves its works
        behind; the next pass re-computes the gradient and must re-reduce every bucket, so those works are waited for and
        dropped here instead of making `reduce_async` skip the bucket."""
        for w in self.works.values():
            w.wait()
        self.works = {}
        self.reduced = set()

    def reduce_async(self, i, after_streams=()):
        """Start the all-reduce of bucket i.  after_streams: the device streams whose enqueued work produces this bucket."""
        if not dist.is_initialized() or i in self.reduced:
            return                        # once per backward pass: a second finish() must not sum the slice over the ranks again
        self.reduced.add(i)
        _, lo, hi = self.ranges[i]
        view = self.flat[lo:hi]
        if self.comm_stream is not None:
            for s in after_streams:
                self.comm_stream.wait_stream(s)
            with torch.cuda.stream(self.comm_stream):
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm_stream)
                self.works[i] = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.timing:
                    e1.record(self.comm_stream)
                    self._events[i] = (e0, e1)
        else:
            self.works[i] = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def comm_ms(self):
        """{bucket name: milliseconds its all-reduce occupied the communication stream} for the last pass (timing = True; call after a
        device synchronisation).  What a scaling run reads next to its step time: the collectives' own duration, bucket by bucket."""
        return {self.ranges[i][0]: e0.elapsed_time(e1) for i, (e0, e1) in sorted(self._events.items())}

    def finish(self):
        """All buckets reduced and visible to the current stream; returns 1 / world.  Idempotent within one backward pass
        (`begin()` re-arms it): a caller that inspects the reduced gradient and then runs the optimizer step — which calls
        `finish()` itself — gets the sum over the ranks once, not world times."""
        if not dist.is_initialized():
            return 1.0
        if self.comm_stream is not None:
            # whatever was not started during the backward waits for the caller's stream (the backward has been joined onto it)
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.flat.device))
        for i in range(len(self.ranges)):
            self.reduce_async(i)
        for w in self.works.values():
            w.wait()                      # device tensors: the current stream waits for the collective; gloo: blocks
        self.works = {}
        return 1.0 / dist.get_world_size(self.group)
