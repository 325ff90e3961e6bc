"""One process per GPU: torch.distributed (backend "nccl" = RCCL over xGMI on ROCm) helpers.

The reference's only multi-GPU construct is single-process nn.DataParallel
(/root/reference/utils/tools.py:140).  Here image pairs shard across ranks: inference needs no
collective at all (replicas); training needs one exchange per step — the all-reduce(mean) of the
3,494,549 fp32 gradients (13.98 MB, one DDP bucket) — see DESIGN.md §multi-GPU.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's RANK/WORLD_SIZE/MASTER_* env.
    Returns (rank, world_size, local_rank).  No-op (0, 1, 0) outside torchrun."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def ddp_wrap(model, device=None):
    """DistributedDataParallel with one 25 MB bucket: the whole 13.98 MB gradient fits one ring
    all-reduce (xGMI is per-link bound, so fewer and larger messages win)."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if device is None and torch.cuda.is_available():
        device = torch.device('cuda', torch.cuda.current_device())
    if device is not None and device.type == 'cuda':
        return DDP(model.to(device), device_ids=[device.index], bucket_cap_mb=25, gradient_as_bucket_view=True)
    return DDP(model, bucket_cap_mb=25)


def shard_indices(n_items, rank, world):
    """Disjoint, contiguous-strided shard of a dataset (DistributedSampler-style, no shuffling)."""
    return list(range(rank, n_items, world))


def max_over_ranks(value, device=None):
    """MAX-reduce a python float over ranks (used for step timing)."""
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device=None):
    """Every rank's python float, in rank order, on every rank (one all_gather of a scalar) — bench.py reports the spread of the
    per-rank step times beside their maximum, so that a slow rank in a scaling run is visible from the one JSON line."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else 'cpu')
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


class no_gc_during_capture(object):
    """with no_gc_during_capture(): ... — python's cyclic garbage collector switched off for the duration of a hipGraph capture
    (after one explicit collection).  A collection that runs INSIDE a capture can free objects whose destructors issue HIP calls that
    are illegal while a stream captures in the global mode — an older Trainer's or GraphedInference's hipGraph, caught by a
    reference cycle, is destroyed (hipGraphExecDestroy) — and the error, thrown from a destructor, aborts the process.  Seen twice in
    ~20 runs of tests/test_hip_train.py (round 5; the backward thread of a capture, "Garbage-collecting" on top of the stack)."""

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False
