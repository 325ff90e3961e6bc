"""One process per GPU: torch.distributed (backend "nccl" = RCCL over xGMI on ROCm) helpers.

The reference's only multi-GPU construct is single-process nn.DataParallel
(/root/reference/utils/tools.py:140).  Here image pairs shard across ranks: inference needs no
collective at all (replicas); training needs one exchange per step — the all-reduce(mean) of the
3,494,549 fp32 gradients (13.98 MB, three DDP buckets in gradient-completion order) — see DESIGN.md §multi-GPU.
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


DDP_BUCKET_MB = 5


def ddp_wrap(model, device=None, bucket_cap_mb=None):
    """DistributedDataParallel over the 13.98 MB of fp32 gradients in THREE buckets (cap 5 MB -> 5.8 + 5.4 + 2.8 MB; round 6 —
    rounds 1-5 used one 25 MB bucket, whose all-reduce could only start when the LAST gradient existed, with nothing left to overlap).
    DDP orders its buckets by the order in which the gradients became ready in the first step (it rebuilds them once, before the
    Trainer captures the step): here the SGU / context / estimator weights — whose multi-level contractions run first at the end of
    backward (ops.shared_conv_grads) — then the feature pyramid, so the ring all-reduce of a finished bucket runs under the
    remaining weight-gradient contractions, on RCCL's own stream, inside the captured graph.  xGMI is per-link bound: three
    messages of ~5 MB are still far above the size where a ring's latency term matters (8 ranks: 7 x 2 steps of 0.66 MB each).
    bucket_cap_mb: None = UPF_DDP_BUCKET_MB from the environment, else DDP_BUCKET_MB; 25 restores the single bucket."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if bucket_cap_mb is None:
        bucket_cap_mb = float(os.environ.get('UPF_DDP_BUCKET_MB', DDP_BUCKET_MB))
    if device is None and torch.cuda.is_available():
        device = torch.device('cuda', torch.cuda.current_device())
    if device is not None and device.type == 'cuda':
        return DDP(model.to(device), device_ids=[device.index], bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
    return DDP(model, bucket_cap_mb=bucket_cap_mb)


def ddp_bucket_bytes(ddp):
    """Sizes (bytes) of the gradient buckets `ddp` all-reduces per step, in launch order: the rebuilt ones once DDP has rebuilt them
    (after its first steps), the initial assignment before that.  [] for a module that is not DDP-wrapped."""
    if not hasattr(ddp, '_get_ddp_logging_data'):
        return []
    ld = ddp._get_ddp_logging_data()
    txt = ld.get('rebuilt_bucket_sizes') or ld.get('bucket_sizes') or ''
    return [int(t) for t in str(txt).replace(' ', '').split(',') if t]


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
