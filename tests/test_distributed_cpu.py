"""N>1 path on CPU: two `gloo` processes (world_size 2) exercise the multi-GPU plumbing that the driver
runs with RCCL — rank/env initialisation, image-pair sharding, DDP gradient all-reduce equivalence
(1 process x B=4 == 2 processes x B=2), the loss all-reduce and bench.py's max-over-ranks timing.
The HIP operators need a GPU, so a small pure-torch module with the UPFlow_net dict contract stands
in for the network here; the Trainer / parallel code under test is the product code."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TinyFlowNet(torch.nn.Module):
    """dict in -> dict out with loss terms, like UPFlow_net.forward (model/upflow.py:370-492)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.c1 = torch.nn.Conv2d(6, 8, 3, padding=1)
        self.c2 = torch.nn.Conv2d(8, 2, 3, padding=1)

    def forward(self, d):
        x = torch.cat([d['im1'], d['im2']], 1)
        flow = self.c2(torch.nn.functional.leaky_relu(self.c1(x), 0.1))
        # per-sample means so that averaging over ranks == averaging over the global batch
        photo = (flow - d['im1'][:, :2]).abs().mean()
        smooth = (flow[:, :, 1:] - flow[:, :, :-1]).abs().mean()
        return {'flow_f_out': flow, 'photo_loss': photo, 'smooth_loss': smooth, 'census_loss': None, 'msd_loss': None}


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_batch():
    g = torch.Generator().manual_seed(7)
    return {'im1': torch.randn(4, 3, 16, 24, generator=g), 'im2': torch.randn(4, 3, 16, 24, generator=g)}


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from upflow_pytorch_amd import parallel
    from upflow_pytorch_amd.train import Trainer
    r, w, _ = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    tr = Trainer(TinyFlowNet(), lr=1e-3, weight_decay=0.0)
    shard = tr.shard(_global_batch())
    assert shard['im1'].shape[0] == 4 // world
    stats = tr.step(shard)
    grads = torch.cat([p.grad.flatten() for p in tr.raw_net.parameters()])
    params = torch.cat([p.detach().flatten() for p in tr.raw_net.parameters()])
    tmax = parallel.max_over_ranks(float(rank + 1))
    q.put((rank, stats, grads.numpy().tolist(), params.numpy().tolist(), tmax, parallel.shard_indices(10, rank, world)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_ddp_two_ranks_match_single_process():
    sys.path.insert(0, ROOT)
    from upflow_pytorch_amd.train import Trainer
    # reference: one process, global batch
    tr = Trainer(TinyFlowNet(), lr=1e-3, weight_decay=0.0, distributed=False)
    ref_stats = tr.step(_global_batch())
    ref_grads = torch.cat([p.grad.flatten() for p in tr.raw_net.parameters()])
    ref_params = torch.cat([p.detach().flatten() for p in tr.raw_net.parameters()])

    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=90) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, s0, g0, p0, t0, i0), (r1, s1, g1, p1, t1, i1) = res
    g0, g1, p0, p1 = (torch.tensor(v) for v in (g0, g1, p0, p1))
    # every rank holds the SAME averaged gradient and identical parameters after the step
    assert torch.allclose(g0, g1, atol=0, rtol=0)
    assert torch.equal(p0, p1)
    # ... equal to the single-process gradient over the global batch (mean of per-shard means, equal shards)
    assert torch.allclose(g0, ref_grads, atol=1e-6, rtol=1e-5)
    assert torch.allclose(p0, ref_params, atol=1e-6, rtol=1e-5)
    assert abs(s0['loss'] - ref_stats['loss']) <= 1e-5 and s0 == s1
    assert t0 == t1 == 2.0                       # max over ranks
    assert sorted(i0 + i1) == list(range(10)) and not set(i0) & set(i1)


def test_single_process_helpers():
    sys.path.insert(0, ROOT)
    from upflow_pytorch_amd import parallel
    from upflow_pytorch_amd.train import synthetic_train_batch, Loss_manager
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        os.environ.pop(k, None)
    assert parallel.init_from_env() == (0, 1, 0)
    assert parallel.max_over_ranks(3.5) == 3.5
    b = synthetic_train_batch(2, crop_hw=(32, 64), raw_hw=(48, 80))
    assert b['im1'].shape == (2, 3, 32, 64) and b['im1_raw'].shape == (2, 3, 48, 80) and b['start'].shape == (2, 2, 1, 1)
    sx, sy = int(b['start'][0, 0, 0, 0]), int(b['start'][0, 1, 0, 0])
    assert torch.equal(b['im1'], b['im1_raw'][:, :, sy:sy + 32, sx:sx + 64])
    total, parts = Loss_manager().compute_loss({'photo_loss': torch.tensor([1.0, 3.0]), 'smooth_loss': torch.tensor(0.5),
                                                'census_loss': None, 'msd_loss': None})
    assert float(total) == 2.5 and set(parts) == {'photo_loss', 'smooth_loss'}
