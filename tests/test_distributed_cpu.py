"""N>1 path on CPU: two `gloo` processes (world_size 2) exercise the multi-GPU plumbing that the driver
runs with RCCL — rank/env initialisation, image-pair sharding, DDP gradient all-reduce equivalence
(1 process x B=4 == 2 processes x B=2), the loss all-reduce and bench.py's max-over-ranks timing.
Two layers: (1) a small pure-torch module with the UPFlow_net dict contract (pure DDP arithmetic); (2) the REAL
UPFlow_net training forward + Trainer + DDP bucket, with the HIP operator entry points replaced by the oracle's
differentiable restatements (tests/_ops_cpu_stub.py — the product itself has no CPU path); (3) bench.py's own
rank launcher (`python bench.py --gpus 2` with no torchrun around it)."""
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


# ------------------------------------------------------------------------------------------------ the real network
NET_FLAGS = {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False, 'norm_moments_across_images': False,
             'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}


def _real_net():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import _ops_cpu_stub
    _ops_cpu_stub.install()
    from upflow_pytorch_amd import synthetic
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    conf = UPFlow_net.config()
    d = dict(NET_FLAGS)
    d.update(synthetic.TRAIN_FLAGS)
    conf.update(d, verbose=False)
    net = conf()
    net.load_state_dict(synthetic.make_state_dict(0, head_scale=0.1))
    return net, synthetic.make_train_batch(B=2, crop_hw=(64, 128), raw_hw=(80, 160))


def _flat_grads(net):
    return torch.cat([p.grad.flatten() for _, p in sorted(net.named_parameters())])


def _real_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    net, batch = _real_net()
    from upflow_pytorch_amd import parallel
    from upflow_pytorch_amd.train import Trainer
    parallel.init_from_env(backend='gloo')
    tr = Trainer(net, lr=1e-4)
    assert tr.distributed and tr.world == world and type(tr.net).__name__ == 'DistributedDataParallel'
    stats = tr.step(tr.shard(batch))
    q.put((rank, stats, _flat_grads(tr.raw_net).numpy(), sum(int(p.grad is not None) for p in tr.raw_net.parameters())))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_real_upflow_net_trainer_under_ddp_two_ranks():
    """UPFlow_net (80 parameters, stacked training schedule, all four loss terms) + Trainer + DDP: both ranks end with the
    same all-reduced gradient, equal to the mean of the two shards' single-process gradients (the loss terms that are
    ratios of sums — occlusion-weighted distillation — are per-replica quantities, exactly as under the reference's
    nn.DataParallel, utils/tools.py:130-148 + `loss.mean()`, scripts/simple_train.py:23-54)."""
    torch.set_num_threads(4)
    net, batch = _real_net()
    from upflow_pytorch_amd.train import Trainer
    ref = []
    for r in range(2):
        net.zero_grad(set_to_none=True)
        tr = Trainer(net, lr=0.0, weight_decay=0.0, distributed=False)        # lr 0: the weights stay put between shards
        shard = {k: (v[r:r + 1] if torch.is_tensor(v) else v) for k, v in batch.items()}
        ref.append((tr.step(shard), _flat_grads(net).clone()))
    want = (ref[0][1] + ref[1][1]) / 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    g0, g1 = torch.from_numpy(res[0][2]), torch.from_numpy(res[1][2])
    assert res[0][3] == res[1][3] == 80, 'every parameter takes part in the bucket'
    assert torch.equal(g0, g1)
    scale = float(want.abs().max())
    assert float((g0 - want).abs().max()) <= 1e-5 * scale, float((g0 - want).abs().max()) / scale
    assert res[0][1] == res[1][1]
    assert abs(res[0][1]['loss'] - (ref[0][0]['loss'] + ref[1][0]['loss']) / 2) <= 1e-5 * abs(res[0][1]['loss'])


def _bucket_worker(rank, world, port, q, cap_mb):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      UPF_DDP_BUCKET_MB=str(cap_mb))
    torch.set_num_threads(2)
    net, batch = _real_net()
    from upflow_pytorch_amd import parallel
    from upflow_pytorch_amd.train import Trainer
    parallel.init_from_env(backend='gloo')
    tr = Trainer(net, lr=1e-4)
    first = parallel.ddp_bucket_bytes(tr.net)
    for _ in range(3):                                   # (DDP rebuilds its buckets in gradient-ready order after the first step)
        stats = tr.step(tr.shard(batch))
    params = torch.cat([p.detach().flatten() for _, p in sorted(tr.raw_net.named_parameters())])
    q.put((rank, stats, _flat_grads(tr.raw_net).numpy(), params.numpy(), first, parallel.ddp_bucket_bytes(tr.net)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_three_gradient_buckets_equal_the_single_bucket():
    """parallel.ddp_wrap's default (round 6): three ~5 MB gradient buckets in completion order instead of one 25 MB bucket — the
    all-reduced gradients and the parameters after three optimizer steps are BIT-identical to the single-bucket run (an element's
    sum over ranks does not depend on which message carried it), on both ranks; the buckets cover all 13,978,196 bytes."""
    ctx = mp.get_context('spawn')
    runs = {}
    for cap in (25, 5):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q, cap)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=400) for _ in procs], key=lambda t: t[0])
        for p in procs:
            p.join(30)
            assert p.exitcode == 0
        assert (res[0][2] == res[1][2]).all() and (res[0][3] == res[1][3]).all()
        runs[cap] = res[0]
    assert len(runs[25][5]) == 1 and sum(runs[25][5]) == 13978196
    assert len(runs[5][5]) == 3 and sum(runs[5][5]) == 13978196 and max(runs[5][5]) <= 7 * 2 ** 20, runs[5][5]      # (a bucket closes with the parameter that crosses the cap: 5.8 + 5.4 + 2.8 MB)
    assert (runs[25][2] == runs[5][2]).all() and (runs[25][3] == runs[5][3]).all()
    assert runs[25][1] == runs[5][1]


@pytest.mark.timeout(300)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it re-executes under torch.distributed.run (here: gloo,
    --mode launch-check, since this box has no GPU) and rank 0 prints ONE JSON line naming the rank count."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--mode', 'launch-check', '--backend', 'gloo'],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['ranks'] == 2 and d['n_gpus'] == 2 and d['max_over_ranks'] == 2.0 and d['backend'] == 'gloo'


@pytest.mark.timeout(600)
@pytest.mark.parametrize('form', ['self_launch', 'driver'])
def test_bench_launch_path_at_eight_ranks(form):
    """The launch path of the driver's 1/2/4/8 scaling run, at EIGHT ranks, on this GPU-less box: rendezvous on
    127.0.0.1 with a free port, W warm-up + K timed steps between barriers, MAX over ranks, the per-rank spread, ONE JSON line.
    `driver` = the command the driver issues (python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 ...); `self_launch` = `python bench.py --gpus 8` spawning its own ranks."""
    import json
    import socket
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    tail = [os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '5', '--warmup', '2', '--mode', 'launch-check', '--backend', 'gloo']
    if form == 'driver':
        s_ = socket.socket()
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
        s_.close()
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
               '--master-port', str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['ranks'] == 8 and d['n_gpus'] == 8 and d['max_over_ranks'] == 8.0 and d['backend'] == 'gloo' and d['steps'] == 5
    assert d['master'].startswith('127.0.0.1:')
    # ranks sleep 2 / 4 / 6 ms per step (rank % 3): the line reports the slowest rank's time and the spread
    sp = d['rank_ms_per_step']
    assert sp['max'] >= 5.5 and sp['min'] <= 4.5 and sp['slowest_rank'] % 3 == 2 and d['ms_per_step'] >= sp['max'] - 0.5


# ------------------------------------------------------------------------------------------------ batch mismatch under DDP
class _FakeGraph(object):
    """Stands for a captured hipGraph on a box without a GPU: the decision logic around it is what is tested."""
    replays = 0

    def replay(self):
        self.replays += 1


def _mismatch_worker(rank, world, port, q, mode):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import warnings
    from upflow_pytorch_amd import parallel
    from upflow_pytorch_amd.train import Trainer
    parallel.init_from_env(backend='gloo')
    tr = Trainer(TinyFlowNet(), lr=1e-3, weight_decay=0.0, batch_check=mode)
    shard = tr.shard(_global_batch())
    tr.step(shard)
    # pretend this step was captured (the GPU path does this after `graph_warmup` steps)
    tr.use_graph, tr._graph = True, _FakeGraph()
    tr._static = {k: v.clone() for k, v in shard.items()}
    tr._static_stats = torch.zeros(3)
    tr.step(shard)                                                        # every rank matches: every rank replays
    matched_replays = tr._graph.replays
    bad = {k: v[:1] for k, v in shard.items()} if rank == 1 else shard      # rank 1 gets a last partial batch
    outcome = 'ok'
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter('always')
        try:
            tr.step(bad)
        except ValueError as e:
            outcome = 'raised: ' + str(e)[:40]
    params = torch.cat([p.detach().flatten() for p in tr.raw_net.parameters()])
    q.put((rank, matched_replays, tr._graph.replays, outcome, len(wlist), params.numpy().tolist()))
    if mode == 'collective':
        torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_batch_mismatch_under_ddp_is_a_collective_decision():
    """VERDICT r3 weak 13 / ADVICE r3: one rank with a batch that differs from the captured one must not step eagerly (its own
    all-reduce sequence) while the others replay their graphs.  batch_check='collective': the ranks exchange one bit per step and
    ALL of them take the eager step (both warn, nobody replays, parameters stay identical across ranks)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mismatch_worker, args=(r, 2, port, q, 'collective')) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=90) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, matched, total, outcome, nwarn, _ in res:
        assert matched == 1 and total == 1, (rank, matched, total)         # the mismatching step replayed on NO rank
        assert outcome == 'ok' and nwarn >= 1
    assert res[0][5] == res[1][5]                                          # the eager DDP step kept the ranks in lockstep


def test_batch_mismatch_raise_mode_single_process_logic():
    """batch_check='raise': the mismatching rank raises (no per-step exchange).  Checked on the decision function alone, with a
    trainer that believes it is rank 1 of 2 (no process group needed: 'raise' never communicates)."""
    sys.path.insert(0, ROOT)
    from upflow_pytorch_amd.train import Trainer
    tr = Trainer(TinyFlowNet(), lr=1e-3, weight_decay=0.0, distributed=False, batch_check='raise')
    b = _global_batch()
    tr.use_graph, tr._graph, tr._static = True, _FakeGraph(), {k: v.clone() for k, v in b.items()}
    tr.distributed, tr.world, tr.rank = True, 2, 1
    assert tr._replay_agreed(b) is True
    with pytest.raises(ValueError, match='rank 1'):
        tr._replay_agreed({k: v[:1] for k, v in b.items()})
    with pytest.raises(ValueError):
        Trainer(TinyFlowNet(), distributed=False, batch_check='maybe')
    assert tr.capture_error is None
