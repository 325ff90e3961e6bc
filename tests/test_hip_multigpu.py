"""Row (e) on hardware: N > 1 on RCCL.  Every test here needs at least two GPUs in one process group and SKIPS itself on a
one-GPU box (the builder's boxes are; the driver's 8-GPU node runs them).  The same assertions run on CPU with two `gloo`
ranks in tests/test_distributed_cpu.py (operators through the oracle stub), so the logic is covered here and now; what
these add is RCCL, DDP's bucket on real devices, the per-device kernel attributes and the captured step with a multi-rank
all-reduce inside the hipGraph.   Replaces /root/reference/utils/tools.py:130-148 (single-process nn.DataParallel)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
         'norm_moments_across_images': False, 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}


def _need_two_gpus():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip('needs >= 2 GPUs (found %d)' % (torch.cuda.device_count() if torch.cuda.is_available() else 0))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(mode):
    sys.path.insert(0, ROOT)
    from upflow_pytorch_amd import synthetic
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    conf = UPFlow_net.config()
    d = dict(FLAGS)
    d.update(synthetic.TRAIN_FLAGS)
    d['train_conv_dtype'] = mode
    conf.update(d, verbose=False)
    net = conf()
    net.load_state_dict(synthetic.make_state_dict(0, head_scale=0.1))
    return net


def _flat_grads(net):
    return torch.cat([p.grad.detach().flatten().float() for _, p in sorted(net.named_parameters())])


def _worker(rank, world, port, mode, graph, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                          HSA_ENABLE_IPC_MODE_LEGACY='0')
        sys.path.insert(0, ROOT)
        from upflow_pytorch_amd import parallel, synthetic
        from upflow_pytorch_amd.train import Trainer
        r, w, local = parallel.init_from_env(backend='nccl')
        dev = torch.device('cuda', local)
        torch.cuda.set_device(dev)
        if world == 1:                            # (init_from_env is a no-op outside a multi-rank launch: a one-rank nccl group)
            torch.distributed.init_process_group(backend='nccl', rank=0, world_size=1)
        gbatch = {k: v.to(dev) for k, v in synthetic.make_train_batch(B=2 * world).items()}
        # the mean of the per-shard gradients, computed locally WITHOUT DDP (same weights, each rank's shard in turn)
        want = None
        for s in range(world):
            net = _build(mode).to(dev).train()
            idx = parallel.shard_indices(2 * world, s, world)
            shard = {k: (v[idx] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 2 * world else v) for k, v in gbatch.items()}
            out = net(dict(shard, if_loss=True))
            sum(out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')).backward()
            g = _flat_grads(net)
            want = g if want is None else want + g
        want = want / world
        tr = Trainer(_build(mode), lr=1e-4, device=dev, graph=graph)
        assert tr.distributed and type(tr.net).__name__ == 'DistributedDataParallel'
        stats = tr.step(tr.shard(gbatch))
        got = _flat_grads(tr.raw_net)
        rel = float((got - want).norm() / want.norm())
        cos = float(torch.dot(got, want) / (got.norm() * want.norm()))
        if graph:
            for _ in range(tr.graph_warmup + 2):
                stats = tr.step(tr.shard(gbatch))
        params = torch.cat([p.detach().flatten() for _, p in sorted(tr.raw_net.named_parameters())])
        chk = torch.stack([params.double().sum(), params.double().abs().sum()])
        both = [torch.zeros_like(chk) for _ in range(world)]
        torch.distributed.all_gather(both, chk)
        q.put((rank, rel, cos, stats, bool(graph) == (tr._graph is not None), tr.capture_fallback,
               all(torch.equal(b, both[0]) for b in both)))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception as e:                       # (surface the failure in the parent instead of a queue timeout)
        import traceback
        q.put((rank, 'error', traceback.format_exc() + str(e)))


@pytest.mark.parametrize('world,mode,graph', [(2, 'fp32', False), (2, 'bf16', False), (2, 'bf16', True), (1, 'bf16', True)])
def test_two_rank_rccl_trainer_step_real_net(world, mode, graph):
    """2 ranks, the real UPFlow_net (ctypes autograd Functions, shared-gradient gate nodes) under DDP on RCCL, eager and with
    the step captured as one hipGraph: the all-reduced gradient equals the mean of the per-shard gradients, both ranks hold
    identical parameters after the steps, and a requested graph is really a graph (no silent eager downgrade)."""
    if world > 1:                                 # (the one-rank variant runs the same worker on a one-GPU box)
        _need_two_gpus()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, graph, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for r in res:
        assert r[1] != 'error', r[2]
    for rank, rel, cos, stats, graph_ok, fallback, same_params in res:
        print('rank %d: DDP gradient vs mean of per-shard gradients: relative difference %.3g, cosine %.6f' % (rank, rel, cos))
        assert rel <= (2e-3 if mode == 'fp32' else 2e-2) and cos >= (0.99999 if mode == 'fp32' else 0.9995)
        assert all(np.isfinite(v) for v in stats.values())
        assert graph_ok and not fallback and same_params
    assert all(r[3] == res[0][3] for r in res)    # the logged loss terms are the all-reduced ones on every rank


def test_operator_on_every_device():
    """Per-device state of the library (the > 48 KB LDS opt-in is per kernel AND device): the same operators right after each
    other on cuda:0 and cuda:1 give bit-identical results."""
    _need_two_gpus()
    sys.path.insert(0, ROOT)
    from upflow_pytorch_amd import ops
    outs = []
    for d in (0, 1, 0):
        dev = torch.device('cuda', d)
        g = torch.Generator().manual_seed(5)
        f1 = torch.randn(2, 196, 12, 40, generator=g).bfloat16().to(dev)
        f2 = torch.randn(2, 196, 12, 40, generator=g).bfloat16().to(dev)
        x = torch.randn(2, 565, 24, 80, generator=g).bfloat16().to(dev)
        w = (torch.randn(128, 565, 3, 3, generator=g) * 0.02).bfloat16().to(dev)
        with torch.cuda.device(dev):
            c = ops.corr81_forward_raw(f1, f2, leaky_slope=0.1)
            y = torch.empty(2, 128, 24, 80, dtype=torch.bfloat16, device=dev)
            ops.conv3x3_forward_raw(x, ops.conv3x3_pack(w), torch.zeros(128, device=dev), y, 1, 0.1)
        outs.append((c.cpu(), y.cpu()))
    for c, y in outs[1:]:
        assert torch.equal(c, outs[0][0]) and torch.equal(y, outs[0][1])


@pytest.mark.parametrize('mode', ['infer', 'train'])
def test_bench_two_gpus(mode):
    """`python bench.py --gpus 2 [--mode train]` exits 0 with one JSON line: ranks 2, backend nccl, the step a hipGraph with no
    capture fallback, (train) the gradient all-reduce timed."""
    _need_two_gpus()
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--no-train-probe']
    if mode == 'train':
        cmd += ['--mode', 'train']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['ranks'] == 2 and 'nccl' in line['config']['backend']
    assert line['config']['hip_graph'] and not line['config']['capture_fallback'] and line['value'] > 0
    if mode == 'train':
        assert line['dtype'] == 'bf16' and line['config']['gradient_allreduce_ms'] > 0
