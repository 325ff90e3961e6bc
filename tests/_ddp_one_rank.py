"""Body of tests/test_hip_train.py::test_trainer_under_rccl_ddp_real_net, run as a script in its own process (the RCCL process group
and its background threads stay out of the pytest process).  Prints DDP-ONE-RANK-OK and exits 0 on success."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _weights                      # noqa: E402
from test_hip_train import build     # noqa: E402


def main():
    import os
    import socket
    import torch.distributed as dist
    from upflow_pytorch_amd import parallel
    from upflow_pytorch_amd.train import Trainer
    assert not dist.is_initialized()
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
    plain = Trainer(build(), lr=1e-4, distributed=False)
    want = plain.step(batch)
    want_g = {n: p.grad.detach().clone() for n, p in plain.raw_net.named_parameters()}
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    saved = {k: os.environ.get(k) for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    try:
        dist.init_process_group(backend='nccl', rank=0, world_size=1)
        tr = Trainer(build(), lr=1e-4, device=torch.device('cuda', 0))
        assert tr.distributed and type(tr.net).__name__ == 'DistributedDataParallel' and dist.get_backend() == 'nccl'
        got = tr.step(tr.shard(batch))
        # DDP + hipGraph: 11 eager warm-up steps, then the captured step (bucket all-reduce included) replays
        trg = Trainer(build(), lr=1e-4, device=torch.device('cuda', 0), graph=True)
        for _ in range(trg.graph_warmup + 2):
            sg = trg.step(batch)
        assert trg._graph is not None and all(np.isfinite(v) for v in sg.values())
        assert parallel.max_over_ranks(1.25, torch.device('cuda', 0)) == 1.25
        for k in want:
            assert abs(got[k] - want[k]) <= 1e-4 * max(1.0, abs(want[k])), (k, got[k], want[k])
        gnorm = float(torch.cat([g.flatten() for g in want_g.values()]).norm())
        worst = 0.0
        for n, p in tr.raw_net.named_parameters():
            assert p.grad is not None, n
            worst = max(worst, float((p.grad - want_g[n]).norm()) / max(float(want_g[n].norm()), 1e-3 * gnorm))
        print('DDP (1 rank, RCCL) vs plain gradient: worst relative difference %.3g' % worst)
        assert worst <= 2e-3          # (MIOpen's backward kernels use atomics: not bit-reproducible run to run)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


if __name__ == '__main__':
    main()
    print('DDP-ONE-RANK-OK')
