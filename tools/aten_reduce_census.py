#!/usr/bin/env python3
"""Which ATen reductions does one config-3 training step launch, on how many elements, from which line?  (Multi-block ATen
reductions zero their semaphores with a memset node when captured — the node class seen mis-ordered in replayed hipGraphs here.)"""
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from upflow_pytorch_amd.train import synthetic_train_batch
import test_hip_train as T

RED = ('sum', 'mean', 'amax', 'amin', 'max', 'min', 'norm', 'var', 'std', 'prod', 'any', 'all')
log = []


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split('.')[0]
        if name in RED and args and torch.is_tensor(args[0]) and args[0].is_cuda:
            where = [f for f in traceback.extract_stack() if 'upflow_pytorch_amd' in f.filename]
            w = where[-1] if where else None
            log.append((name, args[0].numel(), tuple(args[0].shape), '%s:%d' % (os.path.basename(w.filename), w.lineno) if w else '?'))
        return func(*args, **(kwargs or {}))


tr = T._config3_trainer('bf16', False)
batch = synthetic_train_batch(4, device='cuda')
tr.step(batch)
with Spy():
    tr.step(batch)
from collections import Counter
c = Counter((n, ne, sh, w) for (n, ne, sh, w) in log)
for (n, ne, sh, w), k in sorted(c.items(), key=lambda kv: -kv[0][1]):
    print('%-5s x%-3d numel %9d  %-28s %s' % (n, k, ne, sh, w))
