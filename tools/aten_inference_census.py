#!/usr/bin/env python3
"""ATen operators inside one inference forward (what a captured hipGraph of it contains besides libupflow_hip.so's kernels):
reductions (multi-block ones zero their semaphores with a memset node) and same-dtype contiguous copies (memcpy nodes)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from collections import Counter
from torch.utils._python_dispatch import TorchDispatchMode
import bench

RED = ('sum', 'mean', 'amax', 'amin', 'max', 'min', 'norm', 'var', 'std', 'prod', 'any', 'all', 'var_mean', 'std_mean')


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split('.')[0]
        t = args[0] if args and torch.is_tensor(args[0]) else None
        if t is not None and t.is_cuda:
            if name in RED:
                self.c[('REDUCTION ' + name, t.numel())] += 1
            elif name in ('copy_', 'clone', 'contiguous', '_to_copy', 'cat', 'zeros', 'zero_', 'fill_', 'zeros_like', 'full'):
                self.c[(name, None)] += 1
        return func(*args, **(kwargs or {}))


for dt in (torch.bfloat16, torch.float32):
    net = bench.build_net(dt, torch.device('cuda', 0), True, 'hip_x3')
    im1 = torch.rand(4, 3, 384, 1280, device='cuda') - 0.45
    im2 = torch.rand(4, 3, 384, 1280, device='cuda') - 0.45
    with torch.no_grad():
        net({'im1': im1, 'im2': im2, 'if_loss': False})
        with Spy() as spy:
            net({'im1': im1, 'im2': im2, 'if_loss': False})
    print(dt, dict(spy.c))
