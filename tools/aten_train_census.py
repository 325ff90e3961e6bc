#!/usr/bin/env python3
"""Every ATen operator of one config-3 training step (eager) that touches GPU tensors: name, count, the package line that asked for
it ('autograd' = the engine's own gradient accumulation), largest operand.  What a captured step contains besides
libupflow_hip.so's kernels.    python tools/aten_train_census.py [min count]"""
import os
import sys
import traceback
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from upflow_pytorch_amd.train import synthetic_train_batch
import test_hip_train as T

VIEWS = ('view', 'slice', 'select', 'expand', 'permute', 'transpose', 't', 'detach', 'alias', 'as_strided', 'unsqueeze', 'squeeze', 'reshape',
         '_unsafe_view', 'unbind', 'split', 'split_with_sizes', 'narrow', 'empty', 'empty_like', 'empty_strided', 'new_empty', 'size', 'stride',
         'is_contiguous', 'sym_size', 'sym_stride', 'sym_numel', 'numel', 'dim', 'lift_fresh', 'result_type', 'is_pinned', 'new_empty_strided',
         'set_', 'storage_offset', 'sym_storage_offset', '_local_scalar_dense', 'item', 'record_stream', 'is_same_size')
log = defaultdict(lambda: [0, 0])


def tensors(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, (list, tuple)):
        for y in x:
            yield from tensors(y)


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split('.')[0]
        ts = [t for t in tensors(args) if t.is_cuda]
        if name not in VIEWS and ts:
            where = [f for f in traceback.extract_stack() if 'upflow_pytorch_amd' in f.filename]
            w = where[-1] if where else None
            key = (name, '%s:%d' % (os.path.basename(w.filename), w.lineno) if w else 'autograd')
            log[key][0] += 1
            log[key][1] = max(log[key][1], max(t.numel() for t in ts))
        return func(*args, **(kwargs or {}))


tr = T._config3_trainer('bf16', False)
batch = synthetic_train_batch(4, device='cuda')
tr.step(batch)
with Spy():
    tr.step(batch)
lim = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tot = 0
for (n, w), (k, ne) in sorted(log.items(), key=lambda kv: -kv[1][0]):
    tot += k
    if k >= lim:
        print('%-28s x%-4d largest operand %10d  %s' % (n, k, ne, w))
print('# total', tot)
