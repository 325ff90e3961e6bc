#!/usr/bin/env python3
"""Which backward node is the first whose inputs / outputs differ between two identical first training steps?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_hip_train as T
import _weights
from upflow_pytorch_amd import ops

batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
REC = []
def wrap(cls):
    orig = cls.backward
    def bw(ctx, *g):
        out = orig(ctx, *g)
        outs = out if isinstance(out, tuple) else (out,)
        REC[-1].append((cls.__name__, [x.detach().clone() if torch.is_tensor(x) else None for x in g],
                        [x.detach().clone() if torch.is_tensor(x) else None for x in outs]))
        return out
    cls.backward = staticmethod(bw)
for name in dir(ops):
    c = getattr(ops, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c)
from upflow_pytorch_amd.utils import loss as L
wrap(L._GreyFunction)
for r in range(2):
    REC.append([])
    tr = T._config3_trainer('bf16', False)
    tr.net.train()
    b = dict(batch); b['if_loss'] = True
    out = tr.net(b)
    loss, parts = tr.loss_manager.compute_loss(out)
    loss.backward()
    torch.cuda.synchronize()
a, b = REC
print('backward nodes recorded:', len(a), len(b))
shown = 0
for i, (x, y) in enumerate(zip(a, b)):
    def same(p, q):
        return all((u is None and v is None) or (u is not None and v is not None and u.shape == v.shape and torch.equal(u, v)) for u, v in zip(p, q))
    si, so = same(x[1], y[1]), same(x[2], y[2])
    if 88 <= i <= 97:
        print('   (node %d %s in %s out %s: in shapes %s, out shapes %s)' % (i, x[0], si, so, [tuple(t.shape) for t in x[1] if t is not None], [tuple(t.shape) for t in x[2] if t is not None][:6]))
    if not (si and so):
        shapes = [tuple(t.shape) for t in x[1] if t is not None]
        print('node %4d %-28s inputs %s outputs %s   grad-in shapes %s' % (i, x[0], 'same' if si else 'DIFFER', 'same' if so else 'DIFFER', shapes))
        shown += 1
        if shown > 12:
            break
