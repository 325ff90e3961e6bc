#!/usr/bin/env python3
"""Which python lines launch the ATen (non-upf) kernels of one config-3 bf16 training step.  A TorchDispatchMode logs every
aten op that does device work (views / allocations excluded) with the innermost frames inside this repository and the shapes
of its tensor arguments; the backward pass runs on the calling thread (set_multithreading_enabled(False)) so that the mode
and the python stack see it too.  Output: a table (phase, op, count, where, shapes).    python tools/train_glue.py"""
import os, sys, re, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench, _weights
from upflow_pytorch_amd.model.upflow import UPFlow_net
from upflow_pytorch_amd.train import Trainer, synthetic_train_batch

SKIP = re.compile(r'aten\.(view|_unsafe_view|reshape|slice|select|expand|permute|transpose|as_strided|detach|alias|unsqueeze|squeeze|t|'
                  r'empty|empty_like|empty_strided|new_empty|new_empty_strided|split|split_with_sizes|unbind|narrow|chunk|is_same_size|'
                  r'_local_scalar_dense|sym_size|sym_stride|stride|size|lift_fresh|unfold|diagonal|_reshape_alias|result_type|is_nonzero|'
                  r'new_zeros_placeholder|record_stream|set_)\.')
LOG = collections.defaultdict(int)
PHASE = ['forward']


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not SKIP.match(name + '.'):
            ts = [a for a in args if torch.is_tensor(a)]
            if any(t.is_cuda for t in ts) or (torch.is_tensor(out) and out.is_cuda):
                fr = [f for f in traceback.extract_stack() if f.filename.startswith(ROOT) and 'tools/train_glue' not in f.filename]
                where = ' <- '.join('%s:%d(%s)' % (os.path.relpath(f.filename, ROOT).replace('upflow_pytorch_amd/', ''), f.lineno, f.name)
                                    for f in reversed(fr[-3:]))
                shapes = ','.join('x'.join(map(str, t.shape)) + ('h' if t.dtype == torch.bfloat16 else '') for t in ts[:3])
                LOG[(PHASE[0], name.replace('aten.', ''), where, shapes)] += 1
        return out


conf = UPFlow_net.config(); d = dict(bench.FLAGS); d.update(bench.TRAIN_FLAGS); d['train_conv_dtype'] = 'bf16'; conf.update(d, verbose=False)
net = conf(); net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
dev = torch.device('cuda')
tr = Trainer(net, lr=1e-4, device=dev, distributed=False, graph=False)
batch = synthetic_train_batch(4, seed=0, device=dev)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
tr.optimizer.zero_grad(set_to_none=True)
with Mode():
    b = dict(batch); b['if_loss'] = True
    out = tr.net(b)
    PHASE[0] = 'loss'
    loss, parts = tr.loss_manager.compute_loss(out)
    PHASE[0] = 'backward'
    loss.backward()
    PHASE[0] = 'optimizer'
    tr.optimizer.step()
torch.cuda.synchronize()
byphase = collections.defaultdict(int)
for (ph, op, where, shapes), c in LOG.items():
    byphase[ph] += c
print('device-work aten ops in one step:', dict(byphase))
agg = collections.defaultdict(lambda: [0, set()])
for (ph, op, where, shapes), c in LOG.items():
    agg[(ph, op, where)][0] += c
    agg[(ph, op, where)][1].add(shapes)
for (ph, op, where), (c, shp) in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1][0])):
    if ph == 'optimizer':
        continue
    print('%-8s %4d  %-22s %s   [%s]' % (ph, c, op, where, ' | '.join(sorted(shp)[:3])))
