#!/usr/bin/env python3
"""Where a config-3 training step spends its time: forward (no losses) / losses forward / backward + optimizer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import bench, _weights
from upflow_pytorch_amd.model.upflow import UPFlow_net
from upflow_pytorch_amd.train import Trainer, synthetic_train_batch
conf = UPFlow_net.config(); d = dict(bench.FLAGS); d.update(bench.TRAIN_FLAGS); conf.update(d, verbose=False)
net = conf(); net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
dev = torch.device('cuda')
tr = Trainer(net.float(), device=dev)
batch = synthetic_train_batch(4, seed=0, device=dev)
def t(fn, n=6):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
net.train()
def fwd_noloss():
    b = dict(batch); b['if_loss'] = False
    return net(b)
def fwd_loss():
    b = dict(batch); b['if_loss'] = True
    return net(b)
a = t(fwd_noloss); b = t(fwd_loss); c = t(lambda: tr.step(batch))
print('forward without losses %.1f ms | + losses forward %.1f ms | backward + Adam %.1f ms | step %.1f ms' % (a, b - a, c - b, c))
