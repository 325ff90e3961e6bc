#!/usr/bin/env python3
"""Config 3 at its real size (256x832 crops, batch 4): is the first step's gradient bit-reproducible between two fresh runs?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_hip_train as T
from upflow_pytorch_amd.train import synthetic_train_batch
for size, kw in (('256x832 B=4', {}), ('128x192 B=2', dict(crop_hw=(128, 192), raw_hw=(160, 256)))):
    batch = synthetic_train_batch(2 if kw else 4, device='cuda', **kw)
    gs, losses = [], []
    for r in range(2):
        tr = T._config3_trainer('bf16', False)
        tr.net.train()
        b = dict(batch); b['if_loss'] = True
        out = tr.net(b)
        loss, parts = tr.loss_manager.compute_loss(out)
        loss.backward()
        gs.append({n: p.grad.clone() for n, p in tr.raw_net.named_parameters()})
        losses.append(float(loss))
    bad = [n for n in gs[0] if not torch.equal(gs[0][n], gs[1][n])]
    print('%s: loss %r / %r; parameters whose gradient differs between two runs: %d of %d %s' % (size, losses[0], losses[1], len(bad), len(gs[0]), bad[:4]))
