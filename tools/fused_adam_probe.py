#!/usr/bin/env python3
"""fp32 training step, config 3 full size: parameter update of torch.optim.Adam(fused=True) vs the foreach form, per parameter."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from upflow_pytorch_amd.train import synthetic_train_batch, Trainer
import test_hip_train as T
import _weights
from upflow_pytorch_amd.model.upflow import UPFlow_net


def make(mode, fused):
    conf = UPFlow_net.config()
    d = dict(T.FLAGS); d.update(_weights.TRAIN_FLAGS); d['train_conv_dtype'] = mode
    conf.update(d, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
    return Trainer(net, lr=1e-4, device=torch.device('cuda', 0), distributed=False, graph=False, fused_adam=fused)


batch = synthetic_train_batch(4, device='cuda')
for mode in sys.argv[1:] or ['fp32']:
    res = {}
    for fused in (False, True):
        tr = make(mode, fused)
        p0 = {n: p.detach().clone() for n, p in tr.raw_net.named_parameters()}
        s0 = tr.step(batch)
        g = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in tr.raw_net.named_parameters()}
        p1 = {n: p.detach().clone() for n, p in tr.raw_net.named_parameters()}
        s1 = tr.step(batch)
        res[fused] = (p0, p1, g, s0, s1)
        print(mode, 'fused' if fused else 'foreach', 'loss step0 %.4f step1 %.4f' % (s0['loss'], s1['loss']))
        del tr
        torch.cuda.empty_cache()
    worst = []
    for n in res[False][0]:
        d0 = (res[False][1][n] - res[False][0][n]); d1 = (res[True][1][n] - res[True][0][n])
        worst.append((float((d0 - d1).abs().max()), float(d0.abs().max()), float(d1.abs().max()), n))
    worst.sort(reverse=True)
    for w in worst[:8]:
        print('   max |update diff| %.3e   foreach max|update| %.3e   fused max|update| %.3e   %s' % w)
