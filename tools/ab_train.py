#!/usr/bin/env python3
"""A/B of training-step variants on ONE box in ONE process: a config-3 trainer is captured per variant (python switches are
read while the step is captured, so each graph bakes its variant) and the graphs are replayed alternately.
    python tools/ab_train.py "" no_prepack=1"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench, _weights
from upflow_pytorch_amd import ops
from upflow_pytorch_amd.model.upflow import UPFlow_net
from upflow_pytorch_amd.train import Trainer, synthetic_train_batch

def apply(opts):
    ops.shared_conv_grads.no_prepack = bool(int(opts.get('no_prepack', 0)))

dev = torch.device('cuda', 0)
batch = synthetic_train_batch(4, seed=0, device=dev)
variants = sys.argv[1:] or ['', '']
trainers = []
for v in variants:
    opts = dict(kv.split('=') for kv in v.split(',') if kv)
    apply(opts)
    conf = UPFlow_net.config(); d = dict(bench.FLAGS); d.update(bench.TRAIN_FLAGS); d['train_conv_dtype'] = 'bf16'; conf.update(d, verbose=False)
    net = conf(); net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
    tr = Trainer(net, lr=1e-4, device=dev, distributed=False, graph=True)
    for _ in range(tr.graph_warmup + 2):
        tr.step(batch, sync_stats=False)
    assert tr._graph is not None, getattr(tr, 'capture_error', None)
    trainers.append(tr)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    for tr in trainers:
        tr.step(batch, sync_stats=False)
    torch.cuda.synchronize()
tot = [0.0] * len(trainers)
ROUNDS, N = 10, 10
for _ in range(ROUNDS):
    for i, tr in enumerate(trainers):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(N):
            tr.step(batch, sync_stats=False)
        torch.cuda.synchronize(); tot[i] += (time.perf_counter() - t) / N
for v, t in zip(variants, tot):
    print('%-30s %.3f ms / step' % (v or '(defaults)', t / ROUNDS * 1e3), flush=True)
