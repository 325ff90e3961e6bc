#!/usr/bin/env python3
"""Sizing only: what a config-3 training step would gain if the LeakyReLU-mask / residual-add / bias-partial pass (upf_act_grad) cost
nothing, i.e. the upper bound of fusing it into the data-gradient kernels' epilogues.  Three captured steps: the product; the
passes inside the dense stacks removed (their outputs stay whatever the data gradient wrote — wrong numbers, same shapes);
every pass removed.   python tools/act_grad_cost_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench, _weights
from upflow_pytorch_amd import ops
from upflow_pytorch_amd.model.upflow import UPFlow_net
from upflow_pytorch_amd.train import Trainer, synthetic_train_batch
dev = torch.device('cuda', 0)
batch = synthetic_train_batch(4, seed=0, device=dev)
real = ops.act_grad
mode = {'skip': 'none'}
count = {'stack': 0, 'other': 0}


def probe(src, y=None, slope=0.0, add=None, dst=None, want_bias=False):
    in_stack = dst is not None and dst is not False
    count['stack' if in_stack else 'other'] += 1
    if mode['skip'] == 'all' or (mode['skip'] == 'stack' and in_stack):
        part = torch.zeros((src.shape[1], 32), dtype=torch.float32, device=src.device) if want_bias else None
        if dst is None:
            dst = src
        return (dst if dst is not False else None), part
    return real(src, y, slope, add, dst, want_bias)


ops.act_grad = probe
trainers = []
for m in ('none', 'stack', 'all'):
    mode['skip'] = m
    conf = UPFlow_net.config(); d = dict(bench.FLAGS); d.update(bench.TRAIN_FLAGS); d['train_conv_dtype'] = 'bf16'; conf.update(d, verbose=False)
    net = conf(); net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
    tr = Trainer(net, lr=1e-4, device=dev, distributed=False, graph=True)
    count['stack'] = count['other'] = 0
    for _ in range(tr.graph_warmup + 2):
        tr.step(batch, sync_stats=False)
    assert tr._graph is not None
    trainers.append((m, tr))
    print('# mode %-6s act_grad calls per step: %d inside dense stacks, %d elsewhere' % (m, count['stack'] // (tr.graph_warmup + 2), count['other'] // (tr.graph_warmup + 2)))
tot = [0.0] * len(trainers)
for _ in range(8):
    for i, (_, tr) in enumerate(trainers):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): tr._graph.replay()
        torch.cuda.synchronize(); tot[i] += (time.perf_counter() - t) / 10
for (m, _), t in zip(trainers, tot):
    print('passes removed: %-6s %.3f ms / step' % (m, t / 8 * 1e3))
