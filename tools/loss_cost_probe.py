#!/usr/bin/env python3
"""What the loss terms of a config-3 training step cost: the captured step with all terms, without the pyramid-distillation term,
without census, without both (sizing only: the product trains with all of them).   python tools/loss_cost_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench, _weights
from upflow_pytorch_amd.model.upflow import UPFlow_net
from upflow_pytorch_amd.train import Trainer, synthetic_train_batch
dev = torch.device('cuda', 0)
batch = synthetic_train_batch(4, seed=0, device=dev)
variants = [('all terms', {}), ('no distillation', {'multi_scale_distillation_weight': 0}), ('no census', {'photo_loss_census_weight': 0}),
            ('photometric + smoothness only', {'multi_scale_distillation_weight': 0, 'photo_loss_census_weight': 0})]
trainers = []
for name, over in variants:
    conf = UPFlow_net.config(); d = dict(bench.FLAGS); d.update(bench.TRAIN_FLAGS); d['train_conv_dtype'] = 'bf16'; d.update(over); conf.update(d, verbose=False)
    net = conf(); net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
    tr = Trainer(net, lr=1e-4, device=dev, distributed=False, graph=True)
    for _ in range(tr.graph_warmup + 2):
        tr.step(batch, sync_stats=False)
    assert tr._graph is not None
    trainers.append(tr)
tot = [0.0] * len(trainers)
for _ in range(8):
    for i, tr in enumerate(trainers):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): tr._graph.replay()
        torch.cuda.synchronize(); tot[i] += (time.perf_counter() - t) / 10
for (name, _), t in zip(variants, tot):
    print('%-32s %.3f ms / step' % (name, t / 8 * 1e3))
