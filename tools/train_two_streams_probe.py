#!/usr/bin/env python3
"""Would a training step gain from running as two half-batch chains on two streams?  Pessimistic probe: TWO independent
trainers (own nets) at batch B/2, each captured into its own hipGraph, replayed side by side on two streams, against ONE trainer
at batch B (the weight gradients run twice at half the K here; a real split would contract both halves in one launch).
    python tools/train_two_streams_probe.py [B=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench, _weights
from upflow_pytorch_amd.model.upflow import UPFlow_net
from upflow_pytorch_amd.train import Trainer, synthetic_train_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda', 0)


def make(b, seed):
    conf = UPFlow_net.config(); d = dict(bench.FLAGS); d.update(bench.TRAIN_FLAGS); d['train_conv_dtype'] = 'bf16'; conf.update(d, verbose=False)
    net = conf(); net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
    tr = Trainer(net, lr=1e-4, device=dev, distributed=False, graph=True)
    batch = synthetic_train_batch(b, seed=seed, device=dev)
    for _ in range(tr.graph_warmup + 2):
        tr.step(batch, sync_stats=False)
    assert tr._graph is not None, getattr(tr, 'capture_error', None)
    return tr, batch


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


full, fb = make(B, 0)
h1, b1 = make(B // 2, 1)
h2, b2 = make(B // 2, 2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1): h1._graph.replay()
    with torch.cuda.stream(s2): h2._graph.replay()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)


print('one trainer, batch %d          : %.3f ms / step' % (B, timeit(lambda: full._graph.replay())))
print('one trainer, batch %d          : %.3f ms / step' % (B // 2, timeit(lambda: h1._graph.replay())))
print('two trainers, batch %d each, 2 streams: %.3f ms per pair of steps (= one batch-%d step)' % (B // 2, timeit(both), B))
