#!/usr/bin/env python3
"""Every weight-gradient contraction of one config-3 bf16 training step, timed alone (graph replay of 20 launches: kernel +
reduction), with its flop rate: records the (levels, Cin, Cout, k, dilation) of each ops.conv_wgrad_multi / conv_wgrad_s2d call
of an eager step, then replays each on random data.    python tools/wgrad_layers.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import bench, _weights
from kbench import graph_time
from upflow_pytorch_amd import ops
from upflow_pytorch_amd.model.upflow import UPFlow_net
from upflow_pytorch_amd.train import Trainer, synthetic_train_batch

dev = torch.device('cuda')
conf = UPFlow_net.config(); d = dict(bench.FLAGS); d.update(bench.TRAIN_FLAGS); d['train_conv_dtype'] = 'bf16'; conf.update(d, verbose=False)
net = conf(); net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
tr = Trainer(net, lr=1e-4, device=dev, distributed=False, graph=False)
batch = synthetic_train_batch(4, seed=0, device=dev)
tr.step(batch)
calls = []
orig_multi, orig_s2d = ops.conv_wgrad_multi, ops.conv_wgrad_s2d
def rec_multi(uses, Cin, Cout, k, dilation, **kw):
    calls.append(('multi', [tuple(x.shape) + (x.stride(0), g.stride(0)) for x, g in uses], Cin, Cout, k, dilation))
    return orig_multi(uses, Cin, Cout, k, dilation, **kw)
def rec_s2d(xs, g, Cin, Cout):
    calls.append(('s2d', [tuple(xs.shape) + (xs.stride(0), g.stride(0))], Cin, Cout, 3, 1))
    return orig_s2d(xs, g, Cin, Cout)
ops.conv_wgrad_multi, ops.conv_wgrad_s2d = rec_multi, rec_s2d
tr.step(batch)
ops.conv_wgrad_multi, ops.conv_wgrad_s2d = orig_multi, orig_s2d
torch.cuda.synchronize()
print('%d weight-gradient contractions per step' % len(calls))
print('%-5s %5s %5s %2s %3s  %-44s %9s %9s %9s' % ('kind', 'Cin', 'Cout', 'k', 'd', 'levels (B x H x W)', 'GFLOP', 'us', 'TFLOP/s'))
tot_us = tot_f = 0.0
for kind, lv, Cin, Cout, k, dil in calls:
    uses = []
    for (B, C, H, W, xbs, gbs) in lv:
        x = torch.randn(B, C, H, W, device=dev).bfloat16()
        g = (torch.randn(B, Cout, H, W, device=dev) * 0.1).bfloat16()
        uses.append((x, g))
    if kind == 'multi':
        fn = lambda: ops.conv_wgrad_multi(uses, Cin, Cout, k, dil)
        flop = sum(2.0 * k * k * Cin * Cout * B * H * W for (B, C, H, W, _, _) in lv)
    else:
        fn = lambda: ops.conv_wgrad_s2d(uses[0][0], uses[0][1], Cin, Cout)
        flop = sum(2.0 * 9 * Cin * Cout * B * H * W for (B, C, H, W, _, _) in lv)       # (the stride-2 layer's own flop: H, W are the OUTPUT size)
    t = graph_time(fn, iters=5)
    tot_us += t; tot_f += flop
    print('%-5s %5d %5d %2d %3d  %-44s %9.2f %9.2f %9.1f' % (kind, Cin, Cout, k, dil, ' '.join('%dx%dx%d' % (B, H, W) for (B, C, H, W, _, _) in lv), flop / 1e9, t, flop / t / 1e6), flush=True)
print('total %.1f GFLOP in %.1f us = %.1f TFLOP/s' % (tot_f / 1e9, tot_us, tot_f / tot_us / 1e6))
