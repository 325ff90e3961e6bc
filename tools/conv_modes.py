#!/usr/bin/env python3
"""Experiment: how much of the end-to-end step is MIOpen solver choice?  (eager, config 2)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import bench, _weights

def run(tag, benchmark, channels_last, dtype=torch.bfloat16, steps=8):
    torch.backends.cudnn.benchmark = benchmark
    dev = torch.device('cuda')
    net = bench.build_net(dtype, dev)
    im1, im2 = _weights.make_images(2, 4, 384, 1280)
    im1, im2 = im1.to(dev), im2.to(dev)
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
        im1 = im1.to(memory_format=torch.channels_last); im2 = im2.to(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            net({'im1': im1, 'im2': im2, 'if_loss': False})
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            net({'im1': im1, 'im2': im2, 'if_loss': False})
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print('%-40s %8.2f ms/step  %7.1f pairs/s' % (tag, dt * 1e3, 4 / dt), flush=True)

mode = sys.argv[1] if len(sys.argv) > 1 else 'default'
cfg = {'default': (False, False), 'benchmark': (True, False), 'channels_last': (False, True), 'benchmark_cl': (True, True),
       'fp32': (False, False), 'fp32_benchmark': (True, False)}[mode]
run(mode + ' ' + os.environ.get('MIOPEN_FIND_MODE', ''), cfg[0], cfg[1], dtype=torch.float32 if mode.startswith('fp32') else torch.bfloat16)
