#!/usr/bin/env python3
"""The memory-bound narrow convolutions of config 2 (feature pyramid top, SGU output convs, SGU mask estimator)
under different launch options: python tools/conv_narrow_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops


def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = 8
LAYERS = [  # Cin, Cout, H, W, stride
    (3, 16, 384, 1280, 1), (3, 16, 384, 1280, 2), (16, 16, 384, 1280, 2), (16, 16, 192, 640, 1), (16, 32, 192, 640, 1), (16, 32, 192, 640, 2),
    (32, 32, 192, 640, 2), (32, 32, 96, 320, 1), (64, 32, 96, 320, 1), (128, 32, 96, 320, 1), (176, 8, 96, 320, 1), (184, 3, 96, 320, 1), (531, 32, 96, 320, 1)]
for opts in ({}, {'rpw4_min': 1 << 30}):
    prev = {k: ops.conv_set_option(k, v) for k, v in opts.items()}
    print('options', opts)
    for Cin, Cout, H, W, s in LAYERS:
        x = torch.randn(B, Cin, H, W, device='cuda').bfloat16()
        w = (torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.05).bfloat16()
        b = torch.randn(Cout, device='cuda')
        ho, wo = ops.conv3x3_out_hw(H, W, s)
        y = torch.empty(B, Cout, ho, wo, device='cuda', dtype=torch.bfloat16)
        pk = ops.conv3x3_pack(w)
        t = timeit(lambda: ops.conv3x3_forward_raw(x, pk, b, y, 1, 0.1, s))
        mb = (x.numel() + y.numel()) * 2 / 1e6
        print('  %3d->%-3d %4dx%-4d s%d : %7.1f us  %7.1f MB algorithmic -> %5.2f TB/s' % (Cin, Cout, H, W, s, t, mb, mb / t / 1e6 * 1e6 / 1e6 * 1e0 if False else mb / t))
    for k, v in prev.items():
        ops.conv_set_option(k, v)
