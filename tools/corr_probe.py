#!/usr/bin/env python3
"""corr81 forward at the config-2 / config-5 1/4-resolution shapes: average kernel time and roofline fraction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops
for (B, C, H, W) in [(4, 32, 96, 320), (8, 32, 96, 320), (1, 32, 240, 720), (8, 64, 48, 160)]:
    f1 = torch.randn(B, C, H, W, device='cuda').bfloat16(); f2 = torch.randn(B, C, H, W, device='cuda').bfloat16()
    out = torch.empty(B, 81, H, W, device='cuda', dtype=torch.bfloat16)
    ops.corr81_forward_timed(f1, f2, out, 0.1, nrep=20)
    avg, mn = ops.corr81_forward_timed(f1, f2, out, 0.1, nrep=200)
    mb = 2 * B * H * W * (2 * C + 81) / 1e6
    print('[%d,%d,%d,%d] avg %.2f us min %.2f us  %.1f MB -> %.1f%% of 8 TB/s' % (B, C, H, W, avg, mn, mb, 100 * mb / avg / 8e6 * 1e0 * 1e0 if False else 100 * (mb * 1e6) / (avg * 1e-6) / 8e12))

# backward (config 3 shapes, fp32 as the trainer runs it, and bf16)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for dt in (torch.float32, torch.bfloat16):
    for (B, C, H, W) in [(4, 32, 64, 208), (4, 64, 32, 104), (4, 32, 96, 320)]:
        f1 = torch.randn(B, C, H, W, device='cuda').to(dt); f2 = torch.randn(B, C, H, W, device='cuda').to(dt)
        go = torch.randn(B, 81, H, W, device='cuda').to(dt)
        t = timeit(lambda: ops.corr81_backward_raw(f1, f2, go))
        mb = f1.element_size() * B * H * W * (4 * C + 81) / 1e6
        print('bwd %s [%d,%d,%d,%d] %.1f us  %.1f MB -> %.1f%% of 8 TB/s' % (str(dt)[6:], B, C, H, W, t, mb, 100 * mb * 1e6 / (t * 1e-6) / 8e12))
