#!/usr/bin/env python3
"""The 'upup' pyramid-distillation term of ONE direction at config 3's sizes: fused op (upf_msd_upup_*) vs the composition it replaces,
forward + backward, eager wall time per call over 50 calls after warm-up (launch-bound numbers: compare the two, not with kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from upflow_pytorch_amd import ops
B, H, W = 4, 256, 832
g = torch.Generator().manual_seed(0)
y = (torch.randn(B, 2, H, W, generator=g) * 3).cuda()
occ = (torch.rand(B, 1, H, W, generator=g) > 0.3).float().cuda()
xs = [(torch.randn(B, 2, h, w, generator=g)).cuda().requires_grad_(True) for h, w in [(4, 13), (8, 26), (16, 52), (32, 104), (64, 208)]]
def fused():
    return torch.autograd.grad(ops.msd_upup_loss(xs, y, occ, 1.0), xs)
def comp():
    t = 0
    for x in xs:
        s, so = ops.robust_loss_sums(ops.flow_upsample(x, H, W, True), y, occ, q=0.4, eps=0.01)
        t = t + s / (so + 1e-6)
    return torch.autograd.grad(t, xs)
def gtime(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10): fn()
    gr.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / 200 * 1e6
print('fused %.1f us | composition %.1f us   (forward + backward of one direction, graph replay)' % (gtime(fused), gtime(comp)))
