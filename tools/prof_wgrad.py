#!/usr/bin/env python3
"""Profile target: the weight gradient of one layer over the two fine pyramid levels of a config-3 training step,
launched N times (for rocprofv3 --kernel-trace / --pmc passes).
  python tools/prof_wgrad.py Cin Cout dilation [N=6] [B=8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops
a = [int(v) for v in sys.argv[1:]]
Cin, Cout, d = a[:3]
N = a[3] if len(a) > 3 else 6
B = a[4] if len(a) > 4 else 8
uses = []
for H, W in ((64, 208), (32, 104)):
    uses.append((torch.randn(B, Cin, H, W, device='cuda').bfloat16(), (torch.randn(B, Cout, H, W, device='cuda') * 0.1).bfloat16()))
for _ in range(N):
    gw = ops.conv_wgrad_multi(uses, Cin, Cout, 3, d)
torch.cuda.synchronize()
print(float(gw.abs().mean()))
