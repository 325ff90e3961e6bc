#!/usr/bin/env python3
"""Profile target: one convolution shape launched N times (for rocprofv3 --kernel-trace / --pmc passes).
  [UPF_DTYPE=bf16|fp16|fp32] python tools/prof_conv.py Cin Cout H W dilation [B=8] [N=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops
a = [int(v) for v in sys.argv[1:]]
Cin, Cout, H, W, d = a[:5]
B = a[5] if len(a) > 5 else 8
N = a[6] if len(a) > 6 else 20
DT = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[os.environ.get('UPF_DTYPE', 'bf16')]   # fp32: the split-precision kernel
x = torch.randn(B, Cin, H, W, device='cuda').to(DT)
if os.environ.get('UPF_ZERO_X'):
    x.zero_()
w = (torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.02).to(DT)
b = torch.randn(Cout, device='cuda')
y = torch.empty(B, Cout, H, W, device='cuda', dtype=DT)
packed = ops.conv3x3_pack(w)
for _ in range(N):
    ops.conv3x3_forward_raw(x, packed, b, y, d, 0.1)
torch.cuda.synchronize()
