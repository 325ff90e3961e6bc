#!/usr/bin/env python3
"""Ablation of wgrad_pc_kernel on a few config-3 layers (UPF_WGRAD_ABLATE in the environment: 1 = no matrix phase,
2 = no global loads, 4 = no LDS staging writes): kernel + reduction per launch.   UPF_WGRAD_ABLATE=1 python tools/wgrad_ablate.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from kbench import graph_time
from upflow_pytorch_amd import ops
dev = 'cuda'
LV5 = [(8, 64, 208), (8, 32, 104), (8, 16, 52), (8, 8, 26), (8, 4, 13)]
cases = [(563, 2, 1, LV5), (531, 32, 1, LV5), (243, 128, 1, LV5), (565, 128, 1, LV5), (128, 128, 2, LV5), (16, 16, 1, [(8, 128, 416)]), (160, 16, 1, [(8, 64, 208)] * 2 + LV5[1:4])]
out = []
for Cin, Cout, d, lv in cases:
    uses = [(torch.randn(B, Cin, H, W, device=dev).bfloat16(), (torch.randn(B, Cout, H, W, device=dev) * 0.1).bfloat16()) for B, H, W in lv]
    t = graph_time(lambda: ops.conv_wgrad_multi(uses, Cin, Cout, 3, d), iters=5)
    out.append('%d->%d d%d: %.1f' % (Cin, Cout, d, t))
print('ablate=%s  ' % os.environ.get('UPF_WGRAD_ABLATE', '0') + ' | '.join(out))
