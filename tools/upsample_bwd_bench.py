#!/usr/bin/env python3
"""upf_flow_upsample_backward at the resizes of a config-3 training step (graph replay of 20 launches), default build or UPF_HIP_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from kbench import graph_time
from upflow_pytorch_amd import _lib
L = _lib.lib()
out = []
for (B, C, h, w, H, W) in [(4, 2, 4, 13, 256, 832), (4, 2, 8, 26, 256, 832), (4, 2, 16, 52, 256, 832), (4, 2, 32, 104, 256, 832), (4, 2, 64, 208, 256, 832), (8, 3, 64, 208, 256, 832), (8, 2, 32, 104, 64, 208)]:
    gy = torch.randn(B, C, H, W, device='cuda'); gx = torch.empty(B, C, h, w, device='cuda')
    fn = lambda: _lib.call('upf_flow_upsample_backward', _lib.ptr(gy), _lib.ptr(gx), B, C, h, w, H, W, 0, _lib.stream_ptr(gy.device))
    out.append('%dx%d->%dx%d (B%d C%d): %.1f us' % (h, w, H, W, B, C, graph_time(fn, iters=5)))
print(os.environ.get('UPF_HIP_LIB', 'default').split('/')[-1], ' | '.join(out))
