#!/usr/bin/env python3
"""Profile target for the sampling operators (rocprofv3 --kernel-trace / --pmc): warp_fwd [4,32,96,320] bf16, final-level
sgu_blend and occ_check at 384x1280, each launched N times.   python tools/prof_sampling.py [N=12]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = 'cuda'
f2 = torch.randn(4, 32, 96, 320, device=dev).bfloat16(); flow = torch.randn(4, 2, 96, 320, device=dev) * 2
xo = torch.randn(4, 3, 96, 320, device=dev); olf = torch.randn(4, 2, 384, 1280, device=dev)
for _ in range(N):
    ops.WarpFunction.apply(f2, flow, 1, 0)
    ops.sgu_blend(None, xo, olf, want_inter=False)
    ops.occ_check(olf, olf)
torch.cuda.synchronize()
