#!/usr/bin/env python3
"""Profile target for rocprofv3: the dominant kernel (corr81 forward, config-2 l4 shape) launched N times.
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python tools/prof_corr.py
  rocprofv3 --kernel-trace --pmc FETCH_SIZE   --output-format csv -d ... -- python tools/prof_corr.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE   --output-format csv -d ... -- python tools/prof_corr.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops

B, C, H, W = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (4, 32, 96, 320)))
dt = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[sys.argv[5] if len(sys.argv) > 5 else 'bf16']
var = sys.argv[6] if len(sys.argv) > 6 else 'plain'      # norm: the normalising variant (upf_corr81_norm_forward); norm_c8: the same with octet output (the kernel inside the step at the fine levels)
if var in ('norm_c8', 'norm_c8_mixed'):      # norm_c8_mixed (round 6): fp16 features -> octets of type `dt`: the bf16 step's launch since its pyramid stays fp16
    fwd = lambda: ops.corr81_norm_forward_c8(f1, f2, out8, leaky_slope=0.1)
elif var == 'norm':
    fwd = lambda: ops.corr81_norm_forward_raw(f1, f2, out=out, leaky_slope=0.1)
else:
    fwd = lambda: ops.corr81_forward_raw(f1, f2, out=out, leaky_slope=0.1)
g = torch.Generator().manual_seed(2004)
# (norm_c8 at a ragged width — KITTI's native 94x311 level — takes ROW-PITCHED features, like the step's pair buffers: ops.empty_nchw)
fdt = torch.float16 if var == 'norm_c8_mixed' else dt
pair = ops.empty_nchw((2, B, C, H, W), fdt, 'cuda', pitched=var.startswith('norm_c8')) if dt != torch.float32 else torch.empty(2, B, C, H, W, device='cuda')
pair[0].copy_(torch.randn(B, C, H, W, generator=g).cuda())
pair[1].copy_(torch.randn(B, C, H, W, generator=g).cuda())
f1, f2 = pair[0], pair[1]
out = torch.empty(B, 81, H, W, device='cuda', dtype=dt)
out8 = ops.c8_empty(B, 88, H, W, dt, 'cuda') if dt != torch.float32 else None
# evict the 256 MiB infinity cache between launches so FETCH_SIZE reflects HBM, not MALL hits
junk = torch.empty(320 * 1024 * 1024, dtype=torch.uint8, device='cuda')
for i in range(20):
    junk.add_(1)
    fwd()
torch.cuda.synchronize()
# back-to-back (inputs/outputs warm in the infinity cache, as inside the network right after their producer)
for i in range(20):
    fwd()
torch.cuda.synchronize()
print('done')
