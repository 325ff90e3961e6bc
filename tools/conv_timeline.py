#!/usr/bin/env python3
"""Inside one workgroup of the interleaved-staging convolution: per-wave cycle stamps of the first chunks ("ablate" bit 16).
  python tools/conv_timeline.py Cin Cout H W dilation [extra ablate bits]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops, _lib
a = [int(v) for v in sys.argv[1:]]
Cin, Cout, H, W, d = a[:5]
extra = a[5] if len(a) > 5 else 0
B = 8
x = torch.randn(B, Cin, H, W, device='cuda').bfloat16()
w = (torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.02).bfloat16()
b = torch.randn(Cout, device='cuda')
y = torch.empty(B, Cout, H, W, device='cuda', dtype=torch.bfloat16)
packed = ops.conv3x3_pack(w)
dbg = torch.zeros(2 * 4 * 24 * 4, dtype=torch.int64, device='cuda')
_lib.lib().upf_conv_set_debug_buffer(_lib.ptr(dbg))
for _ in range(3):
    ops.conv3x3_forward_raw(x, packed, b, y, d, 0.1)
ops.conv_set_option('ablate', 16 | extra)
ops.conv3x3_forward_raw(x, packed, b, y, d, 0.1)
torch.cuda.synchronize()
ops.conv_set_option('ablate', 0)
t = dbg.cpu().view(2, 4, 24, 4)
for wg in range(2):
    base = int(t[wg][t[wg] > 0].min())
    print('ablate %d workgroup %d: per chunk  start / matrix phase done / barrier passed  (cycles)' % (extra, 0 if wg == 0 else 100))
    for wave in range(4):
        print('  wave %d: ' % wave + ' | '.join('%6d %6d %6d' % tuple(int(v) - base for v in t[wg, wave, c, :3]) for c in range(4, 10)))
