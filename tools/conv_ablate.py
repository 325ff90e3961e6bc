#!/usr/bin/env python3
"""Where does a convolution launch spend its time?  Selected config-2 layers under the kernel's ablation switches
(upf_conv_set_option "ablate": 1 no matrix phase, 2 no x loads, 4 no LDS staging writes, 8 no weight loads).
Results are wrong by construction; only the durations mean something."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops
from conv_layers import bench_layer

LAYERS = [('est.conv5', 531, 32, 3, 1, 96, 320), ('est.conv_last', 563, 2, 3, 1, 96, 320), ('sgu.conv1', 64, 32, 3, 1, 96, 320),
          ('sgu.conv3', 128, 32, 3, 1, 96, 320), ('sgu.conv5', 176, 8, 3, 1, 96, 320), ('est.conv3', 371, 96, 3, 1, 96, 320),
          ('ctx.conv0', 565, 128, 3, 1, 96, 320), ('est.conv4', 467, 64, 3, 1, 96, 320), ('ctx.conv3', 128, 96, 3, 8, 96, 320),
          ('est.conv5@l3', 531, 32, 3, 1, 48, 160)]
CODES = [0, 1, 2, 8, 3, 6, 10, 14, 15]
print('%-14s' % 'layer' + ''.join('%9s' % ('abl%d' % c) for c in CODES))
for (name, Cin, Cout, k, d, H, W) in LAYERS:
    row = []
    for c in CODES:
        prev = ops.conv_set_option('ablate', c)
        try:
            row.append(bench_layer(8, Cin, Cout, k, d, 1, H, W)[0])
        finally:
            ops.conv_set_option('ablate', prev)
    print('%-14s' % name + ''.join('%9.1f' % t for t in row), flush=True)
