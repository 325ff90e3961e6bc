#!/usr/bin/env python3
"""The narrow (Cout <= 32) channel-octet layers under the conv kernel's ablation switches (upf_conv_set_option("ablate")):
1 no matrix phase, 8 no weight loads, 2 no x loads (C8: no LDS-DMA), 4 no LDS staging writes (NCHW only).  us per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import conv_layers as CL
from upflow_pytorch_amd import ops
B, H, W = 8, 96, 320
for (name, C8c, Cout, ycs) in [('est.conv5/pure', 536, 32, True), ('est.conv_last/pure', 568, 2, False), ('sgu.conv3', 128, 32, True), ('sgu.conv_last', 184, 3, False), ('est.conv2/pure', 248, 128, True)]:
    res = []
    for abl in (0, 1, 8, 9, 2, 3, 11):
        prev = ops.conv_set_option('ablate', abl)
        t = CL.bench_layer_c8(B, C8c, 0, Cout, 1, H, W, ycs)[0]
        ops.conv_set_option('ablate', prev)
        res.append('abl%-2d %6.1f' % (abl, t))
    byt = 2.0 * B * H * W * (C8c + (Cout if ycs else Cout))
    print('%-20s %3d->%3d  %s   (%.0f MB)' % (name, C8c, Cout, '  '.join(res), byt / 1e6), flush=True)
