#!/usr/bin/env python3
"""The dense stacks of the fine levels in the channel-octet layout with and without the MERGED NARROW TAIL (pwc_modules._PackedTailC8,
round 6): us per whole-stack forward (hipGraph replay), at [8,.,96,320] and [8,.,48,160] (config 2) and the native-KITTI fine level.
    python tools/tail_bench.py  -> gpurun_out/tail_bench.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops
from upflow_pytorch_amd.model.pwc_modules import FlowEstimatorDense_v2
from tools.conv_layers import graph_time


def main():
    dt = torch.bfloat16
    lines = []
    for kind, args in (('sgu', (64, (32, 32, 32, 16, 8), 3)), ('est', (120, (128, 128, 96, 64, 32), 2))):
        stack = FlowEstimatorDense_v2(args[0], f_channels=args[1], out_channel=args[2]).cuda().to(dt).eval()
        for (B, H, W) in ((8, 96, 320), (8, 48, 160), (8, 94, 311), (2, 94, 311)):
            nconv, n_in = sum(stack._f) // 8, stack._ch_in // 8
            buf8 = torch.randn(B, nconv + n_in, H, W, 8, device='cuda').to(dt)
            out = torch.empty(B, args[2], H, W, device='cuda', dtype=dt)
            t = {}
            for merge in (False, True):
                stack._no_merge_tail = not merge
                with torch.no_grad():
                    t[merge] = graph_time(lambda: stack.forward_in_buffer_c8(buf8, out=out))
            lines.append('%s stack [%d,.,%d,%d]: layer per pass %7.1f us, merged tail %7.1f us (%+.1f us)' % (kind, B, H, W, t[False], t[True], t[True] - t[False]))
            print(lines[-1], flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    open('gpurun_out/tail_bench.txt', 'w').write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
