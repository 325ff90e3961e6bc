#!/usr/bin/env python3
"""Matrix-core conv3x3 vs MIOpen (torch conv2d + LeakyReLU) at the estimator/context shapes of config 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from upflow_pytorch_amd import ops


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = 8   # both directions of batch 4
    shapes = [(115, 128, 1), (243, 128, 1), (371, 96, 1), (467, 64, 1), (531, 32, 1), (563, 2, 1),
              (565, 128, 1), (128, 128, 2), (128, 128, 4), (128, 96, 8), (64, 32, 1), (32, 2, 1), (64, 32, 1), (184, 3, 1)]
    miopen = '--miopen' in sys.argv
    for (H, W) in [(96, 320), (48, 160), (24, 80), (12, 40), (6, 20)]:
        tot_m, tot_h = 0.0, 0.0
        for Cin, Cout, d in shapes:
            x = torch.randn(B, Cin, H, W, device='cuda').bfloat16()
            w = (torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.02).bfloat16()
            b = torch.randn(Cout, device='cuda')
            y = torch.empty(B, Cout, H, W, device='cuda', dtype=torch.bfloat16)
            packed = ops.conv3x3_pack(w)
            bb = b.bfloat16()
            t_m = timeit(lambda: F.leaky_relu(F.conv2d(x, w, bb, padding=d, dilation=d), 0.1)) if miopen else float('nan')
            t_h = timeit(lambda: ops.conv3x3_forward_raw(x, packed, b, y, d, 0.1))
            fl = 2.0 * B * H * W * Cin * Cout * 9
            tot_m += t_m; tot_h += t_h
            print('%3dx%-3d Cin %3d Cout %3d d%d : MIOpen %8.1f us (%6.1f TF/s) | mfma conv %8.1f us (%6.1f TF/s)  x%.2f' %
                  (H, W, Cin, Cout, d, t_m, fl / t_m / 1e6, t_h, fl / t_h / 1e6, t_m / t_h), flush=True)
        print('sum: MIOpen %.0f us, mfma %.0f us' % (tot_m, tot_h))


if __name__ == '__main__':
    main()
