#!/usr/bin/env python3
"""Per-kernel timing of the HIP operators at the pyramid-level shapes of BASELINE configs 2/4/5.
Algorithmic bytes per SURVEY.md §8(d).  Run under gpurun."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops


def timeit(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    dev = 'cuda'
    res = []
    levels = {2: (4, [(196, 6, 20), (128, 12, 40), (96, 24, 80), (64, 48, 160), (32, 96, 320)]),
              4: (8, [(196, 7, 16), (128, 14, 32), (96, 28, 64), (64, 56, 128), (32, 112, 256)]),
              5: (1, [(196, 15, 45), (128, 30, 90), (96, 60, 180), (64, 120, 360), (32, 240, 720)])}
    for cfg, (B, lv) in levels.items():
        for (C, H, W) in lv:
            for dt in (torch.bfloat16, torch.float32):
                s = 2 if dt == torch.bfloat16 else 4
                f1 = torch.randn(B, C, H, W, device=dev).to(dt); f2 = torch.randn(B, C, H, W, device=dev).to(dt)
                flow = torch.randn(B, 2, H, W, device=dev) * 2
                out = torch.empty(B, 81, H, W, device=dev, dtype=dt)
                t = timeit(lambda: ops.corr81_forward_raw(f1, f2, out=out, leaky_slope=0.1))
                byt = s * B * H * W * (2 * C + 81)
                res.append(dict(op='corr81_fwd', cfg=cfg, B=B, C=C, H=H, W=W, dtype=str(dt), us=t, GBs=byt / t / 1e3))
                t = timeit(lambda: ops.warp(f2, flow, 'literal'))
                byt = B * H * W * (2 * s * C + 8)
                res.append(dict(op='warp_fwd', cfg=cfg, B=B, C=C, H=H, W=W, dtype=str(dt), us=t, GBs=byt / t / 1e3))
                t = timeit(lambda: ops.normalize(f1))
                byt = B * H * W * C * 2 * s
                res.append(dict(op='normalize', cfg=cfg, B=B, C=C, H=H, W=W, dtype=str(dt), us=t, GBs=byt / t / 1e3))
    for r in res:
        print('%-11s cfg%d B%d C%3d %4dx%-4d %-15s %8.1f us %8.1f GB/s' % (r['op'], r['cfg'], r['B'], r['C'], r['H'], r['W'], r['dtype'], r['us'], r['GBs']))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(res, open('gpurun_out/kbench.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
