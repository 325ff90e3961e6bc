#!/usr/bin/env python3
"""Per-kernel timing of the HIP operators at the pyramid-level shapes of BASELINE configs 2/4/5.

Each operator is captured NREP times into one HIP graph and the graph replay is timed with events on
the capture stream: that removes the ~13 us python/ctypes launch overhead, leaving kernel time plus
the ~1.5 us dependent-kernel boundary.  Algorithmic bytes per SURVEY.md §8(d).  Run under gpurun.
"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops

NREP = 20


def graph_time(fn, iters=10):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(NREP):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(iters):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / NREP * 1e3)
    return best  # us per launch


def train_kernels(res):
    """The training-only kernels at the two fine levels of config 3 (B = 8: both directions stacked), bf16."""
    dev, dt = 'cuda', torch.bfloat16
    B = 8
    lv = [(64, 208), (32, 104)]
    for (Cin, Cout, d) in [(567, 128, 1), (531, 32, 1), (563, 2, 1), (243, 128, 1), (128, 128, 4), (96, 64, 16)]:
        uses = [(torch.randn(B, Cin, H, W, device=dev).to(dt), (torch.randn(B, Cout, H, W, device=dev) * 0.1).to(dt)) for H, W in lv]
        t = graph_time(lambda: ops.conv_wgrad_multi(uses, Cin, Cout, 3, d), iters=5)
        flop = sum(2.0 * 9 * Cin * Cout * B * H * W for H, W in lv)
        res.append(dict(op='wgrad+reduce', cfg=3, B=B, shape='%d->%d d%d 64x208+32x104' % (Cin, Cout, d), dtype='bfloat16', us=t, TFs=flop / t / 1e6))
    for (C, H, W) in [(128, 64, 208), (32, 64, 208), (128, 32, 104)]:
        a = torch.randn(B, C, H, W, device=dev).to(dt); b = torch.randn(B, C, H, W, device=dev).to(dt); y = torch.randn(B, C, H, W, device=dev).to(dt)
        dst = torch.empty_like(a)
        t = graph_time(lambda: ops.act_grad(a, y, 0.1, add=b, dst=dst, want_bias=True))
        res.append(dict(op='act_grad', cfg=3, B=B, C=C, H=H, W=W, dtype='bfloat16', us=t, GBs=4 * 2 * B * C * H * W / t / 1e3))
    for (C, H, W) in [(16, 256, 832), (32, 128, 416), (64, 64, 208)]:
        a = torch.randn(B, C, H, W, device=dev).to(dt)
        t = graph_time(lambda: ops.space_to_depth2(a))
        res.append(dict(op='space_to_depth', cfg=3, B=B, C=C, H=H, W=W, dtype='bfloat16', us=t, GBs=2 * 2 * B * C * H * W / t / 1e3))
    g1 = torch.rand(B // 2, 1, 256, 832, device=dev); g2 = (g1 + 0.05 * torch.randn_like(g1)).requires_grad_(True)
    gout = torch.randn(B // 2, 1, 256, 832, device=dev)

    def census():
        d = ops.census_distance(g1, g2)
        torch.autograd.grad(d, g2, gout)
    t = graph_time(census, iters=5)
    res.append(dict(op='census fwd+bwd', cfg=3, B=B // 2, C=1, H=256, W=832, dtype='float32', us=t, GBs=(B // 2) * 256 * 832 * 4 * 6 / t / 1e3))


def main(only=None):
    dev = 'cuda'
    res = []
    levels = {2: (4, [(196, 6, 20), (128, 12, 40), (96, 24, 80), (64, 48, 160), (32, 96, 320)]),
              3: (4, [(196, 4, 13), (128, 8, 26), (96, 16, 52), (64, 32, 104), (32, 64, 208)]),
              4: (8, [(196, 7, 16), (128, 14, 32), (96, 28, 64), (64, 56, 128), (32, 112, 256)]),
              5: (1, [(196, 15, 45), (128, 30, 90), (96, 60, 180), (64, 120, 360), (32, 240, 720)])}
    for cfg, (B, lv) in levels.items():
        if only and cfg not in only:
            continue
        for (C, H, W) in lv:
            for dt in (torch.bfloat16, torch.float32):
                s = 2 if dt == torch.bfloat16 else 4
                f1 = torch.randn(B, C, H, W, device=dev).to(dt); f2 = torch.randn(B, C, H, W, device=dev).to(dt)
                flow = torch.randn(B, 2, H, W, device=dev) * 2
                out = torch.empty(B, 81, H, W, device=dev, dtype=dt)
                go = torch.randn(B, 81, H, W, device=dev).to(dt)
                y = torch.empty_like(f2)

                def rec(op, t, byt):
                    res.append(dict(op=op, cfg=cfg, B=B, C=C, H=H, W=W, dtype=str(dt).replace('torch.', ''), us=t, GBs=byt / t / 1e3))
                rec('corr81_fwd', graph_time(lambda: ops.corr81_forward_raw(f1, f2, out=out, leaky_slope=0.1)), s * B * H * W * (2 * C + 81))
                rec('warp_fwd', graph_time(lambda: ops.WarpFunction.apply(f2, flow, 1, 0)), B * H * W * (2 * s * C + 8))
                rec('normalize', graph_time(lambda: ops.normalize(f1)), B * H * W * C * 2 * s)
                if cfg == 3:
                    rec('corr81_bwd', graph_time(lambda: ops.corr81_backward_raw(f1, f2, go)), s * B * H * W * (4 * C + 81))
            xo = torch.randn(B, 3, H, W, device=dev)
            fl = torch.randn(B, 2, H, W, device=dev)
            t = graph_time(lambda: ops.sgu_blend(fl, xo, None, want_inter=False))
            res.append(dict(op='sgu_blend', cfg=cfg, B=B, C=3, H=H, W=W, dtype='float32', us=t, GBs=B * H * W * (8 + 12 + 8) / t / 1e3))
        # final-level blend and full-res ops
        C, H, W = lv[-1]
        Hf, Wf = 4 * H, 4 * W
        xo = torch.randn(B, 3, H, W, device=dev)
        olf = torch.randn(B, 2, Hf, Wf, device=dev)
        t = graph_time(lambda: ops.sgu_blend(None, xo, olf, want_inter=False))
        res.append(dict(op='sgu_blend_final', cfg=cfg, B=B, C=3, H=Hf, W=Wf, dtype='float32', us=t, GBs=(B * Hf * Wf * 16 + B * H * W * 12) / t / 1e3))
        fl = torch.randn(B, 2, H, W, device=dev)
        t = graph_time(lambda: ops.flow_upsample(fl, Hf, Wf, True))
        res.append(dict(op='flow_upsample', cfg=cfg, B=B, C=2, H=Hf, W=Wf, dtype='float32', us=t, GBs=(B * Hf * Wf * 8 + B * H * W * 8) / t / 1e3))
        t = graph_time(lambda: ops.occ_check(olf, olf))
        res.append(dict(op='occ_check', cfg=cfg, B=B, C=2, H=Hf, W=Wf, dtype='float32', us=t, GBs=(B * Hf * Wf * 24) / t / 1e3))
    if not only or 3 in only:
        train_kernels(res)
    for r in res:
        if 'TFs' in r:
            print('%-15s cfg%d B%d %-28s %-9s %8.2f us %8.1f TFLOP/s %5.1f%% of 2.5 PF' % (r['op'], r['cfg'], r['B'], r['shape'], r['dtype'], r['us'], r['TFs'], r['TFs'] / 25.0))
            continue
        print('%-15s cfg%d B%d C%3d %4dx%-4d %-9s %8.2f us %8.1f GB/s  %5.1f%% of 8TB/s' % (r['op'], r['cfg'], r['B'], r['C'], r['H'], r['W'], r['dtype'], r['us'], r['GBs'], r['GBs'] / 80.0))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(res, open('gpurun_out/kbench.json', 'w'), indent=1)


if __name__ == '__main__':
    main([int(a) for a in sys.argv[1:]] or None)
