#!/usr/bin/env python3
"""What time constant does the clock / power governor have?  A captured graph of [wide layer (565->128 at 96x320: ~1350 W alone,
clock held at ~1.93 GHz), idle gap (a one-thread spin kernel)] x 20 is replayed for gaps of 0 ... 2 ms; the layer's own duration
(HIP events inside the graph are not portable, so: (replay time - gap-only replay time) / 20) shows whether idle time between
launches buys clock for the next one — i.e. whether the step is ENERGY bound (long window) or each kernel is bound by its own
instantaneous power (short window).   python tools/governor_probe.py > profiles/r04_governor_probe.txt"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upflow_pytorch_amd import ops  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 565, 96, 320, generator=g).to(dev).bfloat16()
    w = (torch.randn(128, 565, 3, 3, generator=g) * (2.0 / (9 * 565)) ** 0.5).to(dev).bfloat16()
    b = torch.zeros(128, device=dev)
    y = torch.empty(8, 128, 96, 320, device=dev, dtype=torch.bfloat16)
    pk = ops.conv3x3_pack(w)
    conv = lambda: ops.conv3x3_forward_raw(x, pk, b, y, 1, 0.1)
    # calibrate the spin kernel: cycles per microsecond
    torch.cuda._sleep(1000000)
    torch.cuda.synchronize()
    t = time.perf_counter()
    torch.cuda._sleep(20000000)
    torch.cuda.synchronize()
    cyc_per_us = 20000000 / ((time.perf_counter() - t) * 1e6)
    print('# spin kernel: %.1f cycles per us' % cyc_per_us)

    def graph_of(n_conv, gap_us):
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            conv()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            for _ in range(20):
                for _ in range(n_conv):
                    conv()
                if gap_us > 0:
                    torch.cuda._sleep(int(gap_us * cyc_per_us))
        return gr

    def timed(gr, seconds=3.0):
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            gr.replay()
            n += 1
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n / 20 * 1e6           # us per [convs + gap] unit

    print('%-14s %14s %14s %16s' % ('gap us', 'unit us', 'gap-only us', 'conv us (diff)'))
    for gap in (0, 50, 135, 270, 540, 1080, 2160):
        unit = timed(graph_of(1, gap))
        gap_only = timed(graph_of(0, gap)) if gap > 0 else 0.0
        print('%-14d %14.1f %14.1f %16.1f' % (gap, unit, gap_only, unit - gap_only), flush=True)
    # and the other way round: does a LONG busy burst get slower as it goes?  4 convs back to back per unit, gap 1080
    for n, gap in ((4, 0), (4, 1080), (8, 2160)):
        unit = timed(graph_of(n, gap))
        gap_only = timed(graph_of(0, gap)) if gap > 0 else 0.0
        print('%d convs + gap %-5d %12.1f %14.1f %16.1f per conv' % (n, gap, unit, gap_only, (unit - gap_only) / n), flush=True)


if __name__ == '__main__':
    main()
