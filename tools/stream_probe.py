#!/usr/bin/env python3
"""Throughput of the config-2 step (a) as one graph at batch 4 / 8 / 16, (b) as TWO batch-4 graphs replayed on two streams
(the coarse pyramid levels of one step under the fine levels of the other).   python tools/stream_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from upflow_pytorch_amd import synthetic  # noqa: E402
from upflow_pytorch_amd.runtime import GraphedInference  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    net = bench.build_net(torch.bfloat16, dev)
    for B in (4, 8, 16):
        im1, im2 = synthetic.make_images(2, B, 384, 1280)
        r = GraphedInference(net, B, 384, 1280, device=dev)
        r.load(im1.to(dev), im2.to(dev))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:
            r.replay()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            r.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 40 * 1e3
        print('one graph, batch %2d: %.3f ms/step = %.0f pairs/s' % (B, ms, B * 1e3 / ms), flush=True)
        if B != 4:
            del r
            torch.cuda.empty_cache()
        else:
            r4 = r
    # two batch-4 graphs on two streams
    net2 = bench.build_net(torch.bfloat16, dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    im1, im2 = synthetic.make_images(3, 4, 384, 1280)
    with torch.cuda.stream(s2):
        rb = GraphedInference(net2, 4, 384, 1280, device=dev)
        rb.load(im1.to(dev), im2.to(dev))
    torch.cuda.synchronize()
    for offset in (False, True):
        for _ in range(10):
            with torch.cuda.stream(s1):
                r4.replay()
            with torch.cuda.stream(s2):
                rb.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        if offset:                                     # start stream 2 half a step late
            with torch.cuda.stream(s2):
                torch.cuda._sleep(int(1.5e-3 * 2.4e9))
        for _ in range(n):
            with torch.cuda.stream(s1):
                r4.replay()
            with torch.cuda.stream(s2):
                rb.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (2 * n) * 1e3
        print('two batch-4 graphs on two streams%s: %.3f ms per step (2 steps in flight) = %.0f pairs/s' % (' (offset half a step)' if offset else '', ms, 4e3 / ms), flush=True)


if __name__ == '__main__':
    main()
