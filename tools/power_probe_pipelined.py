#!/usr/bin/env python3
"""What bounds the PIPELINED inference step?  rocm-smi socket power + sclk sampled while the config-2 step
(384x1280, bf16) runs as  B x streams  =  4x1, 4x2, 4x3, 4x4, 4x5, 4x6, 8x1, 8x2, 16x1  (frame pairs per step x steps in flight),
~4 s each: pairs/s, ms per step, mean / max power, mean sclk.   python tools/power_probe_pipelined.py [BxS ...] > profiles/r05_power_probe_pipelined.txt   (GPU_MAX_HW_QUEUES=8 to lift the HIP runtime's 4 hardware queues)"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench  # noqa: E402
from power_probe import smi  # noqa: E402
from upflow_pytorch_amd import synthetic  # noqa: E402
from upflow_pytorch_amd.runtime import PipelinedInference  # noqa: E402


def num(v):
    try:
        return float(str(v).strip('()').lower().replace('mhz', '').replace('w', ''))
    except Exception:
        return float('nan')


def run(net, B, streams, dev, seconds=4.0):
    pipe = PipelinedInference(net, B, 384, 1280, streams=streams, device=dev)
    for s in range(streams):
        a, b = synthetic.make_images(2 + 100 * s, B, 384, 1280)
        pipe.load(s, a.to(dev), b.to(dev))
    pipe.synchronize()
    for _ in range(10 * streams):
        pipe.replay()
    pipe.synchronize()
    stop, samples = [False], []

    def sampler():
        while not stop[0]:
            samples.append(smi()[:2])
            time.sleep(0.15)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8 * streams):
            pipe.replay()
        pipe.synchronize()
        n += 8 * streams
    dt = time.perf_counter() - t0
    stop[0] = True
    th.join()
    samples = samples[3:]
    pw = [num(p) for p, _ in samples if p is not None]
    ck = [num(c) for _, c in samples if c is not None]
    del pipe
    torch.cuda.empty_cache()
    return B * n / dt, dt / n * 1e3, (sum(pw) / len(pw) if pw else float('nan')), (max(pw) if pw else float('nan')), (sum(ck) / len(ck) if ck else float('nan'))


def main():
    dev = torch.device('cuda', 0)
    net = bench.build_net(torch.bfloat16, dev)
    p0, s0, _ = smi()
    print('# %s; idle: %s W, sclk %s' % (torch.cuda.get_device_name(0), p0, s0))
    print('%-8s %-8s %12s %12s %10s %10s %10s' % ('B/step', 'streams', 'pairs/s', 'ms/step', 'W mean', 'W max', 'sclk MHz'))
    cases = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(4, 1), (4, 2), (4, 3), (4, 4), (4, 5), (4, 6), (8, 1), (8, 2), (16, 1), (4, 4)]
    print('# GPU_MAX_HW_QUEUES=%s' % os.environ.get('GPU_MAX_HW_QUEUES', '(default: 4)'))
    for B, st in cases:
        r = run(net, B, st, dev)
        print('%-8d %-8d %12.1f %12.3f %10.0f %10.0f %10.0f' % ((B, st) + r), flush=True)


if __name__ == '__main__':
    main()
