#!/usr/bin/env python3
"""Is the inference step POWER (energy) bound?  Samples rocm-smi (socket power, sclk) while the GPU runs, for ~5 s each:
the captured config-2 step back to back; the widest layer (565->128 at 96x320) back to back; a coarse-level layer (6x20) back to
back; the full-resolution stem layer (3->16, bandwidth bound); idle.   python tools/power_probe.py > profiles/r04_power_probe.txt"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from upflow_pytorch_amd import ops, synthetic  # noqa: E402
from upflow_pytorch_amd.runtime import GraphedInference  # noqa: E402


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        c = d[sorted(d)[0]]
        pw = [float(v) for k, v in c.items() if 'ower' in k and 'W' in k and v not in ('N/A', None)]
        sclk = [v for k, v in c.items() if k.startswith('sclk')]
        return (pw[0] if pw else None), (sclk[0] if sclk else None), c
    except Exception as e:
        return None, None, {'error': str(e)}


def sample_while(fn, seconds=5.0):
    stop = [False]
    samples = []

    def sampler():
        while not stop[0]:
            samples.append(smi()[:2])
            time.sleep(0.2)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.perf_counter() - t0
    stop[0] = True
    th.join()
    return samples[2:], dt / max(n, 1)


def main():
    dev = torch.device('cuda', 0)
    p0, s0, raw = smi()
    print('# rocm-smi keys:', sorted(raw)[:40])
    print('idle: power %s W, sclk %s' % (p0, s0))
    net = bench.build_net(torch.bfloat16, dev)
    im1, im2 = synthetic.make_images(2, 4, 384, 1280)
    r = GraphedInference(net, 4, 384, 1280, device=dev)
    r.load(im1.to(dev), im2.to(dev))
    g = torch.Generator().manual_seed(1)

    def conv_fn(N, Cin, Cout, h, w):
        x = torch.randn(N, Cin, h, w, generator=g).to(dev).bfloat16()
        wt = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(dev).bfloat16()
        b = torch.zeros(Cout, device=dev)
        y = torch.empty(N, Cout, h, w, device=dev, dtype=torch.bfloat16)
        pk = ops.conv3x3_pack(wt)
        return lambda: ops.conv3x3_forward_raw(x, pk, b, y, 1, 0.1)
    jobs = [('full step (graph replay)', r.replay), ('565->128 @ [8,565,96,320]', conv_fn(8, 565, 128, 96, 320)),
            ('565->128 @ [8,565,6,20] (coarsest level)', conv_fn(8, 565, 128, 6, 20)), ('3->16 @ [8,3,384,1280] (stem)', conv_fn(8, 3, 16, 384, 1280))]
    for name, fn in jobs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        samples, per = sample_while(fn)
        pw = [s[0] for s in samples if s[0] is not None]
        print('%-44s %8.1f us/call  power avg %6.1f W (min %.0f max %.0f, %d samples)  sclk %s' %
              (name, per * 1e6, sum(pw) / max(len(pw), 1), min(pw or [0]), max(pw or [0]), len(pw), sorted(set(str(s[1]) for s in samples))[-3:]))
    cap = subprocess.run(['rocm-smi', '--showmaxpower'], capture_output=True, text=True).stdout
    print(cap.strip().splitlines()[-6:])


if __name__ == '__main__':
    main()
