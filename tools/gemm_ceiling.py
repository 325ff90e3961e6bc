#!/usr/bin/env python3
"""Calibration of "power bound" (VERDICT r5 weak 7 / next 5): what does the VENDOR's bf16 GEMM sustain on this box at the
M x N x K of the widest convolutions, on the same random data, beside this library's conv kernel?

    python tools/gemm_ceiling.py            -> gpurun_out/gemm_ceiling.txt (copied to profiles/r06_gemm_ceiling.txt)

GEMM: torch.matmul in bf16 (hipBLASLt / rocBLAS underneath) at M = 8*96*320 = 245,760 pixels, N = 128 output channels,
K = 9*565 = 5085 (est/ctx.conv0 at the 1/4 level of config 2) and K = 9*128 = 1152 (ctx.conv1/2); both operand orders
(pixels x K times K x N, and N x K times K x pixels) and a square 8192^3 for the part's own number.  The conv kernel:
565->128 and 128->128 (dilation 1 and 2) at [8,.,96,320] through ops.conv3x3_forward_raw.  Every timing is a hipGraph
replay of NREP launches, alternated GEMM / conv / GEMM / conv so that thermal state is shared; socket power and sclk are
sampled with rocm-smi while each one runs back to back for ~2 s.
"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops

NREP = 10


def graphed(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(NREP):
            fn()
    g.replay()
    torch.cuda.synchronize()
    return g


def time_graph(g, iters=8):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / NREP * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def smi_sample(stop, out):
    import json
    while not stop.is_set():
        try:
            d = json.loads(subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=10).stdout)
            c = d[sorted(d)[0]]
            pw = [float(v) for k, v in c.items() if 'ower' in k and 'W' in k and v not in ('N/A', None)]
            sclk = [v for k, v in c.items() if k.startswith('sclk')]
            out.append((pw[0] if pw else None, sclk[0] if sclk else None))
        except Exception:
            pass
        time.sleep(0.2)


def parse_smi(samples):
    import re
    pw = [p for p, _ in samples if p is not None]
    ck = []
    for _, s in samples:
        m = re.search(r'(\d+)', str(s)) if s is not None else None
        if m:
            ck.append(float(m.group(1)))
    avg = lambda a: sum(a) / len(a) if a else float('nan')
    return avg(pw), avg(ck)


def sustained(g, seconds=2.0):
    stop, samples = threading.Event(), []
    th = threading.Thread(target=smi_sample, args=(stop, samples))
    th.start()
    t0 = time.time()
    n = 0
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(10):
            g.replay()
        n += 10
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) / (n * NREP) * 1e3
    w, mhz = parse_smi(samples)
    return us, w, mhz


def main():
    dev = 'cuda'
    dt = torch.bfloat16
    torch.manual_seed(0)
    M, N = 8 * 96 * 320, 128
    cases = []
    for K in (5085, 5088, 1152):
        a = torch.randn(M, K, device=dev).to(dt)
        b = (torch.randn(K, N, device=dev) * 0.02).to(dt)
        c = torch.empty(M, N, device=dev, dtype=dt)
        cases.append(('gemm  [M=%d,K=%d]x[K,N=%d]' % (M, K, N), 2.0 * M * N * K, (lambda a=a, b=b, c=c: torch.matmul(a, b, out=c))))
        at = (torch.randn(N, K, device=dev) * 0.02).to(dt)
        bt = torch.randn(K, M, device=dev).to(dt)
        ct = torch.empty(N, M, device=dev, dtype=dt)
        cases.append(('gemmT [N=%d,K=%d]x[K,M=%d]' % (N, K, M), 2.0 * M * N * K, (lambda a=at, b=bt, c=ct: torch.matmul(a, b, out=c))))
        bn = torch.randn(M, K, device=dev).to(dt)            # pixels x K, weights N x K: C = A . W^T (the NT form)
        wn = (torch.randn(N, K, device=dev) * 0.02).to(dt)
        cn = torch.empty(M, N, device=dev, dtype=dt)
        cases.append(('gemmNT[M=%d,K=%d]x[N=%d,K]^T' % (M, K, N), 2.0 * M * N * K, (lambda a=bn, b=wn, c=cn: torch.matmul(a, b.t(), out=c))))
    sq = 8192
    a = torch.randn(sq, sq, device=dev).to(dt); b = torch.randn(sq, sq, device=dev).to(dt); c = torch.empty(sq, sq, device=dev, dtype=dt)
    cases.append(('gemm  8192^3', 2.0 * sq ** 3, (lambda a=a, b=b, c=c: torch.matmul(a, b, out=c))))
    az = torch.zeros(sq, sq, device=dev, dtype=dt)
    cases.append(('gemm  8192^3 zeros', 2.0 * sq ** 3, (lambda a=az, b=az, c=c: torch.matmul(a, b, out=c))))

    def conv_case(Cin, Cout, d, zeros=False):
        B, H, W = 8, 96, 320
        x = (torch.zeros if zeros else torch.randn)(B, Cin, H, W, device=dev).to(dt)
        w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02).to(dt)
        bias = torch.randn(Cout, device=dev)
        y = torch.empty(B, Cout, H, W, device=dev, dtype=dt)
        packed = ops.conv3x3_pack(w)
        return ('conv  %d->%d d%d [8,.,96,320]%s' % (Cin, Cout, d, ' zeros' if zeros else ''), 2.0 * B * H * W * Cin * Cout * 9,
                (lambda: ops.conv3x3_forward_raw(x, packed, bias, y, d, 0.1, 1, 3)))
    cases.append(conv_case(565, 128, 1))
    cases.append(conv_case(128, 128, 1))
    cases.append(conv_case(128, 128, 2))
    cases.append(conv_case(565, 128, 1, zeros=True))

    graphs = [(name, flop, graphed(fn)) for name, flop, fn in cases]
    lines = ['# tools/gemm_ceiling.py on %s, torch %s; us per launch (graph replay of %d), best / median of 8; then ~2 s back to back with rocm-smi sampling' %
             (torch.cuda.get_device_name(0), torch.__version__, NREP)]
    for rnd in range(2):                         # two alternated rounds: thermal state shared
        for name, flop, g in graphs:
            best, med = time_graph(g)
            lines.append('round %d  %-44s best %8.1f us %7.1f TFLOP/s | median %8.1f us %7.1f TFLOP/s' % (rnd, name, best, flop / best / 1e6, med, flop / med / 1e6))
            print(lines[-1], flush=True)
    for name, flop, g in graphs:
        us, w, mhz = sustained(g)
        lines.append('sustained %-44s %8.1f us %7.1f TFLOP/s  socket %6.0f W  sclk %5.0f MHz' % (name, us, flop / us / 1e6, w, mhz))
        print(lines[-1], flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    open('gpurun_out/gemm_ceiling.txt', 'w').write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
