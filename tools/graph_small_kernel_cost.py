#!/usr/bin/env python3
"""What ONE small dependent kernel costs inside a replayed hipGraph (single stream): chains of N scalar adds, N 1-workgroup
elementwise ops on 4x2x256x832 fp32 (an autograd accumulation), captured and replayed."""
import time, torch
dev = torch.device('cuda')
def run(make, n, reps=20):
    x = make()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            y = x
            for _ in range(8): y = y + 1.0
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            y = x
            for _ in range(n): y = y + 1.0
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps / n * 1e6
for name, make in (('scalar add', lambda: torch.zeros((), device=dev)),
                   ('add 4x2x256x832 fp32 (6.8 MB r+w 13.6 MB)', lambda: torch.zeros(4, 2, 256, 832, device=dev)),
                   ('add 8x128x8x26 bf16', lambda: torch.zeros(8, 128, 8, 26, device=dev, dtype=torch.bfloat16))):
    for n in (100, 1000):
        print('%-45s chain of %4d: %.2f us per kernel' % (name, n, run(make, n)), flush=True)
