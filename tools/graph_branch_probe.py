"""Does a captured hipGraph run independent branches concurrently?  Two chains of 60 small (coarse-level, ~10 us, a few dozen
workgroups) convolutions: captured on ONE stream (serial), and captured on two forked streams joined at the end (parallel branches
of one graph).  Also a latency-bound chain beside a fat one."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upflow_pytorch_amd import ops
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(1)

def mk(N, Cin, Cout, h, w):
    x = torch.randn(N, Cin, h, w, generator=g).to(dev).bfloat16()
    wt = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.02).to(dev).bfloat16()
    b = torch.zeros(Cout, device=dev)
    y = torch.empty(N, Cout, h, w, device=dev, dtype=torch.bfloat16)
    pk = ops.conv3x3_pack(wt)
    return lambda: ops.conv3x3_forward_raw(x, pk, b, y, 1, 0.1)

small_a, small_b = mk(8, 565, 128, 6, 20), mk(8, 565, 128, 6, 20)
fat = mk(8, 565, 128, 96, 320)
for f in (small_a, small_b, fat):
    f()
torch.cuda.synchronize()

def capture(fn):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    return gr

def timed(gr, n=30):
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6

def serial2():
    for _ in range(60):
        small_a()
    for _ in range(60):
        small_b()

def forked(fa, na, fb, nb):
    def fn():
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(nb):
                fb()
        for _ in range(na):
            fa()
        cur.wait_stream(side)
    return fn

print('60 + 60 small convolutions, one stream          : %8.1f us' % timed(capture(serial2)))
print('60 | 60 small convolutions, two branches        : %8.1f us' % timed(capture(forked(small_a, 60, small_b, 60))))
print('60 small convolutions alone                     : %8.1f us' % timed(capture(lambda: [small_a() for _ in range(60)])))
print('4 fat convolutions alone                        : %8.1f us' % timed(capture(lambda: [fat() for _ in range(4)])))
print('60 small then 4 fat, one stream                 : %8.1f us' % timed(capture(lambda: ([small_a() for _ in range(60)], [fat() for _ in range(4)]))))
print('60 small | 4 fat, two branches                  : %8.1f us' % timed(capture(forked(small_a, 60, fat, 4))))
