"""Experiment: config 2 (batch 4) as ONE graph vs TWO graphs of batch 2 replayed on two streams concurrently."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from upflow_pytorch_amd.runtime import GraphedInference

dev = torch.device('cuda', 0)
net = bench.build_net(torch.bfloat16, dev)
H, W = 384, 1280


def timed(fn, n=50, w=10):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


one = GraphedInference(net, 4, H, W, device=dev)
print('one graph, batch 4: %.3f ms' % timed(one.replay), flush=True)
for parts in (2, 4):
    runners = [GraphedInference(net, 4 // parts, H, W, device=dev) for _ in range(parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]

    def both():
        cur = torch.cuda.current_stream(dev)
        for r, s in zip(runners, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                r.replay()
        for s in streams:
            cur.wait_stream(s)
    print('%d graphs of batch %d on %d streams: %.3f ms' % (parts, 4 // parts, parts, timed(both)), flush=True)

    def serial():
        for r in runners:
            r.replay()
    print('%d graphs of batch %d, one stream: %.3f ms' % (parts, 4 // parts, timed(serial)), flush=True)
