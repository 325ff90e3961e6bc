#!/usr/bin/env python3
"""Do CU-masked streams help several steps in flight?  runtime.PipelinedInference shares all 256 CUs dynamically between its
streams; here each stream's HW queue is restricted to a disjoint share of the CUs (hipExtStreamCreateWithCUMask), in two ways:
contiguous mask bits, and mask bits interleaved modulo the number of streams (whichever the driver maps onto whole XCDs)."""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from upflow_pytorch_amd import synthetic
from upflow_pytorch_amd.runtime import GraphedInference

hip = ctypes.CDLL('libamdhip64.so')
dev = torch.device('cuda', 0)
B, H, W = 4, 384, 1280
net = bench.build_net(torch.bfloat16, dev)
a, b = [t.to(dev) for t in synthetic.make_smooth_images(5, B, H, W)]
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    words = (ctypes.c_uint32 * ((NCU + 31) // 32))()
    for i in bits:
        words[i // 32] |= (1 << (i % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def run(streams, steps=60, warm=12):
    runners = [GraphedInference(net, B, H, W, device=dev, warmup=3 if i == 0 else 1) for i in range(len(streams))]
    for r in runners:
        r.load(a, b)
    torch.cuda.synchronize()

    def go(n):
        for i in range(n):
            k = i % len(streams)
            with torch.cuda.stream(streams[k]):
                runners[k].replay()
    go(warm)
    torch.cuda.synchronize()
    t = time.perf_counter()
    go(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


print('CUs', NCU)
for n in (2, 4):
    plain = [torch.cuda.Stream(device=dev) for _ in range(n)]
    per = NCU // n
    contiguous = [masked_stream(range(k * per, (k + 1) * per)) for k in range(n)]
    interleaved = [masked_stream([i for i in range(NCU) if i % n == k]) for k in range(n)]
    xcd = [masked_stream([i for i in range(NCU) if (i % 8) * n // 8 == k]) for k in range(n)]
    res = {name: run(st) for name, st in (('plain', plain), ('contiguous mask', contiguous), ('interleaved mask', interleaved), ('mask by (bit % 8)', xcd))}
    print('%d streams: ' % n + '   '.join('%s %.3f ms/step' % kv for kv in res.items()), flush=True)
