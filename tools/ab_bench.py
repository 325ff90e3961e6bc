#!/usr/bin/env python3
"""A/B of convolution launch options on ONE box, in ONE process: the config-2 inference step is captured once per option set
(launch shapes are baked into a hipGraph at capture time) and the graphs are replayed alternately, so box-to-box and
thermal differences cancel.    python tools/ab_bench.py "" "no_c8=1" ["sk_grid=96,ph_fit=0" ...]    ("" = defaults; no_c8: model switch)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from upflow_pytorch_amd import ops, synthetic
from upflow_pytorch_amd.runtime import GraphedInference

B, H, W, dname = bench.WORKLOADS[os.environ.get('UPF_AB_WORKLOAD', 'config2')]
dev = torch.device('cuda', 0)
net = bench.build_net(bench.DT[dname], dev)
im1, im2 = synthetic.make_images(2, B, H, W)
im1, im2 = im1.to(dev), im2.to(dev)
variants = sys.argv[1:] or ['', '']
STREAMS = int(os.environ.get('UPF_AB_STREAMS', '1'))
runners = []
for v in variants:
    opts = dict(kv.split('=') for kv in v.split(',') if kv)
    net._no_c8 = bool(int(opts.pop('no_c8', 0)))          # (model switch, not a library option: NCHW at every level)
    net._no_c8_est = bool(int(opts.pop('no_c8_est', 0)))  # (model switch: the flow estimator of the fine levels in NCHW)
    from upflow_pytorch_amd.model import pwc_modules
    pwc_modules._NO_NARROW[0] = bool(int(opts.pop('no_narrow', 0)))   # (Cout <= 16 octet layers on the 32-channel kernel)
    pwc_modules.MERGE_TAIL[0] = not bool(int(opts.pop('no_merge', 0)))  # (merged narrow tails of the octet stacks, round 6)
    pwc_modules.FUSE_PAIRS[0] = not bool(int(opts.pop('no_pairs', 0)))  # (fused SGU guidance stem, round 6)
    from upflow_pytorch_amd.model import upflow as _mu
    _mu.FLOW16_IN_BLEND[0] = not bool(int(opts.pop('no_flow16', 0)))    # (the blend stores the estimator's flow slot itself, round 6)
    pwc_modules.DUAL_1X1[0] = not bool(int(opts.pop('no_dual', 0)))     # (1x1 projection stored into both stacks' buffers by one launch, round 6)
    for m in net.modules():
        m.__dict__.pop('_packed8', None)                  # (packed operands are cached per module: rebuild for this variant)
        m.__dict__.pop('_fast_cache', None)
    prev = {k: ops.conv_set_option(k, int(val)) for k, val in opts.items()}
    if STREAMS > 1:                                       # UPF_AB_STREAMS=4: the headline's form, S captured steps in flight on S streams
        from upflow_pytorch_amd.runtime import PipelinedInference
        pipe = PipelinedInference(net, B, H, W, streams=STREAMS, device=dev)
        tickets = [pipe.submit(im1, im2) for _ in range(STREAMS)]

        class _R(object):
            def __init__(self, pipe, tickets):
                self.pipe, self.tickets = pipe, tickets

            def replay(self):
                for t_ in self.tickets:
                    self.pipe.replay(t_)
        r = _R(pipe, tickets)
    else:
        r = GraphedInference(net, B, H, W, device=dev)
        r.load(im1, im2)
    r.replay(); torch.cuda.synchronize()
    runners.append(r)
    for k, val in prev.items():
        ops.conv_set_option(k, val)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    for r in runners:
        r.replay()
    torch.cuda.synchronize()
best = [1e9] * len(runners); tot = [0.0] * len(runners)
ROUNDS, N = 12, 20
for _ in range(ROUNDS):
    for i, r in enumerate(runners):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(N):
            r.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / N
        best[i] = min(best[i], dt); tot[i] += dt
for i, v in enumerate(variants):
    print('%-40s mean %.3f ms (%.1f pairs/s)   best %.3f ms%s' % (v or '(defaults)', tot[i] / ROUNDS * 1e3, STREAMS * B / (tot[i] / ROUNDS), best[i] * 1e3, ' [%d steps in flight per replay]' % STREAMS if STREAMS > 1 else ''), flush=True)
