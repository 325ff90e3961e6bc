"""Which intermediate differs first when two captured steps run concurrently?  (net._taps keeps references to the fast schedule's
buffers; no extra launches.)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from upflow_pytorch_amd import synthetic
from upflow_pytorch_amd.runtime import GraphedInference
dev = torch.device('cuda', 0)
B, H, W = 4, 384, 1280
a, b = synthetic.make_smooth_images(2, 2, H, W)
idx = [1, 0, 0, 1]
a, b = a[idx].contiguous().to(dev), b[idx].contiguous().to(dev)
net = bench.build_net(torch.bfloat16, dev)
runners, taps = [], []
for i in range(3):
    net._taps = []
    r = GraphedInference(net, B, H, W, device=dev, warmup=3 if i == 0 else 1)
    # the capture ran last: its taps are the graph's buffers (the warm-up taps precede them)
    n = len(net._taps) // (4 if i == 0 else 2)
    taps.append(net._taps[-n:])
    runners.append(r)
    r.load(a, b)
net._taps = None
torch.cuda.synchronize()
ref_r, ref_t = runners[0], taps[0]
ref_r.replay(); torch.cuda.synchronize()
ref_vals = [(n_, t.clone()) for n_, t in ref_t]
ref_out = ref_r.out['flow_f_out'].clone()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
first = {}
nbad = 0
for it in range(40):
    with torch.cuda.stream(s1):
        runners[1].replay()
    with torch.cuda.stream(s2):
        runners[2].replay()
    torch.cuda.synchronize()
    for ri in (1, 2):
        if torch.equal(runners[ri].out['flow_f_out'], ref_out):
            continue
        nbad += 1
        for (n_, t), (rn, rv) in zip(taps[ri], ref_vals):
            assert n_ == rn
            if not torch.equal(t, rv):
                d = (t.float() - rv.float()).abs()
                key = n_
                if key not in first:
                    first[key] = 0
                first[key] += 1
                if sum(first.values()) <= 6:
                    nz = (d > 0).nonzero()
                    print('iter %d runner %d: first differing tap %-18s shape %s  max diff %.3g  #diff %d  first idx %s last idx %s' %
                          (it, ri, n_, tuple(t.shape), float(d.max()), int((d > 0).sum()), nz[0].tolist(), nz[-1].tolist()), flush=True)
                    if n_.endswith('buf8'):
                        rng = {'conv5': (0, 4), 'conv4': (4, 12), 'conv3': (12, 24), 'conv2': (24, 40), 'conv1': (40, 56), 'corr': (56, 67), 'feat': (67, 71), 'flow': (71, 72), 'refined': (72, 73)}
                        print('      per range #diff:', {k: int((d[:, lo:hi] > 0).sum()) for k, (lo, hi) in rng.items()})
                        dc = d[:, 56:67]
                        if float(dc.max()) > 0:
                            nzc = (dc > 0).nonzero()
                            print('      corr diffs: items', sorted(set(nzc[:, 0].tolist())), 'octets', sorted(set(nzc[:, 1].tolist())), 'rows', int(nzc[:, 2].min()), '-', int(nzc[:, 2].max()), 'cols', int(nzc[:, 3].min()), '-', int(nzc[:, 3].max()), 'max', float(dc.max()))
                break
print('mismatching replays: %d of 80; first differing tap histogram: %s' % (nbad, first))
