#!/usr/bin/env python3
"""Is a training step bit-reproducible run to run?  Two fresh trainers, the same weights and batch: which parameter gradients of
the FIRST step differ, and which backward operator makes them differ (each operator twice on the same inputs)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_hip_train as T
import _weights
from upflow_pytorch_amd import ops

batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
gs = []
for r in range(2):
    tr = T._config3_trainer('bf16', False)
    tr.net.train()
    b = dict(batch); b['if_loss'] = True
    out = tr.net(b)
    loss, parts = tr.loss_manager.compute_loss(out)
    loss.backward()
    gs.append({n: p.grad.clone() for n, p in tr.raw_net.named_parameters()})
bad = [n for n in gs[0] if not torch.equal(gs[0][n], gs[1][n])]
print('parameters whose first-step gradient differs between two runs: %d of %d' % (len(bad), len(gs[0])))
for n in bad:
    d = (gs[0][n] - gs[1][n]).abs().max() / gs[0][n].abs().max()
    print('   %-55s max rel diff %.2e' % (n, float(d)))

g = torch.Generator().manual_seed(1)
def rnd(*s, dt=torch.bfloat16):
    return torch.randn(*s, generator=g).to(dt).cuda()
def twice(name, fn):
    a = fn(); b = fn()
    a = a if isinstance(a, (tuple, list)) else (a,)
    b = b if isinstance(b, (tuple, list)) else (b,)
    same = all(torch.equal(x, y) for x, y in zip(a, b) if x is not None)
    print('%-40s %s' % (name, 'bit-identical' if same else 'DIFFERS'))
for (B, C, H, W) in [(8, 32, 64, 208), (8, 64, 32, 104), (8, 196, 4, 13)]:
    f1, f2, go = rnd(B, C, H, W), rnd(B, C, H, W), rnd(B, 81, H, W)
    twice('corr81_backward %s' % ((B, C, H, W),), lambda: ops.corr81_backward_raw(f1, f2, go))
    x, flow, gy = rnd(B, C, H, W), (torch.randn(B, 2, H, W, generator=g) * 3).cuda(), rnd(B, C, H, W)
    def wb():
        xx = x.clone().requires_grad_(True); ff = flow.clone().requires_grad_(True)
        y = ops.warp(xx, ff, 'literal', B // 2); y.backward(gy); return xx.grad, ff.grad
    twice('warp backward %s' % ((B, C, H, W),), wb)
    def nb():
        xx = x.clone().requires_grad_(True); y = ops.normalize(xx); y.backward(gy); return xx.grad
    twice('normalize backward %s' % ((B, C, H, W),), nb)

# ---- the other backward operators, and every convolution gradient form, each twice on the same inputs; between the two
# calls the allocator's free blocks are poisoned with NaN (an operator that reads memory it did not write shows up as NaN / a difference)
def poison():
    t = torch.full((256 << 20,), float('nan'), dtype=torch.float32, device='cuda')
    del t
def twice_p(name, fn):
    a = fn(); poison(); b = fn()
    a = a if isinstance(a, (tuple, list)) else (a,)
    b = b if isinstance(b, (tuple, list)) else (b,)
    same = all(torch.equal(x, y) for x, y in zip(a, b) if x is not None)
    fin = all(bool(torch.isfinite(y.float()).all()) for y in b if y is not None)
    print('%-60s %s%s' % (name, 'bit-identical' if same else 'DIFFERS', '' if fin else '  NON-FINITE'))

B, H, W = 8, 64, 208
fl = (torch.randn(B, 2, H // 2, W // 2, generator=g) * 2).cuda()
gy2 = torch.randn(B, 2, H, W, generator=g).cuda()
def fu():
    x = fl.clone().requires_grad_(True); y = ops.flow_upsample(x, H, W, True); y.backward(gy2); return x.grad
twice_p('flow_upsample backward', fu)
xo = rnd(B, 3, H, W)
fi = (torch.randn(B, 2, H, W, generator=g) * 2).cuda()
def sb():
    a = fi.clone().requires_grad_(True); b = xo.clone().requires_grad_(True)
    _, up, _, _ = ops.sgu_blend(a, b); up.backward(gy2); return a.grad, b.grad
twice_p('sgu_blend backward (level)', sb)
fi4 = (torch.randn(4, 2, 256, 832, generator=g) * 2).cuda(); xo4 = rnd(4, 3, 64, 208); gy4 = torch.randn(4, 2, 256, 832, generator=g).cuda()
fl4 = (torch.randn(4, 2, 64, 208, generator=g) * 2).cuda()
def sb4():
    a = fl4.clone().requires_grad_(True); b = xo4.clone().requires_grad_(True); c = fi4.clone().requires_grad_(True)
    _, up, _, _ = ops.sgu_blend(a, b, output_level_flow=c); up.backward(gy4); return a.grad, b.grad, c.grad
twice_p('sgu_blend backward (final level)', sb4)
g1, g2, gd = torch.rand(4, 1, 256, 832, generator=g).cuda(), torch.rand(4, 1, 256, 832, generator=g).cuda(), torch.randn(4, 1, 256, 832, generator=g).cuda()
def cb():
    a = g1.clone().requires_grad_(True); b = g2.clone().requires_grad_(True); d = ops.census_distance(a, b, 3); d.backward(gd); return a.grad, b.grad
twice_p('census backward', cb)
im, imr = torch.rand(4, 3, 256, 832, generator=g).cuda(), torch.rand(4, 3, 288, 864, generator=g).cuda()
start = torch.tensor([[16., 16.]] * 4).view(4, 2, 1, 1).cuda()
def bw():
    f = fi4.clone().requires_grad_(True); y = ops.boundary_warp(imr, f, start); y.backward(torch.ones_like(y)); return f.grad
twice_p('boundary_warp backward', bw)
occ = (torch.rand(4, 1, 256, 832, generator=g) > 0.3).float().cuda()
def rl():
    a = im.clone().requires_grad_(True); s, so = ops.robust_loss_sums(a, im * 0.9, occ); s.backward(); return a.grad, s.detach()
twice_p('robust_loss forward + backward', rl)
def se():
    f = fi4.clone().requires_grad_(True); s = ops.smooth_edge1(im, f); s.backward(); return f.grad, s.detach()
twice_p('smooth_edge1 forward + backward', se)

import torch.nn as nn
for (Cin, Cout, d, stride, hh, ww) in [(3, 16, 1, 2, 256, 832), (16, 16, 1, 1, 128, 416), (64, 96, 1, 2, 32, 104), (128, 128, 2, 1, 64, 208), (128, 96, 8, 1, 64, 208),
                                       (96, 64, 16, 1, 32, 104), (32, 2, 1, 1, 64, 208), (565, 128, 1, 1, 32, 104), (196, 32, 0, 1, 4, 13), (32, 32, 0, 1, 64, 208)]:
    k = 1 if d == 0 else 3
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).cuda(); bsv = torch.randn(Cout, generator=g).cuda()
    x = rnd(8, Cin, hh, ww)
    def cv():
        xx = x.clone().requires_grad_(True); wq = w.clone().requires_grad_(True); bq = bsv.clone().requires_grad_(True)
        y = ops.conv_train(xx, wq, bq, max(d, 1), 0.1, stride)
        gg = torch.ones_like(y) * 0.01 + (y.detach() * 0.001)
        y.backward(gg); return xx.grad, wq.grad, bq.grad
    twice_p('conv_train %d->%d k%d d%d s%d %dx%d (gx, gw, gb)' % (Cin, Cout, k, d, stride, hh, ww), cv)
