"""Packed-fp32 fault REPRODUCER (round 4; cited by _build.py): the cost volume beside the 16x16x32-MFMA narrow convolution on two streams, launch by launch bit comparison with an idle-GPU reference."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upflow_pytorch_amd import ops
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(3)
dt = torch.bfloat16
x = torch.randn(8, 184, 96, 320, generator=g).to(dt).to(dev)
w = (torch.randn(3, 184, 3, 3, generator=g) * 0.02).to(dt).to(dev)
b = torch.zeros(3, device=dev)
x8 = ops.to_c8(x); pk = ops.conv_c8_pack16(w, list(range(184))); yn = torch.empty(8, 3, 96, 320, dtype=dt, device=dev)
ofn = lambda: ops.conv_c8_forward_narrow_raw(x8, pk, b, yn, 0.1)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
B, C, H, W = 8, 32, 96, 320
f1 = torch.randn(B, C, H, W, generator=g).to(dt).to(dev); f2 = torch.randn(B, C, H, W, generator=g).to(dt).to(dev)
o1 = torch.empty(B, 81, H, W, dtype=dt, device=dev)
fn = lambda: ops.corr81_norm_forward_raw(f1, f2, out=o1, leaky_slope=0.1)
fn(); torch.cuda.synchronize(); ref = o1.clone()
bad = 0; maxd = 0; tiles = set()
for it in range(30):
    with torch.cuda.stream(sB):
        for _ in range(12):
            ofn()
    with torch.cuda.stream(sA):
        fn()
    torch.cuda.synchronize()
    if not torch.equal(o1, ref):
        bad += 1
        d = (o1.float() - ref.float()).abs()
        maxd = max(maxd, float(d.max()))
        nz = (d > 0).nonzero()
        if it < 3:
            print('   iter', it, '#diff', len(nz), 'items', sorted(set(nz[:, 0].tolist())), 'channels', len(set(nz[:, 1].tolist())), 'rows%8', sorted(set((nz[:, 2] % 8).tolist())), 'tile rows', sorted(set((nz[:, 2] // 8).tolist()))[:12], 'tile cols', sorted(set((nz[:, 3] // 32).tolist())))
print(sys.argv[1] if len(sys.argv) > 1 else '', 'mismatches %d/30 max diff %.3g' % (bad, maxd))
