"""The fused-normalisation cost volume under concurrency: is its output independent of what else runs on the GPU?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upflow_pytorch_amd import ops
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(3)

def mk_conv(dt, Cin, Cout, H, W, B=8):
    x = torch.randn(B, Cin, H, W, generator=g).to(dt).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dt).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    pk = ops.conv3x3_pack(w)
    y = torch.empty(B, Cout, H, W, dtype=dt, device=dev)
    return lambda: ops.conv3x3_forward_raw(x, pk, b, y, 1, 0.1)

def mk_narrow(dt, Cin, Cout, H, W, y_c8, B=8):
    x = torch.randn(B, Cin, H, W, generator=g).to(dt).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dt).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    x8 = ops.to_c8(x)
    pk = ops.conv_c8_pack16(w, list(range(Cin)))
    y = ops.c8_empty(B, Cout, H, W, dt, dev) if y_c8 else torch.empty(B, Cout, H, W, dtype=dt, device=dev)
    return lambda: ops.conv_c8_forward_narrow_raw(x8, pk, b, y, 0.1)

def mk_c8(dt, Cin, Cout, H, W, B=8, d=1):
    x = torch.randn(B, Cin, H, W, generator=g).to(dt).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dt).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    x8 = ops.to_c8(x)
    pk = ops.conv_c8_pack(w, list(range(Cin)))
    y = ops.c8_empty(B, Cout, H, W, dt, dev)
    return lambda: ops.conv_c8_forward_raw(x8, None, pk, b, y, d, 0.1)

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
for dt in (torch.bfloat16, torch.float16):
    others = {'narrow184->3': mk_narrow(dt, 184, 3, 96, 320, False), 'narrow176->8': mk_narrow(dt, 176, 8, 96, 320, True), 'narrow160->16@48': mk_narrow(dt, 160, 16, 48, 160, True),
              'c8_64->32': mk_c8(dt, 64, 32, 96, 320), 'c8_128->128d4': mk_c8(dt, 128, 128, 96, 320, d=4), 'c8_568->128': mk_c8(dt, 568, 128, 96, 320)}
    for (B, C, H, W) in [(8, 32, 96, 320), (8, 64, 48, 160)]:
        f1 = torch.randn(B, C, H, W, generator=g).to(dt).to(dev)
        f2 = torch.randn(B, C, H, W, generator=g).to(dt).to(dev)
        variants = {}
        if W % 8 == 0 and H * W >= 48 * 160:
            out8 = ops.c8_empty(B, 88, H, W, dt, dev)
            variants['norm_c8'] = (lambda out8=out8: ops.corr81_norm_forward_c8(f1, f2, out8, 0.1), out8)
        o1 = torch.empty(B, 81, H, W, dtype=dt, device=dev)
        variants['norm'] = (lambda o1=o1: ops.corr81_norm_forward_raw(f1, f2, out=o1, leaky_slope=0.1), o1)
        o2 = torch.empty(B, 81, H, W, dtype=dt, device=dev)
        variants['plain'] = (lambda o2=o2: ops.corr81_forward_raw(f1, f2, out=o2, leaky_slope=0.1), o2)
        for vn, (fn, y) in variants.items():
            fn(); torch.cuda.synchronize()
            ref = y.clone()
            res = []
            for oname, ofn in others.items():
                bad = 0
                for it in range(40):
                    with torch.cuda.stream(sB):
                        for _ in range(12):
                            ofn()
                    with torch.cuda.stream(sA):
                        y.zero_()
                        fn()
                    torch.cuda.synchronize()
                    bad += int(not torch.equal(y, ref))
                res.append('%s %d/40' % (oname, bad))
            print('%-8s %-8s [%d,%d,%d,%d]  mismatches beside: %s' % (str(dt).split('.')[-1], vn, B, C, H, W, '  '.join(res)), flush=True)
