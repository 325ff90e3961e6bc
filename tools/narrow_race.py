"""Is a kernel's output independent of what else runs on the GPU?  Each candidate kernel is replayed on stream A while stream B
runs another kernel concurrently; every output is compared with the kernel's own result when run alone."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upflow_pytorch_amd import ops
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(3)

def mk_narrow(dt, Cin, Cout, H, W, y_c8, B=8):
    x = torch.randn(B, Cin, H, W, generator=g).to(dt).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dt).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    x8 = ops.to_c8(x)
    pk = ops.conv_c8_pack16(w, list(range(Cin)))
    y = ops.c8_empty(B, Cout, H, W, dt, dev) if y_c8 else torch.empty(B, Cout, H, W, dtype=dt, device=dev)
    return (lambda: ops.conv_c8_forward_narrow_raw(x8, pk, b, y, 0.1)), y

def mk_c8(dt, Cin, Cout, H, W, B=8, d=1):
    x = torch.randn(B, Cin, H, W, generator=g).to(dt).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dt).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    x8 = ops.to_c8(x)
    pk = ops.conv_c8_pack(w, list(range(Cin)))
    y = ops.c8_empty(B, Cout, H, W, dt, dev)
    return (lambda: ops.conv_c8_forward_raw(x8, None, pk, b, y, d, 0.1)), y

def mk_nchw(dt, Cin, Cout, H, W, B=8):
    x = torch.randn(B, Cin, H, W, generator=g).to(dt).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dt).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    pk = ops.conv3x3_pack(w)
    y = torch.empty(B, Cout, H, W, dtype=dt, device=dev)
    return (lambda: ops.conv3x3_forward_raw(x, pk, b, y, 1, 0.1)), y

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
for dt in (torch.bfloat16, torch.float16):
    cands = {'narrow 184->3 nchw 96x320': mk_narrow(dt, 184, 3, 96, 320, False), 'narrow 176->8 c8 96x320': mk_narrow(dt, 176, 8, 96, 320, True),
             'narrow 160->16 c8 96x320': mk_narrow(dt, 160, 16, 96, 320, True), 'narrow 184->3 nchw 48x160': mk_narrow(dt, 184, 3, 48, 160, False),
             'c8 64->32 96x320': mk_c8(dt, 64, 32, 96, 320), 'c8 128->32 96x320': mk_c8(dt, 128, 32, 96, 320), 'c8 128->128 d4': mk_c8(dt, 128, 128, 96, 320, d=4),
             'nchw 565->128 48x160': mk_nchw(dt, 565, 128, 48, 160)}
    others = {'wide nchw 565->128 96x320': mk_nchw(dt, 565, 128, 96, 320)[0], 'stem 3->16 384x1280': mk_nchw(dt, 3, 16, 384, 1280)[0],
              'coarse 565->128 6x20': mk_nchw(dt, 565, 128, 6, 20)[0]}
    for name, (fn, y) in cands.items():
        fn(); torch.cuda.synchronize()
        ref = y.clone()
        res = []
        for oname, ofn in others.items():
            bad = 0
            for it in range(30):
                with torch.cuda.stream(sB):
                    for _ in range(6):
                        ofn()
                with torch.cuda.stream(sA):
                    y.fill_(7.0)
                    fn()
                torch.cuda.synchronize()
                bad += int(not torch.equal(y, ref))
            res.append('%s: %d/30' % (oname.split()[0], bad))
        print('%-8s %-28s mismatches beside  %s' % (str(dt).split('.')[-1], name, '   '.join(res)), flush=True)
