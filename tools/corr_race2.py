"""Packed-fp32 bisect (round 4): the fused-normalisation cost volume beside conv kernels on a second stream — which neighbour kernel makes its output change?"""
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
    return lambda: ops.conv_c8_forward_narrow_raw(x8, pk, b, y, 0.1)

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
others = {'narrow_bf16': mk_narrow(torch.bfloat16, 184, 3, 96, 320, False), 'narrow_fp16': mk_narrow(torch.float16, 184, 3, 96, 320, False)}
for dt in (torch.bfloat16, torch.float16):
    B, C, H, W = 8, 32, 96, 320
    f1 = torch.randn(B, C, H, W, generator=g).to(dt).to(dev)
    f2 = torch.randn(B, C, H, W, generator=g).to(dt).to(dev)
    o1 = torch.empty(B, 81, H, W, dtype=dt, device=dev)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    cands = {'normalize(stats+apply)': (lambda: ops.normalize(f1), None),
             'corr81_norm': (lambda: ops.corr81_norm_forward_raw(f1, f2, out=o1, leaky_slope=0.1), o1),
             'warp': (lambda: ops.warp(f1, torch.full((B, 2, H, W), 0.37, device=dev), 'robust'), None),
             'flow_upsample': (lambda: ops.flow_upsample(torch.full((B, 2, H // 2, W // 2), 0.37, device=dev) + f1[:, :2, ::2, ::2].float(), H, W, True), None)}
    for cn, (fn, y) in cands.items():
        r = fn(); torch.cuda.synchronize()
        ref = (y if y is not None else r).clone()
        res = []
        for oname, ofn in others.items():
            bad = 0; maxd = 0.0
            for it in range(30):
                with torch.cuda.stream(sB):
                    for _ in range(12):
                        ofn()
                with torch.cuda.stream(sA):
                    r = fn()
                torch.cuda.synchronize()
                out = y if y is not None else r
                if not torch.equal(out, ref):
                    bad += 1
                    maxd = max(maxd, float((out.float() - ref.float()).abs().max()))
            res.append('%s %d/30 (max diff %.3g)' % (oname, bad, maxd))
        print('%-8s %-24s beside: %s' % (str(dt).split('.')[-1], cn, '   '.join(res)), flush=True)
