"""Packed-fp32 bisect (round 4): the same pair of kernels as corr_race3.py under the cost-volume kernel's ablation switches (UPF_ALLC_ABL builds), to find the instruction class involved."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upflow_pytorch_amd import ops
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(3)
def narrow(dt):
    x = torch.randn(8, 184, 96, 320, generator=g).to(dt).to(dev)
    w = (torch.randn(3, 184, 3, 3, generator=g) * 0.02).to(dt).to(dev)
    b = torch.zeros(3, device=dev)
    x8 = ops.to_c8(x); pk = ops.conv_c8_pack16(w, list(range(184))); yn = torch.empty(8, 3, 96, 320, dtype=dt, device=dev)
    return lambda: ops.conv_c8_forward_narrow_raw(x8, pk, b, yn, 0.1)
ofn = narrow(torch.bfloat16)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
for dt in (torch.bfloat16, torch.float16):
    for shape in [(16, 32, 96, 320), (8, 32, 96, 320), (32, 32, 96, 320), (16, 64, 48, 160)]:
        x = torch.randn(*shape, generator=g).to(dt).to(dev)
        fn = lambda: ops.normalize(x)
        ref = fn().clone(); torch.cuda.synchronize()
        bad = 0; maxd = 0.0
        for it in range(30):
            with torch.cuda.stream(sB):
                for _ in range(12):
                    ofn()
            with torch.cuda.stream(sA):
                r = fn()
            torch.cuda.synchronize()
            if not torch.equal(r, ref):
                bad += 1; maxd = max(maxd, float((r.float() - ref.float()).abs().max()))
        print('%s normalize %s: mismatches %d/30 max diff %.3g' % (str(dt).split('.')[-1], shape, bad, maxd), flush=True)
