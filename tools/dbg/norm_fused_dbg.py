import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from upflow_pytorch_amd import ops as hip
for dtype in (torch.float16, torch.bfloat16):
  for shape in [(1, 32, 9, 33), (1, 40, 19, 70), (2, 120, 10, 64), (1, 196, 15, 45), (2, 32, 24, 40)]:
    B, C, H, W = shape
    g = torch.Generator().manual_seed(200 + sum(shape))
    pair = (torch.randn((2,) + shape, generator=g) * 1.7 + 0.3).to(dtype).cuda()
    normed = hip.normalize(pair.view(2 * B, C, H, W))
    want = hip.corr81_forward_raw(normed[:B], normed[B:])
    got = hip.corr81_norm_forward_raw(pair[0], pair[1])
    d = (got.float() - want.float()).abs()
    print(dtype, shape, 'ndiff', int((d > 0).sum()), 'max', float(d.max()))
    if d.max() > 0:
        idx = (d > 0).nonzero()[:5]
        print(idx.tolist())
        # delta probe: which normalised elements differ?  f1 = one-hot channel/pixel picks f2's normalised value
        x = pair.view(2 * B, C, H, W).float()
        mean = x.mean(dim=(2, 3), keepdim=True); var = x.var(dim=(2, 3), keepdim=True)
        emu = ((x - mean) * (1.0 / torch.sqrt(var + 1e-16))).to(dtype)
        print('normalize kernel vs torch emulation: ndiff', int((emu != normed).sum()), 'of', emu.numel())
        sub = normed.float().abs()
        print('min |normed| nonzero', float(sub[sub > 0].min()), 'count below 6.2e-5', int(((sub > 0) & (sub < 6.2e-5)).sum()))
