import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from upflow_pytorch_amd import ops as hip, _lib
for dtype in (torch.float16, torch.bfloat16):
  for shape in [(2, 32, 240, 720), (1, 32, 240, 720), (2, 32, 240, 704), (2, 32, 96, 320), (1, 32, 96, 320), (1, 8, 96, 320), (1, 8, 240, 720), (1, 4, 512, 512), (4, 4, 512, 512), (1, 32, 64, 64)]:
    B, C, H, W = shape
    g = torch.Generator().manual_seed(200 + sum(shape))
    pair = (torch.randn((2,) + shape, generator=g) * 1.7 + 0.3).to(dtype).cuda()
    normed = hip.normalize(pair.view(2 * B, C, H, W))
    want = hip.corr81_forward_raw(normed[:B], normed[B:])
    got = hip.corr81_norm_forward_raw(pair[0], pair[1])
    nseg = _lib.lib().upf_normalize_workspace_bytes(2 * B * C, H * W) // (2 * B * C * 12)
    print(dtype, shape, 'nseg', nseg, 'ndiff', int((got != want).sum()))
