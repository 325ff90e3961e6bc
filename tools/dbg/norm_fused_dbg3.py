import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from upflow_pytorch_amd import ops as hip, _lib
dtype = torch.float16
shape = (2, 32, 240, 720)
B, C, H, W = shape
g = torch.Generator().manual_seed(200 + sum(shape))
pair = (torch.randn((2,) + shape, generator=g) * 1.7 + 0.3).to(dtype).cuda()
x = pair.view(2 * B, C, H, W)
y = torch.empty_like(x)
N = 2 * B * C
mean = torch.empty(N, device='cuda'); rstd = torch.empty(N, device='cuda')
ws = torch.empty(_lib.lib().upf_normalize_workspace_bytes(N, H * W), dtype=torch.uint8, device='cuda')
_lib.call('upf_normalize_forward', _lib.ptr(x), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(ws), N, H * W, _lib.dtype_code(x), _lib.stream_ptr(x.device))
emu = ((x.float() - mean.view(2 * B, C, 1, 1)) * rstd.view(2 * B, C, 1, 1)).to(dtype)
print('unfused normalize vs emulation from its own mean/rstd: ndiff', int((emu != y).sum()))
want = hip.corr81_forward_raw(y[:B], y[B:])
got = hip.corr81_norm_forward_raw(pair[0], pair[1])
idx = (got != want).nonzero().tolist()
f2loc, f1loc = collections.Counter(), collections.Counter()
for n, ch, yy, xx in idx:
    dy, dx = ch // 9 - 4, ch % 9 - 4
    f2loc[(n, yy + dy, xx + dx)] += 1
    f1loc[(n, yy, xx)] += 1
print('f2 candidates', f2loc.most_common(8))
print('f1 candidates', f1loc.most_common(4))
wsf = ws.view(torch.float32).view(N, -1, 3).cpu()
print('nseg', wsf.shape[1])
for (n, yy, xx), cnt in f2loc.most_common(4):
    r = (B + n) * C
    print('f2 item', n, 'pos', yy, xx, 'count', cnt, ' raw', x[B + n, :4, yy, xx].tolist(), 'normed', y[B + n, :4, yy, xx].tolist())
