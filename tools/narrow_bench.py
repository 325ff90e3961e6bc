"""The Cout <= 16 octet layers on the 16-channel MFMA kernel (upf_conv_forward_c8_narrow) vs the 32-channel kernel, per layer at the config-2 level shapes."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
import conv_layers as CL
from upflow_pytorch_amd import ops
B = 8
for (H, W) in [(96, 320), (48, 160)]:
    for (name, Cin, Cout, yc8) in [('est.conv_last', 568, 2, False), ('sgu.conv_last', 184, 3, False), ('sgu.conv5', 176, 8, True), ('sgu.conv4', 160, 16, True), ('ctx.conv6', 32, 2, False)]:
        dt = torch.bfloat16
        x8 = torch.randn(B, Cin // 8, H, W, 8, device='cuda').to(dt)
        w = (torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.02).to(dt)
        b = torch.randn(Cout, device='cuda')
        y = ops.c8_empty(B, Cout, H, W, dt, 'cuda') if yc8 else torch.empty(B, Cout, H, W, device='cuda', dtype=dt)
        cmap = list(range(Cin))
        p32, p16 = ops.conv_c8_pack(w, cmap), ops.conv_c8_pack16(w, cmap)
        t32 = CL.graph_time(lambda: ops.conv_c8_forward_raw(x8, None, p32, b, y, 1, 0.1))
        res = []
        for abl in (0, 1, 2):
            prev = ops.conv_set_option('ablate', abl)
            res.append(CL.graph_time(lambda: ops.conv_c8_forward_narrow_raw(x8, p16, b, y, 0.1)))
            ops.conv_set_option('ablate', prev)
        print('%dx%d %-14s %3d->%2d  32-channel kernel %6.1f us | narrow %6.1f us (staging only %5.1f, matrix only %5.1f)' % (H, W, name, Cin, Cout, t32, res[0], res[1], res[2]), flush=True)
