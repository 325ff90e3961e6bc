#!/usr/bin/env python3
"""The fp32 parity mode under every convolution back end (config `fp32_conv`): whole-net EPE vs the REFERENCE's goldens at
realistic motion (head_scale 1: tests/golden/net_*_hs1_robust.npz) and at the small-motion headline vector, per-layer error of the
565->128 layer vs fp64, and the step time at config 2's shape (384x1280, B=4, hipGraph).   python tools/x3_modes.py > profiles/r04_fp32_conv_modes.txt"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upflow_pytorch_amd import ops, synthetic  # noqa: E402
from upflow_pytorch_amd.model.upflow import UPFlow_net  # noqa: E402
from upflow_pytorch_amd.model.pwc_modules import fp32_conv_mode, FP32_CONV_NPROD  # noqa: E402
from upflow_pytorch_amd.runtime import GraphedInference  # noqa: E402

FLAGS = {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
         'norm_moments_across_images': False, 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}


def build(mode, hs):
    conf = UPFlow_net.config()
    conf.update(dict(FLAGS, fp32_conv=mode), verbose=False)
    net = conf()
    net.load_state_dict(synthetic.make_state_dict(0, head_scale=hs))
    return net.cuda().eval()


def epe(a, b):
    return float((a - b).pow(2).sum(1).sqrt().mean())


def main():
    G = os.path.join(ROOT, 'tests', 'golden')
    cases = [('net_384x1280_hs1_robust', 1.0, (2, 12), 384, 1280), ('net_256x256_hs1_robust', 1.0, (1,), 256, 256), ('net_384x1280_robust', 0.1, (2,), 384, 1280)]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 565, 96, 320, generator=g).cuda()
    w = (torch.randn(128, 565, 3, 3, generator=g) * (2.0 / (565 * 9)) ** 0.5).cuda()
    b = torch.zeros(128).cuda()
    want = torch.cat([F.conv2d(x[i:i + 1].double(), w.double(), padding=1) for i in range(2)])
    print('%-9s | %-46s | %-34s | %s' % ('fp32_conv', 'whole-net EPE vs REFERENCE (px): 384x1280 hs1 / 256x256 hs1 / 384x1280 hs0.1', '565->128 @ [2,565,96,320] vs fp64: max / rms', 'config-2 shape B=4 graphed'))
    for mode in ['miopen'] + sorted(FP32_CONV_NPROD):
        row = []
        for name, hs, cids, H, W in cases:
            ims = [synthetic.make_smooth_images(c, 1, H, W) for c in cids]
            im1, im2 = torch.cat([a for a, _ in ims]).cuda(), torch.cat([b_ for _, b_ in ims]).cuda()
            gold = torch.from_numpy(np.load(os.path.join(G, name + '.npz'))['flow_f_out']).cuda()
            with torch.no_grad():
                out = build(mode, hs)({'im1': im1, 'im2': im2, 'if_loss': False})['flow_f_out']
            row.append(epe(out, gold))
        with torch.no_grad(), fp32_conv_mode(mode):
            if mode == 'miopen':
                y = F.conv2d(x[:2], w, b, padding=1)
            else:
                y = torch.empty(2, 128, 96, 320, device='cuda')
                ops.conv3x3_forward_raw(x[:2], ops.conv3x3_pack(w), b, y, 1, 0.0)
        e = (y.double() - want)
        lay = (float(e.abs().max()) / float(want.abs().max()), float(e.pow(2).mean().sqrt()) / float(want.pow(2).mean().sqrt()))
        signed = float((e * want.sign()).mean() / want.abs().mean())
        net = build(mode, 0.1)
        im1, im2 = synthetic.make_images(2, 4, 384, 1280)
        r = GraphedInference(net, 4, 384, 1280, device=torch.device('cuda', 0))
        r.load(im1.cuda(), im2.cuda())
        for _ in range(5):
            r.replay()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            r.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 10 * 1e3
        print('%-9s | %.3g / %.3g / %.3g %s | %.2e / %.2e (signed mean %.1e) | %.2f ms = %.0f pairs/s' % (mode, row[0], row[1], row[2], ' ' * 12, lay[0], lay[1], signed, ms, 4e3 / ms), flush=True)
        del r, net
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
