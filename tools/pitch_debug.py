#!/usr/bin/env python3
"""Which intermediate differs first between the pitched + octet schedule and the contiguous ragged one (round 5)?
python tools/pitch_debug.py [B H W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import _weights
from upflow_pytorch_amd import ops
from upflow_pytorch_amd.model.upflow import UPFlow_net

B, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 375, 1242)
conf = UPFlow_net.config()
conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False, 'norm_moments_across_images': False,
             'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}, verbose=False)
net = conf()
net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
net = net.cuda().bfloat16().eval()
im1, im2 = _weights.make_smooth_images(9, B, H, W)
im1, im2 = im1.cuda(), im2.cuda()


def run(**flags):
    for k in ('_no_pitch', '_no_c8', '_no_c8_est', '_no_c8_sgu', '_no_c8_ctx'):
        setattr(net, k, False)
    for k, v in flags.items():
        setattr(net, k, v)
    net._taps = []
    with torch.no_grad():
        out = net({'im1': im1, 'im2': im2, 'if_loss': False})
    taps = [(n, (ops.from_c8(t) if t.dim() == 5 and t.shape[-1] == 8 and 'pair' not in n else t).float().clone()) for n, t in net._taps]
    return out, taps


variants = {'pitched(no_c8_est)': dict(_no_c8_est=True), 'pitched,no_c8': dict(_no_c8=True), 'ragged,c8 n/a': dict(_no_pitch=True),
            'ragged(no_pitch,no_c8)': dict(_no_pitch=True, _no_c8=True)}
res = {k: run(**v) for k, v in variants.items()}
base = res['ragged(no_pitch,no_c8)'][1]
for name in list(variants)[:-1]:
    taps = res[name][1]
    print('==', name, 'vs ragged')
    bd = dict(base)
    shown = 0
    for n, t in taps:
        if n not in bd or bd[n].shape != t.shape:
            continue
        d = (t - bd[n]).abs()
        nz = int((d > 0).sum())
        if nz:
            idx = (d > 0).nonzero()[0].tolist()
            print('  %-22s differs: %d of %d elements, max %.3g, first at %s' % (n, nz, d.numel(), float(d.max()), idx))
            shown += 1
            if shown >= 6:
                break
    if not shown:
        print('  every common tap identical')
