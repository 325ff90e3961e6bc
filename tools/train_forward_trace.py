#!/usr/bin/env python3
"""The config-3-style training FORWARD at realistic motion (tests/golden/train_128x416_hs1) in fp32 and in the bf16 matrix-core mode, operator by
operator and level by level: where do the two forwards part?  (round 6: the 0.83 px between them is a validity-mask flip of the feature warp
at the 4x13 level — one pixel of 52 — not accumulated rounding; tools/precision_localise_train.py has the per-class picture.)
    python tools/train_forward_trace.py > profiles/r06_train_forward_trace.txt"""
import os, sys
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import numpy as np, torch
from upflow_pytorch_amd import synthetic, ops
from upflow_pytorch_amd.model.upflow import UPFlow_net
from upflow_pytorch_amd.model import upflow as mu
FLAGS = {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False, 'norm_moments_across_images': False, 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}

def run(mode):
    conf = UPFlow_net.config(); d = dict(FLAGS); d.update(synthetic.TRAIN_FLAGS); d['train_conv_dtype'] = mode; conf.update(d, verbose=False)
    net = conf(); net.load_state_dict(synthetic.make_state_dict(0, head_scale=1.0)); net = net.cuda().train()
    rec = []
    def wrap(obj, name, label, pick=lambda r: r):
        o = getattr(obj, name)
        def f(*a, **k):
            r = o(*a, **k)
            v = pick(r)
            rec.append((label, v.detach().float().clone() if torch.is_tensor(v) else None))
            return r
        setattr(obj, name, f)
    wrap(net.feature_pyramid_extractor, 'forward', 'pyramid', lambda r: torch.cat([p.flatten() for p in r]))
    wrap(net.sgi_model, 'forward', 'sgu.flow_up', lambda r: r[1])
    wrap(net.warping_layer, 'forward', 'warp')
    wrap(net, '_corr_leaky', 'corr')
    wrap(net, '_level_update', 'level.flow')
    o_norm = mu.network_tools.normalize_features
    def nf(cls, fl, *a, **k):
        r = o_norm(fl, *a, **k); rec.append(('normalize', torch.cat([t.detach().float().flatten() for t in r]))); return r
    mu.network_tools.normalize_features = classmethod(nf)
    o_up = mu.upsample2d_flow_as
    def up(*a, **k):
        r = o_up(*a, **k); rec.append(('upsample', r.detach().float().clone())); return r
    mu.upsample2d_flow_as = up
    o_blend = ops.sgu_blend
    def bl(flow_init, x_out, olf=None, **k):
        rec.append(('sgu.x_out', x_out.detach().float().clone())); r = o_blend(flow_init, x_out, olf, **k); return r
    ops.sgu_blend = bl
    try:
        batch = {k: v.cuda() for k, v in synthetic.make_train_batch(**synthetic.TRAIN_HS1).items()}; batch['if_loss'] = True
        out = net(batch)
    finally:
        mu.network_tools.normalize_features = o_norm; mu.upsample2d_flow_as = o_up; ops.sgu_blend = o_blend
    rec.append(('flow_f_out', out['flow_f_out'].detach().float()))
    return rec
a = run('fp32'); b = run('bf16')
print(len(a), len(b))
for (la, ta), (lb, tb) in zip(a, b):
    if ta is None or tb is None or ta.shape != tb.shape:
        print(la, lb, 'shape', None if ta is None else tuple(ta.shape), None if tb is None else tuple(tb.shape)); continue
    d = (ta - tb).abs()
    if la == 'warp':        # pixels whose validity (all channels zero = masked out) differs between the two runs
        za, zb = (ta.abs().sum(1) == 0), (tb.abs().sum(1) == 0)
        print('   warp mask: %d of %d pixels masked in fp32, %d in bf16, %d differ' % (int(za.sum()), za.numel(), int(zb.sum()), int((za != zb).sum())))
    print('%-14s %-14s shape %-22s  |ref| mean %.4f  diff mean %.5f max %.4f  rel %.5f' % (la, lb, tuple(ta.shape), float(ta.abs().mean()), float(d.mean()), float(d.max()), float(d.mean() / ta.abs().mean().clamp_min(1e-9))))
