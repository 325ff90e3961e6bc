#!/usr/bin/env python3
"""Where does the 16-bit inference path's distance to fp32 come from?  ("localise the 0.196 px".)

The fp32 forward of the package (generic stacked schedule) is run with 16-bit ROUNDING injected at one tensor class at a time
— x -> x.to(dtype).float() on exactly the tensors the 16-bit path stores in 16 bits — and the end-point error against the clean
fp32 forward (and against the reference's golden output) is tabulated.  Classes:

  weights:<group>      the convolution weights of one module group
  act:<group>          the outputs of the group's hidden convolutions (LeakyReLU layers)
  head:res / fine / sgu_x_out    the 2- / 3-channel linear heads (estimator conv_last, context conv6, SGU conv_last)
  op:images            the frames
  op:warp              warped features (both the decoder's and the SGU's warps)
  op:normalize         the normalised features the cost volume multiplies
  op:corr              the 81 cost-volume channels
  in:flow_est          the flow channels of the estimator input (flow_up stored in the 16-bit buffer)
  in:flow_ctx          the refined-flow channels of the context network's input
  ALL                  everything above at once (an emulation of the 16-bit path);  ALL-heads: all but the heads;  ...

Also printed: the real 16-bit all-HIP path (what bench.py times), against the same fp32 forward and the golden.

  python tools/precision_localise.py [--size 384x1280] [--dtype bf16] [--head-scale 1.0] > profiles/r04_precision_localise.txt
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upflow_pytorch_amd import ops, synthetic  # noqa: E402
from upflow_pytorch_amd.model import upflow as mu  # noqa: E402
from upflow_pytorch_amd.model import pwc_modules as pm  # noqa: E402

FLAGS = {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
         'norm_moments_across_images': False, 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}
GROUPS = {'pyramid': 'feature_pyramid_extractor.', 'conv_1x1': 'conv_1x1.', 'estimator': 'flow_estimators.', 'context': 'context_networks.',
          'sgu_est': 'sgi_model.dense_estimator_mask.', 'sgu_stem': 'sgi_model.upsample_output_conv.'}
HEADS = {'res': 'flow_estimators.conv_last', 'fine': 'context_networks.convs.6', 'sgu_x_out': 'sgi_model.dense_estimator_mask.conv_last'}


def epe(a, b):
    return float((a.float() - b.float()).pow(2).sum(1).sqrt().mean())


def build(sd, dtype=torch.float32):
    conf = mu.UPFlow_net.config()
    conf.update(FLAGS, verbose=False)
    net = conf()
    net.load_state_dict(sd)
    return net.cuda().to(dtype).eval()


class Inject(object):
    """Context manager: installs the rounding hooks named in `classes` on `net` (fp32)."""

    def __init__(self, net, classes, dt):
        self.net, self.classes, self.dt = net, set(classes), dt
        self.handles, self.saved, self.wsaved = [], {}, {}

    def rnd(self, t):
        return t.to(self.dt).float()

    def __enter__(self):
        net, C, rnd = self.net, self.classes, self.rnd
        mods = dict(net.named_modules())
        head_names = set(HEADS.values())
        for g, prefix in GROUPS.items():
            if 'weights:' + g in C:
                for n, p in net.named_parameters():
                    if n.startswith(prefix) and n.endswith('weight'):
                        self.wsaved[n] = p.data.clone()
                        p.data.copy_(rnd(p.data))
            if 'act:' + g in C:
                for n, m in mods.items():
                    if n.startswith(prefix) and isinstance(m, nn.Sequential) and len(m) and isinstance(m[0], nn.Conv2d) and n not in head_names:
                        self.handles.append(m.register_forward_hook(lambda mod, i, o: rnd(o)))
        for h, name in HEADS.items():
            if 'head:' + h in C:
                self.handles.append(mods[name].register_forward_hook(lambda mod, i, o: rnd(o)))
        if 'in:flow_est' in C:
            def pre(mod, args):
                x = args[0].clone()
                x[:, -2:] = rnd(x[:, -2:])
                return (x,)
            self.handles.append(net.flow_estimators.register_forward_pre_hook(pre))
        if 'in:flow_ctx' in C:
            def pre2(mod, args):
                x = args[0].clone()
                x[:, -2:] = rnd(x[:, -2:])
                return (x,)
            self.handles.append(net.context_networks.register_forward_pre_hook(pre2))
        for name in ('warp', 'normalize', 'corr81_forward_raw'):
            self.saved[name] = getattr(ops, name)
        if 'op:warp' in C:
            f = self.saved['warp']
            ops.warp = lambda *a, **k: rnd(f(*a, **k))
        if 'op:normalize' in C:
            f2 = self.saved['normalize']
            ops.normalize = lambda x: rnd(f2(x))
        if 'op:corr' in C:
            f3 = self.saved['corr81_forward_raw']

            def corr(f1, f2_, out=None, leaky_slope=0.0):
                r = f3(f1, f2_, out=out, leaky_slope=leaky_slope)
                r.copy_(rnd(r))
                return r
            ops.corr81_forward_raw = corr
        return self

    def __exit__(self, *a):
        for h in self.handles:
            h.remove()
        for k, v in self.saved.items():
            setattr(ops, k, v)
        for n, p in self.net.named_parameters():
            if n in self.wsaved:
                p.data.copy_(self.wsaved[n])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', default='384x1280')
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--head-scale', type=float, default=1.0)
    ap.add_argument('--cid', type=int, default=2)
    a = ap.parse_args()
    H, W = [int(v) for v in a.size.split('x')]
    dt = {'bf16': torch.bfloat16, 'fp16': torch.float16}[a.dtype]
    sd = synthetic.make_state_dict(0, head_scale=a.head_scale)
    im1, im2 = synthetic.make_smooth_images(a.cid, 1, H, W)
    im1, im2 = im1.cuda(), im2.cuda()
    gold = None
    gp = os.path.join(ROOT, 'tests', 'golden', 'net_%dx%d_hs1_robust.npz' % (H, W))
    if a.head_scale == 1.0 and os.path.exists(gp) and a.cid in (1, 2):
        gold = torch.from_numpy(np.load(gp)['flow_f_out'][:1]).cuda()
    net = build(sd)

    def run(classes, images_rounded=False):
        with torch.no_grad(), Inject(net, classes, dt):
            x1, x2 = (im1.to(dt).float(), im2.to(dt).float()) if 'op:images' in classes else (im1, im2)
            return net({'im1': x1, 'im2': x2, 'if_loss': False})['flow_f_out']

    clean = run([])
    mag = float(clean.pow(2).sum(1).sqrt().mean())
    print('# %dx%d %s head_scale %.2f: mean |flow| %.3f px; fp32 HIP path vs reference golden: %s px'
          % (H, W, a.dtype, a.head_scale, mag, 'n/a' if gold is None else '%.3g' % epe(clean, gold)))
    singles = (['weights:' + g for g in GROUPS] + ['act:' + g for g in GROUPS] + ['head:' + h for h in HEADS]
               + ['op:images', 'op:warp', 'op:normalize', 'op:corr', 'in:flow_est', 'in:flow_ctx'])
    rows = []
    print('%-28s %12s %12s' % ('rounded tensor class', 'EPE px', '% of |flow|'))
    for c in singles:
        e = epe(run([c]), clean)
        rows.append((c, e))
        print('%-28s %12.5f %12.4f' % (c, e, 100 * e / mag), flush=True)
    allc = list(singles)
    combos = {'ALL': allc,
              'ALL - heads': [c for c in allc if not c.startswith('head:')],
              'ALL - heads - in:flow': [c for c in allc if not c.startswith('head:') and not c.startswith('in:flow')],
              'ALL - weights': [c for c in allc if not c.startswith('weights:')],
              'ALL - act': [c for c in allc if not c.startswith('act:')],
              'ALL - op:corr - op:normalize': [c for c in allc if c not in ('op:corr', 'op:normalize')],
              'heads only': [c for c in allc if c.startswith('head:')],
              'weights only': [c for c in allc if c.startswith('weights:')],
              'act only': [c for c in allc if c.startswith('act:')]}
    for name, cl in combos.items():
        e = epe(run(cl), clean)
        rows.append((name, e))
        print('%-28s %12.5f %12.4f' % (name, e, 100 * e / mag), flush=True)
    rss = float(np.sqrt(sum(e * e for c, e in rows[:len(singles)])))
    print('%-28s %12.5f %12.4f   (root-sum-square of the single classes)' % ('RSS singles', rss, 100 * rss / mag))
    # the real 16-bit path
    net16 = build(sd, dt)
    with torch.no_grad():
        real = net16({'im1': im1, 'im2': im2, 'if_loss': False})['flow_f_out']
    print('%-28s %12.5f %12.4f' % ('REAL %s all-HIP path' % a.dtype, epe(real, clean), 100 * epe(real, clean) / mag))
    if gold is not None:
        print('%-28s %12.5f %12.4f   (vs the reference golden)' % ('REAL %s all-HIP path' % a.dtype, epe(real, gold), 100 * epe(real, gold) / mag))
    print(json.dumps({'size': [H, W], 'dtype': a.dtype, 'mean_flow_px': mag, 'rows': rows}))


if __name__ == '__main__':
    main()
