#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); the GPU box never sees the reference.
The fixtures are data only: seeded inputs and the reference's outputs.

Harness shims — none of them changes reference code:
  1. empty stub modules for cv2 / png / imageio / correlation_cuda (never touched on this path);
  2. torch.utils.data.dataloader._DataLoaderIter alias (utils/tools.py:2 wants torch 1.1);
  3. grid_sample(align_corners=None) -> align_corners=True, the torch-1.1 behaviour the model was
     written for (requirements.txt:12; pwc_modules.py:197-200 normalises with W-1 / H-1);
  4. train mode only: model.upflow.upsample2d_flow_as replaced by its out-of-place equivalent
     (pwc_modules.py:86-88 mutates chunk views in place, which torch 2.x autograd rejects).

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import io
import contextlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _weights  # noqa: E402

REF = '/root/reference'


def import_reference():
    for m in ['cv2', 'png', 'imageio', 'correlation_cuda']:
        sys.modules.setdefault(m, types.ModuleType(m))
    import torch.utils.data.dataloader as dl
    if not hasattr(dl, '_DataLoaderIter'):
        dl._DataLoaderIter = dl._BaseDataLoaderIter
    import torch.nn.functional as F
    if not getattr(F.grid_sample, '_upf_shim', False):
        _gs = F.grid_sample

        def grid_sample(input, grid, mode='bilinear', padding_mode='zeros', align_corners=None):
            return _gs(input, grid, mode=mode, padding_mode=padding_mode,
                       align_corners=True if align_corners is None else align_corners)
        grid_sample._upf_shim = True
        F.grid_sample = grid_sample
    sys.path.insert(0, REF)
    import model.upflow as upflow
    import model.pwc_modules as pwc
    from utils.tools import tools
    from utils.pytorch_correlation import Corr_pyTorch
    return upflow, pwc, tools, Corr_pyTorch


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def gen(seed):
    return torch.Generator().manual_seed(seed)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024))


TEST_FLAGS = {  # test.py:22-30
    'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
    'norm_moments_across_images': False, 'if_froze_pwc': False,
    'if_use_cor_pytorch': True, 'if_sgu_upsample': True,
}


def build_net(upflow, extra=None, seed=0, head_scale=1.0):
    conf = upflow.UPFlow_net.config()
    d = dict(TEST_FLAGS)
    d.update(extra or {})
    quiet(conf.update, d)
    net = conf()
    sd = _weights.make_state_dict(seed, head_scale=head_scale)
    net.load_state_dict(sd)
    net.eval()
    return net


# ------------------------------------------------------------------------------------------------
def golden_corr(Corr_pyTorch):
    corr = Corr_pyTorch(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1)
    for i, (B, C, H, W) in enumerate([(2, 32, 24, 40), (1, 7, 5, 9), (1, 196, 6, 20), (1, 3, 3, 3)]):
        g = gen(2100 + i)
        f1 = torch.randn(B, C, H, W, generator=g, requires_grad=True)
        f2 = torch.randn(B, C, H, W, generator=g, requires_grad=True)
        out = corr(f1, f2)
        go = torch.randn(out.shape, generator=g)
        g1, g2 = torch.autograd.grad(out, (f1, f2), go)
        save('corr_%d' % i, f1=f1, f2=f2, out=out, grad_out=go, g1=g1, g2=g2)


def flow_cases(B, H, W, base_seed):
    g = gen(base_seed)
    return {
        'zero': torch.zeros(B, 2, H, W),
        'int2': torch.full((B, 2, H, W), 2.0),
        'n3': torch.randn(B, 2, H, W, generator=g) * 3.0,
        'n5': torch.randn(B, 2, H, W, generator=g) * 5.0,
        'edge': torch.cat([torch.linspace(-6, W + 5, W).view(1, 1, 1, W).expand(B, 1, H, W)
                           - torch.arange(W).float().view(1, 1, 1, W),
                           torch.linspace(-3, H + 2, H).view(1, 1, H, 1).expand(B, 1, H, W)
                           - torch.arange(H).float().view(1, 1, H, 1)], 1).contiguous(),
    }


def golden_warp(pwc, tools):
    layer = pwc.WarpingLayer_no_div()
    for si, (B, C, H, W) in enumerate([(1, 3, 48, 160), (1, 2, 96, 320), (1, 5, 6, 20), (1, 2, 1, 2)]):
        g = gen(2200 + si)
        x = torch.rand(B, C, H, W, generator=g) + 1.0
        for name, flow in flow_cases(B, H, W, 2250 + si).items():
            if si == 1 and name not in ('zero', 'n5'):
                continue  # keep the 96x320 fixtures small
            xr = x.clone().requires_grad_(True)
            fr = flow.clone().requires_grad_(True)
            y = layer(xr, fr)
            # the validity mask itself (pwc_modules.py:201-206), recomputed through the reference's
            # own formula: warp a ones tensor and test >= 1.0
            ones = torch.ones(B, 1, H, W)
            mask = (layer(ones, flow) > 0).to(torch.uint8)
            go = torch.randn(y.shape, generator=g)
            gx, gf = torch.autograd.grad(y, (xr, fr), go)
            # unmasked variant: tools.torch_warp (utils/tools.py:1274-1319)
            xr2 = x.clone().requires_grad_(True)
            fr2 = flow.clone().requires_grad_(True)
            y2 = tools.torch_warp(xr2, fr2)
            gx2, gf2 = torch.autograd.grad(y2, (xr2, fr2), go)
            save('warp_%d_%s' % (si, name), x=x, flow=flow, y=y, mask=np.packbits(mask.numpy()),
                 grad_out=go, gx=gx, gflow=gf, y_nomask=y2, gx_nomask=gx2, gflow_nomask=gf2)


def oop_upsample2d_flow_as(inputs, target_as, mode="bilinear", if_rate=False):
    """Out-of-place equivalent of pwc_modules.py:77-90 (shim 4)."""
    import torch.nn.functional as F
    _, _, h, w = target_as.size()
    res = F.interpolate(inputs, [h, w], mode=mode, align_corners=True)
    if if_rate:
        _, _, h_, w_ = inputs.size()
        u, v = res.chunk(2, dim=1)
        res = torch.cat([u * (w / w_), v * (h / h_)], dim=1)
    return res


def golden_upsample(pwc):
    for i, ((h_, w_), (h, w)) in enumerate([((6, 20), (12, 40)), ((24, 80), (96, 320)), ((4, 13), (8, 26)),
                                            ((7, 16), (14, 32)), ((1, 2), (2, 4)), ((5, 7), (5, 7))]):
        g = gen(2300 + i)
        x = torch.randn(2, 2, h_, w_, generator=g) * 4
        tgt = torch.zeros(2, 1, h, w)
        y = pwc.upsample2d_flow_as(x.clone(), tgt, mode='bilinear', if_rate=True)
        y_norate = pwc.upsample2d_flow_as(x.clone(), tgt, mode='bilinear', if_rate=False)
        y_uf = pwc.upsample_flow(x.clone(), target_size=(h, w))
        xr = x.clone().requires_grad_(True)
        go = torch.randn(y.shape, generator=g)
        gx, = torch.autograd.grad(oop_upsample2d_flow_as(xr, tgt, if_rate=True), xr, go)
        save('upsample_%d' % i, x=x, y=y, y_norate=y_norate, y_upsample_flow=y_uf, grad_out=go, gx=gx,
             size=np.array([h, w]))


def golden_normalize(upflow):
    nt = upflow.network_tools
    for i, (B, C, H, W) in enumerate([(2, 32, 12, 20), (1, 196, 6, 20), (2, 5, 3, 7)]):
        g = gen(2400 + i)
        a = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
        b = torch.randn(B, C, H, W, generator=g) * 0.3 - 1
        ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        na, nb = nt.normalize_features((ar, br), normalize=True, center=True,
                                       moments_across_channels=False, moments_across_images=False)
        goa = torch.randn(na.shape, generator=g)
        gob = torch.randn(nb.shape, generator=g)
        ga, gb = torch.autograd.grad([na, nb], [ar, br], [goa, gob])
        save('normalize_%d' % i, a=a, b=b, na=na, nb=nb, goa=goa, gob=gob, ga=ga, gb=gb)


def golden_sgu_blend(upflow, tools):
    """The interpolation-blend of model/upflow.py:79-88, driven through the reference's own
    sgu_model.forward with its dense estimator replaced by a recorded x_out."""
    for i, (B, H, W, Hf, Wf) in enumerate([(2, 24, 80, None, None), (1, 6, 20, None, None),
                                           (2, 12, 40, 48, 160), (1, 16, 32, 64, 128)]):
        g = gen(2500 + i)
        sgu = upflow.network_tools.sgu_model()
        sgu.eval()
        x_out = torch.randn(B, 3, H, W, generator=g) * torch.tensor([2.0, 2.0, 1.5]).view(1, 3, 1, 1)
        x_out = x_out.requires_grad_(True)

        class Fixed(torch.nn.Module):
            def forward(self, x):
                return x, x_out
        sgu.dense_estimator_mask = Fixed()
        flow_init = (torch.randn(B, 2, H, W, generator=g) * 3).requires_grad_(True)
        f1 = torch.randn(B, 32, H, W, generator=g)
        f2 = torch.randn(B, 32, H, W, generator=g)
        olf = None
        if Hf is not None:
            # a smooth field (like a real up-sampled flow): white noise would turn the fp32 ulp of the
            # sampling position (1.5e-5 at x~160) into 1e-4 output differences and pin nothing
            import torch.nn.functional as F_
            olf = F_.interpolate(torch.randn(B, 2, Hf // 8, Wf // 8, generator=g) * 5, size=(Hf, Wf),
                                 mode='bicubic', align_corners=True).contiguous().requires_grad_(True)
        import model.upflow as mu
        old = mu.upsample2d_flow_as
        mu.upsample2d_flow_as = oop_upsample2d_flow_as
        try:
            fi, flow_up, inter_flow, inter_mask = sgu(flow_init, f1, f2, output_level_flow=olf)
        finally:
            mu.upsample2d_flow_as = old
        go = torch.randn(flow_up.shape, generator=g)
        ins = [x_out, flow_init] + ([olf] if olf is not None else [])
        grads = torch.autograd.grad(flow_up, ins, go, allow_unused=True)
        d = dict(x_out=x_out, flow_init=flow_init, flow_up=flow_up, inter_flow=inter_flow, inter_mask=inter_mask,
                 grad_out=go, g_x_out=grads[0])
        if olf is not None:
            d.update(output_level_flow=olf, g_output_level_flow=grads[2])
        else:
            d.update(g_flow_init=grads[1])
        save('sgu_blend_%d' % i, **d)


def golden_occ(tools):
    occ = tools.occ_check_model(occ_type='for_back_check', occ_alpha_1=0.1, occ_alpha_2=0.5, obj_out_all='obj')
    for i, (B, H, W) in enumerate([(2, 24, 80), (1, 64, 128)]):
        g = gen(2600 + i)
        ff = torch.randn(B, 2, H, W, generator=g) * 2
        fb = -ff + torch.randn(B, 2, H, W, generator=g) * 0.5
        o1, o2 = occ(flow_f=ff, flow_b=fb)
        save('occ_%d' % i, flow_f=ff, flow_b=fb, occ_fw=o1, occ_bw=o2)


# ------------------------------------------------------------------------------------------------
class Tracer:
    """Records inputs/outputs of every hot-op call of one reference forward, in call order."""

    def __init__(self, upflow, pwc, tools, net):
        self.events = []
        self.arrays = {}
        self.upflow, self.pwc, self.tools, self.net = upflow, pwc, tools, net
        self._x_out = None

    def rec(self, op, **tensors):
        idx = len(self.events)
        ev = {'op': op, 'idx': idx, 'keys': sorted(tensors)}
        for k, v in tensors.items():
            self.arrays['e%03d_%s' % (idx, k)] = v.detach().clone().numpy()
        self.events.append(ev)

    def __enter__(self):
        import model.upflow as mu
        self.mu = mu
        T = self
        self.handles = []
        self.handles.append(self.net.correlation_pytorch.register_forward_hook(
            lambda m, i, o: T.rec('corr', f1=i[0], f2=i[1], out=o)))

        def warp_hook(m, i, o):
            T.rec('warp_mask', x=i[0], flow=i[1], y=o)
        self.handles.append(self.net.warping_layer.register_forward_hook(warp_hook))
        self.handles.append(self.net.sgi_model.warping_layer.register_forward_hook(warp_hook))

        def est_hook(m, i, o):
            T._x_out = o[1]
        self.handles.append(self.net.sgi_model.dense_estimator_mask.register_forward_hook(est_hook))

        def sgu_hook(m, args, kwargs, o):
            olf = kwargs.get('output_level_flow', args[3] if len(args) > 3 else None)
            d = dict(flow_init=o[0], x_out=T._x_out, flow_up=o[1], inter_flow=o[2], inter_mask=o[3])
            if olf is not None:
                d['output_level_flow'] = olf
            T.rec('sgu_blend', **d)
        self.handles.append(self.net.sgi_model.register_forward_hook(sgu_hook, with_kwargs=True))

        self.old_tw = self.tools.torch_warp

        def torch_warp(x, flo):
            y = T.old_tw(x, flo)
            T.rec('warp', x=x, flow=flo, y=y)
            return y
        self.tools.torch_warp = torch_warp

        self.old_up = mu.upsample2d_flow_as

        def up(inputs, target_as, mode="bilinear", if_rate=False):
            x_in = inputs.detach().clone()
            y = T.old_up(inputs, target_as, mode=mode, if_rate=if_rate)
            T.rec('upsample_rate' if if_rate else 'upsample', x=x_in, y=y)
            return y
        mu.upsample2d_flow_as = up

        self.old_norm = self.upflow.network_tools.normalize_features

        def norm(feature_list, normalize, center, moments_across_channels=True, moments_across_images=True):
            out = T.old_norm(feature_list, normalize, center, moments_across_channels=moments_across_channels,
                             moments_across_images=moments_across_images)
            T.rec('normalize', a=feature_list[0], b=feature_list[1], na=out[0], nb=out[1])
            return out
        self.upflow.network_tools.normalize_features = norm
        return self

    def __exit__(self, *a):
        for h in self.handles:
            h.remove()
        self.tools.torch_warp = self.old_tw
        self.mu.upsample2d_flow_as = self.old_up
        self.upflow.network_tools.normalize_features = self.old_norm


def robust_mask_patch(pwc):
    """H2/P3b: the same warp with the exact in-bounds predicate instead of `mask >= 1.0`.
    Oracle-side monkeypatch of WarpingLayer_no_div.forward; reference code untouched on disk."""
    import torch.nn.functional as F
    old = pwc.WarpingLayer_no_div.forward

    def forward(self, x, flow):
        B, C, H, W = x.size()
        xx = torch.arange(0, W).view(1, 1, 1, W).expand(B, 1, H, W).float()
        yy = torch.arange(0, H).view(1, 1, H, 1).expand(B, 1, H, W).float()
        px = xx + flow[:, 0:1]
        py = yy + flow[:, 1:2]
        gx = 2.0 * px / max(W - 1, 1) - 1.0
        gy = 2.0 * py / max(H - 1, 1) - 1.0
        vgrid = torch.cat([gx, gy], 1).permute(0, 2, 3, 1)
        x_warp = F.grid_sample(x, vgrid, padding_mode='zeros')
        mask = ((px >= 0) & (px <= W - 1) & (py >= 0) & (py <= H - 1)).float()
        return x_warp * mask
    pwc.WarpingLayer_no_div.forward = forward
    return old


def golden_net(upflow, pwc, tools):
    meta = {'weights_sha256': _weights.state_dict_sha256(_weights.make_state_dict(0)),
            'weights_sha256_hs': _weights.state_dict_sha256(_weights.make_state_dict(0, head_scale=0.1))}
    cases = [('net_64x128', 1, 64, 128, 1), ('net_256x256', 1, 256, 256, 1)]
    for name, B, H, W, cid in cases:
        for variant in ['literal', 'robust']:
            old = robust_mask_patch(pwc) if variant == 'robust' else None
            try:
                net = build_net(upflow, head_scale=0.1)
                im1, im2 = _weights.make_smooth_images(cid, B, H, W)
                with torch.no_grad():
                    if name == 'net_64x128' and variant == 'literal':
                        with Tracer(upflow, pwc, tools, net) as tr:
                            out = net({'im1': im1, 'im2': im2, 'if_loss': False})
                        save('trace_64x128', **tr.arrays)
                        with open(os.path.join(HERE, 'trace_64x128.json'), 'w') as f:
                            json.dump(tr.events, f, indent=0)
                    else:
                        out = net({'im1': im1, 'im2': im2, 'if_loss': False})
                    # self-sensitivity of the reference (H2/P3a): same forward, N(0,1e-7²) on im1
                    g = gen(77)
                    out_n = net({'im1': im1 + 1e-7 * torch.randn(im1.shape, generator=g), 'im2': im2, 'if_loss': False})
                sens = float((out['flow_f_out'] - out_n['flow_f_out']).pow(2).sum(1).sqrt().mean())
                meta['%s_%s_self_sensitivity_epe' % (name, variant)] = sens
                save('%s_%s' % (name, variant), flow_f_out=out['flow_f_out'], flow_b_out=out['flow_b_out'],
                     occ_fw=out['occ_fw'].to(torch.uint8), occ_bw=out['occ_bw'].to(torch.uint8))
            finally:
                if old is not None:
                    pwc.WarpingLayer_no_div.forward = old
    with open(os.path.join(HERE, 'net_meta.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    print(json.dumps(meta, indent=1))


def golden_net_headline(upflow, pwc, tools):
    """BASELINE config 2's resolution (384x1280, B=1, fp32, robust mask on both sides so that the comparison is
    well-posed): the reference's forward flow + occlusion mask, and its self-sensitivity."""
    old = robust_mask_patch(pwc)
    try:
        net = build_net(upflow, head_scale=0.1)
        im1, im2 = _weights.make_smooth_images(2, 1, 384, 1280)
        with torch.no_grad():
            out = net({'im1': im1, 'im2': im2, 'if_loss': False})
            g = gen(77)
            out_n = net({'im1': im1 + 1e-7 * torch.randn(im1.shape, generator=g), 'im2': im2, 'if_loss': False})
        sens = float((out['flow_f_out'] - out_n['flow_f_out']).pow(2).sum(1).sqrt().mean())
        save('net_384x1280_robust', flow_f_out=out['flow_f_out'], occ_fw=np.packbits(out['occ_fw'].to(torch.uint8).numpy()),
             flow_b_checksum=np.array([float(out['flow_b_out'].double().sum()), float(out['flow_b_out'].double().abs().sum())]),
             self_sensitivity_epe=np.array([sens]))
        print('384x1280 robust: self-sensitivity %.3g, mean |flow| %.3g' % (sens, float(out['flow_f_out'].abs().mean())))
    finally:
        pwc.WarpingLayer_no_div.forward = old


def golden_net_realistic(upflow, pwc, tools):
    """whole-net vectors at REALISTIC motion.  The other whole-net fixtures use head_scale=0.1 weights
    (mean |flow| 0.35-0.92 px); KITTI flows are tens of pixels (README.md:10, test.py:22-47).  With full-scale heads
    (head_scale=1.0) the same synthetic network produces mean |flow| ~10 px / p99 ~21 px at 256x256 and more at 384x1280:
    the +-4 search range, the border masks and the SGU warp are exercised at whole-net level outside the sub-pixel regime.
    Robust mask on both sides (well-posed comparison); two distinct pairs at 384x1280 so that the
    batched / graphed bench path can be checked on more than one item.  The reference's self-sensitivity (1e-7 input noise)
    and the flow statistics go to net_meta.json."""
    meta_path = os.path.join(HERE, 'net_meta.json')
    meta = json.load(open(meta_path))
    meta['weights_sha256_hs1'] = _weights.state_dict_sha256(_weights.make_state_dict(0, head_scale=1.0))
    old = robust_mask_patch(pwc)
    try:
        net = build_net(upflow, head_scale=1.0)
        for name, cids, H, W in [('net_256x256_hs1_robust', (1,), 256, 256), ('net_384x1280_hs1_robust', (2, 12), 384, 1280)]:
            ims = [_weights.make_smooth_images(c, 1, H, W) for c in cids]
            im1 = torch.cat([a for a, _ in ims])
            im2 = torch.cat([b for _, b in ims])
            with torch.no_grad():
                out = net({'im1': im1, 'im2': im2, 'if_loss': False})
                g = gen(77)
                out_n = net({'im1': im1 + 1e-7 * torch.randn(im1.shape, generator=g), 'im2': im2, 'if_loss': False})
            f = out['flow_f_out']
            mag = f.pow(2).sum(1).sqrt()
            sens = float((f - out_n['flow_f_out']).pow(2).sum(1).sqrt().mean())
            stats = {'self_sensitivity_epe': sens, 'mean_flow_px': float(mag.mean()), 'p99_flow_px': float(mag.flatten().quantile(0.99)),
                     'max_flow_px': float(mag.max())}
            meta[name] = stats
            fb = out['flow_b_out']
            save(name, flow_f_out=f, occ_fw=np.packbits(out['occ_fw'].to(torch.uint8).numpy()),
                 flow_b_out=fb if H * W <= 256 * 256 else fb[:, :, ::4, ::4].contiguous(),   # (full at 256x256, every 4th pixel at 384x1280)
                 flow_b_checksum=np.array([float(fb.double().sum()), float(fb.double().abs().sum())]),
                 self_sensitivity_epe=np.array([sens]), mean_flow_px=np.array([stats['mean_flow_px']]),
                 p99_flow_px=np.array([stats['p99_flow_px']]))
            print(name, stats, flush=True)
    finally:
        pwc.WarpingLayer_no_div.forward = old
    with open(meta_path, 'w') as fo:
        json.dump(meta, fo, indent=1)


def golden_train(upflow, pwc, tools):
    """Train-mode forward + backward of the reference (BASELINE config 3 at a small crop): loss terms and
    the gradient norm of every parameter.  Robust mask on (so the comparison is well-posed, §7-H2) and the
    out-of-place upsample2d_flow_as shim (SURVEY.md §8c shim 4)."""
    import model.upflow as mu
    old_fwd = robust_mask_patch(pwc)
    old_up = mu.upsample2d_flow_as
    mu.upsample2d_flow_as = oop_upsample2d_flow_as
    try:
        net = build_net(upflow, extra=_weights.TRAIN_FLAGS, head_scale=0.1)
        net.train()
        batch = _weights.make_train_batch()
        batch['if_loss'] = True
        out = net(batch)
        terms = {k: out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')}
        loss = sum(terms.values())
        loss.backward()
        names = sorted(n for n, _ in net.named_parameters())
        params = dict(net.named_parameters())
        gnorm = np.array([float(params[n].grad.norm()) for n in names], dtype=np.float64)
        # direction, not only size: 64 seeded random projections of every gradient, and every bias gradient in full
        proj = _weights.grad_projections({n: params[n].grad for n in names})
        bias = {'gbias_%d' % i: params[n].grad for i, n in enumerate(names) if n.endswith('.bias')}
        save('train_128x192', loss=np.array([float(loss)]), **{k: np.array([float(v)]) for k, v in terms.items()},
             grad_norms=gnorm, grad_proj=proj, flow_f_out=out['flow_f_out'], occ_fw=out['occ_fw'].to(torch.uint8), **bias)
        print({k: float(v) for k, v in terms.items()}, 'grad norm sum', gnorm.sum())
        # the reference's OWN directional sensitivity to a 16-bit-sized perturbation: the same step with the four input frames
        # rounded to bf16 (nothing else changes: fp32 network, fp32 arithmetic).  The hard masks of the model (warp validity,
        # occlusion thresholds) make its gradients discontinuous, so this — not 1.0 — is the yardstick for the bf16 / fp16
        # training modes of the build (tests/test_hip_train.py).
        net.zero_grad()
        b16 = {k: (v.bfloat16().float() if k in ('im1', 'im2', 'im1_raw', 'im2_raw') else v) for k, v in batch.items()}
        out16 = net(b16)
        sum(out16[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')).backward()
        proj16 = _weights.grad_projections({n: params[n].grad for n in names})
        cos = (proj * proj16).sum(1) / (np.linalg.norm(proj, axis=1) * np.linalg.norm(proj16, axis=1))
        print('reference vs reference-with-bf16-rounded-frames: gradient cosine min %.5f median %.5f' % (cos.min(), np.median(cos)))
        z = dict(np.load(os.path.join(HERE, 'train_128x192.npz')))
        z['grad_proj_bf16_frames'] = proj16
        np.savez_compressed(os.path.join(HERE, 'train_128x192.npz'), **z)
    finally:
        pwc.WarpingLayer_no_div.forward = old_fwd
        mu.upsample2d_flow_as = old_up


def golden_train_trajectory(upflow, pwc, tools):
    """The reference's OWN optimisation trajectory: 121 steps of the unsupervised recipe (Adam(amsgrad), lr 1e-4, weight decay
    1e-4: scripts/simple_train.py:121-122; loss = the four terms of model/upflow.py:394-491) on one synthetic batch
    (upflow_pytorch_amd.train.synthetic_train_batch: two 128x192 crops, 2-pixel horizontal motion), with the full set of loss
    terms and with the pyramid distillation switched off (its weight defaults to 0 in the reference, model/upflow.py:312).  The
    loss terms every 20 steps -> train_traj_128x192.json: tests/test_hip_train.py runs the SAME steps on the GPU and must stay
    on this trajectory (the recipe is chaotic on this pair — the reference itself leaves it after ~200 steps — so the test
    stops at 100).  ~4 minutes on 8 CPU threads."""
    import model.upflow as mu
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from upflow_pytorch_amd.train import synthetic_train_batch
    old_fwd = robust_mask_patch(pwc)
    old_up = mu.upsample2d_flow_as
    mu.upsample2d_flow_as = oop_upsample2d_flow_as
    res = {}
    try:
        for tag, extra in (('full', {}), ('no_distillation', {'multi_scale_distillation_weight': 0})):
            flags = dict(_weights.TRAIN_FLAGS)
            flags.update(extra)
            net = build_net(upflow, extra=flags, head_scale=0.1)
            net.train()
            batch = synthetic_train_batch(2, crop_hw=(128, 192), raw_hw=(160, 256))
            batch['if_loss'] = True
            opt = torch.optim.Adam(net.parameters(), lr=1e-4, amsgrad=True, weight_decay=1e-4)
            gt = torch.zeros(2, 2, 128, 192)
            gt[:, 0] = 2.0
            rows = []
            for i in range(121):
                opt.zero_grad()
                out = net(batch)
                terms = {k: out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss') if out.get(k) is not None}
                loss = sum(terms.values())
                loss.backward()
                opt.step()
                if i % 20 == 0:
                    f = out['flow_f_out'].detach()[:, :, 16:-16, 16:-16]
                    row = {k: float(v) for k, v in terms.items()}
                    row.update(step=i, loss=float(loss), epe_vs_motion=float((f - gt[:, :, 16:-16, 16:-16]).pow(2).sum(1).sqrt().mean()))
                    rows.append(row)
                    print(tag, row, flush=True)
            res[tag] = rows
    finally:
        pwc.WarpingLayer_no_div.forward = old_fwd
        mu.upsample2d_flow_as = old_up
    json.dump(res, open(os.path.join(HERE, 'train_traj_128x192.json'), 'w'), indent=1)


def golden_census():
    """utils/loss.py:50-91 (loss_functions.census_loss_torch): the soft census distance is only reachable through the
    reduced loss, so the fixture holds the reference's scalar for several masks / reductions plus its gradient wrt
    the warped image."""
    from utils.loss import loss_functions
    for i, (B, H, W) in enumerate([(2, 24, 40), (1, 13, 17)]):
        g = gen(7000 + i)
        im1 = torch.rand(B, 3, H, W, generator=g) - 0.45
        im2 = im1 + 0.1 * torch.randn(B, 3, H, W, generator=g)
        masks = (torch.rand(3, B, 1, H, W, generator=g) > 0.3).float()
        d = {'img1': im1, 'img1_warp': im2, 'masks': masks}
        for k in range(3):
            w = im2.clone().requires_grad_(True)
            v = loss_functions.census_loss_torch(img1=im1, img1_warp=w, mask=masks[k], q=0.4, charbonnier_or_abs_robust=False,
                                                 if_use_occ=True, averge=True)
            (gw,) = torch.autograd.grad(v, w)
            d['loss_occ_%d' % k] = np.array([float(v)])
            d['grad_occ_%d' % k] = gw
        v = loss_functions.census_loss_torch(img1=im1, img1_warp=im2, mask=masks[0], q=0.4, charbonnier_or_abs_robust=False,
                                             if_use_occ=False, averge=True)
        d['loss_mean'] = np.array([float(v)])
        save('census_%d' % i, **d)


def golden_losses(upflow, tools):
    """Loss-side operators: tools.boundary_dilated_warp.warp_im (utils/tools.py:351-499) with its
    gradient wrt the flow; network_tools.photo_loss_multi_type('abs_robust') with and without the occlusion weighting
    (model/upflow.py:265-288) and its gradients; network_tools.edge_aware_smoothness_order1 (:197-216) and its gradient."""
    nt = upflow.network_tools
    for i, (B, C, Hi, Wi, h, w, sx, sy) in enumerate([(2, 3, 40, 56, 24, 40, 8, 8), (1, 3, 20, 30, 20, 30, 0, 0), (1, 2, 9, 7, 5, 4, 2, 3)]):
        g = gen(7100 + i)
        I = torch.rand(B, C, Hi, Wi, generator=g) - 0.45
        flow = (torch.randn(B, 2, h, w, generator=g) * 4).requires_grad_(True)        # some samples leave the frame: clamping
        start = torch.tensor([sx, sy], dtype=torch.float32).view(1, 2, 1, 1).repeat(B, 1, 1, 1)
        out = tools.boundary_dilated_warp.warp_im(I, flow, start)
        go = torch.randn(out.shape, generator=g)
        (gf,) = torch.autograd.grad(out, flow, go)
        save('bwarp_%d' % i, image=I, flow=flow, start=start, out=out, grad_out=go, gflow=gf)
    for i, (B, C, H, W) in enumerate([(2, 3, 24, 40), (1, 2, 13, 17)]):
        g = gen(7200 + i)
        x = (torch.rand(B, C, H, W, generator=g) - 0.45).requires_grad_(True)
        y = (x.detach() + 0.2 * torch.randn(B, C, H, W, generator=g)).requires_grad_(True)
        y.data[0, 0, 0, :3] = x.data[0, 0, 0, :3]                                       # exact zeros: sign(0) = 0 in the gradient
        occ = (torch.rand(B, 1, H, W, generator=g) > 0.3).float()
        d = {'x': x, 'y': y, 'occ': occ}
        for use_occ in (False, True):
            v = nt.photo_loss_multi_type(x, y, occ, photo_loss_type='abs_robust', photo_loss_delta=0.4, photo_loss_use_occ=use_occ)
            gx, gy = torch.autograd.grad(v, (x, y))
            tag = 'occ' if use_occ else 'mean'
            d.update({'loss_' + tag: np.array([float(v)]), 'gx_' + tag: gx, 'gy_' + tag: gy})
        save('robust_%d' % i, **d)
    for i, (B, H, W) in enumerate([(2, 24, 40), (1, 7, 5)]):
        g = gen(7300 + i)
        img = torch.rand(B, 3, H, W, generator=g) - 0.45
        pred = (torch.randn(B, 2, H, W, generator=g) * 2).requires_grad_(True)
        pred.data[0, 0, 1, :2] = pred.data[0, 0, 0, :2]                                 # exact zero differences
        v = nt.edge_aware_smoothness_order1(img=img, pred=pred)
        (gp,) = torch.autograd.grad(v, pred)
        save('smooth1_%d' % i, img=img, pred=pred, loss=np.array([float(v)]), gpred=gp)


def golden_eval(tools):
    """Data / evaluation edge pinned on the reference's own code:
      * tools.write_flo / write_flow / read_flo / read_flow (utils/tools.py:1557-1632, numpy only): the bytes the reference
        writes for seeded flows, and — checked HERE, at generation time — that the reference's readers return the same array
        from a file written by the build's writer;
      * tools.write_flow_png / write_kitti_png_file (:1482-1525): the reference hands a uint16 array to pypng / cv2, which are
        not installed; capturing stand-ins for `png.Writer` / `cv2.imwrite` record that array (the quantisation arithmetic is
        the reference's, the PNG container is the codec's);
      * img_func.read_png_flow / get_process_img_only_img / frame_name_to_num (dataset/kitti_dataset.py:67-147), with
        `png.Reader` standing on the build's decoder (the decode arithmetic is the reference's);
      * kitti_flow.Evaluation_bench.flow_error_avg / outlier_pct (:464-499, torch only) on seeded flows and masks, incl.
        an empty mask, a full mask, every pixel an outlier, relative=None.
    dataset/kitti_dataset.py imports tensorflow / torchvision / cv2 / png / imageio at module level: empty stub modules."""
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from upflow_pytorch_amd.utils import flow_io
    for m in ['tensorflow', 'torchvision', 'rarfile', 'h5py', 'skimage', 'thop']:
        sys.modules.setdefault(m, types.ModuleType(m))
    tv = sys.modules['torchvision']
    if not hasattr(tv, 'transforms'):
        tv.transforms = types.ModuleType('torchvision.transforms')
        sys.modules['torchvision.transforms'] = tv.transforms
    if not hasattr(np, 'float'):
        np.float = float                                   # (numpy < 1.24 spelling used at kitti_dataset.py:143)
    import dataset.kitti_dataset as kd
    import utils.tools as tmod
    d = {}
    tmp = tempfile.mkdtemp()
    # ---- .flo
    rng = np.random.default_rng(4100)
    for i, (h, w) in enumerate([(7, 13), (1, 1), (5, 3)]):
        flow = (rng.normal(size=(h, w, 2)) * 20).astype(np.float32)
        pr, pb = os.path.join(tmp, 'r%d.flo' % i), os.path.join(tmp, 'b%d.flo' % i)
        (tools.write_flo if i != 1 else tools.write_flow)(flow, pr)
        flow_io.write_flo(flow, pb)
        # the reference's reader on the BUILD's file (tools.read_flow, :1583-1599, passes a numpy array as `count`, which
        # numpy 2 rejects; read_flo is the same code with int())
        assert np.array_equal(tools.read_flo(pb), flow)
        d['flo_flow_%d' % i] = flow
        d['flo_bytes_%d' % i] = np.frombuffer(open(pr, 'rb').read(), dtype=np.uint8)
    # ---- KITTI flow PNG: what the reference hands to pypng / cv2
    captured = {}

    class Writer(object):
        def __init__(self, **kw):
            captured['pypng_args'] = kw

        def write(self, f, rows):
            captured['pypng'] = np.array(rows)

    tmod.png.Writer = Writer
    tmod.cv2.imwrite = lambda fn, img: captured.__setitem__('cv2', np.array(img))
    uv = rng.normal(size=(9, 11, 2)) * 30
    uv[0, 0] = (600.0, -600.0)                              # outside the 16-bit range: write_flow_png clips
    uv[0, 1] = (0.0078125, -0.0078125)                      # half a quantisation step
    mask = (rng.random((9, 11)) > 0.4).astype(np.uint16)
    tools.write_flow_png(os.path.join(tmp, 'a.png'), uv, mask=mask)
    d['png_uv'], d['png_mask'], d['png_raw_pypng'] = uv, mask, captured['pypng'].reshape(9, 11, 3)
    assert captured['pypng_args'] == dict(width=11, height=9, bitdepth=16, compression=3, greyscale=False)
    tools.write_flow_png(os.path.join(tmp, 'a.png'), uv[:, :, 0], uv[:, :, 1])
    d['png_raw_pypng_nomask'] = captured['pypng'].reshape(9, 11, 3)
    uv2 = np.clip(uv, -500, 500)
    tools.write_kitti_png_file(os.path.join(tmp, 'b.png'), uv2, mask)
    d['png_uv_cv2'], d['png_raw_cv2_bgr'] = uv2, captured['cv2']
    # ---- reading: the reference's decode arithmetic on the raw samples of a file the build wrote
    pk = os.path.join(tmp, 'k.png')
    flow_io.write_kitti_png_file(pk, uv2, mask)

    class Reader(object):
        def __init__(self, path):
            self.raw = flow_io.read_png(path)

        def asDirect(self):
            h, w, c = self.raw.shape
            return w, h, [row.reshape(-1) for row in self.raw], {}

    kd.png.Reader = Reader
    f, m = kd.img_func.read_png_flow(pk)
    d['png_file_bytes'] = np.frombuffer(open(pk, 'rb').read(), dtype=np.uint8)
    d['png_read_flow'], d['png_read_mask'] = np.asarray(f, dtype=np.float64), m
    # ---- frame normalisation / names
    img = rng.integers(0, 256, size=(6, 10, 3)).astype(np.uint8)
    d['img'] = img
    d['img_norm'] = kd.img_func.get_process_img_only_img(img, normalize=True, if_horizontal_flip=False)
    d['img_norm_flip'] = kd.img_func.get_process_img_only_img(img, normalize=True, if_horizontal_flip=True)
    d['img_raw'] = kd.img_func.get_process_img_only_img(img, normalize=False, if_horizontal_flip=False)
    names = ['000000.png', '000123.png', '10.png', '0.png', '000.jpg']     # (the reference parses what precedes the first dot)
    d['frame_nums'] = np.array([kd.img_func.frame_name_to_num(n) for n in names])
    d['frame_names'] = np.array(names)
    # ---- EPE / F1
    EB = kd.kitti_flow.Evaluation_bench
    g = gen(4200)
    gt = torch.randn(3, 2, 12, 17, generator=g) * 20
    pred = gt + torch.randn(3, 2, 12, 17, generator=g) * 3
    masks = {'rand': (torch.rand(3, 1, 12, 17, generator=g) > 0.4).float(), 'full': torch.ones(3, 1, 12, 17),
             'empty': torch.zeros(3, 1, 12, 17)}
    d['ev_gt'], d['ev_pred'] = gt, pred
    for k, mk in masks.items():
        d['ev_mask_' + k] = mk
        d['ev_epe_' + k] = np.array([float(EB.flow_error_avg(gt, pred, mk))])
        d['ev_f1_' + k] = np.array([float(EB.outlier_pct(gt, pred, mk))])            # (empty mask: 0 / 0 = nan)
    d['ev_f1_rand_abs'] = np.array([float(EB.outlier_pct(gt, pred, masks['rand'], threshold=2.0, relative=None))])
    d['ev_f1_rand_t1'] = np.array([float(EB.outlier_pct(gt, pred, masks['rand'], threshold=1.0, relative=0.1))])
    far = gt + 100.0
    d['ev_epe_all_outliers'] = np.array([float(EB.flow_error_avg(gt, far, masks['rand']))])
    d['ev_f1_all_outliers'] = np.array([float(EB.outlier_pct(gt, far, masks['rand']))])
    d['ev_f1_exact'] = np.array([float(EB.outlier_pct(gt, gt, masks['rand']))])
    save('eval_edge', **d)


def golden_train_realistic(upflow, pwc, tools):
    """the training vector at REALISTIC motion.  train_128x192 uses head_scale 0.1 (sub-pixel flows): the
    boundary-dilated warp never leaves the crop there and the occlusion masks are almost empty.  Here the heads are full scale
    (mean |flow| ~10 px), the crop is 128x416 with its corner 3 / 2 px from the corner of the 144x448 frame (_weights.TRAIN_HS1):
    the photometric warp samples outside the crop and is clamped at the frame border (utils/tools.py:351-499), the forward /
    backward check masks a large share of the pixels (utils/tools.py:501-677), and the warp gradients see flows that leave the
    frame.  Same recorded quantities as golden_train, plus the statistics that show the regime (share of occluded pixels, share of
    photometric samples outside the crop / outside the frame)."""
    import model.upflow as mu
    old_fwd = robust_mask_patch(pwc)
    old_up = mu.upsample2d_flow_as
    mu.upsample2d_flow_as = oop_upsample2d_flow_as
    try:
        net = build_net(upflow, extra=_weights.TRAIN_FLAGS, head_scale=1.0)
        net.train()
        batch = _weights.make_train_batch(**_weights.TRAIN_HS1)
        batch['if_loss'] = True
        out = net(batch)
        terms = {k: out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')}
        loss = sum(terms.values())
        loss.backward()
        names = sorted(n for n, _ in net.named_parameters())
        params = dict(net.named_parameters())
        gnorm = np.array([float(params[n].grad.norm()) for n in names], dtype=np.float64)
        proj = _weights.grad_projections({n: params[n].grad for n in names})
        bias = {'gbias_%d' % i: params[n].grad for i, n in enumerate(names) if n.endswith('.bias')}
        f = out['flow_f_out'].detach()
        B, _, h, w = f.shape
        H, W = batch['im1_raw'].shape[2:]
        sx, sy = _weights.TRAIN_HS1['start_xy']
        xx = torch.arange(w).view(1, 1, w) + f[:, 0]
        yy = torch.arange(h).view(1, h, 1) + f[:, 1]
        out_crop = float(((xx < 0) | (xx > w - 1) | (yy < 0) | (yy > h - 1)).float().mean())
        out_frame = float(((xx + sx < 0) | (xx + sx > W - 1) | (yy + sy < 0) | (yy + sy > H - 1)).float().mean())
        regime = np.array([float(f.pow(2).sum(1).sqrt().mean()), float(1.0 - out['occ_fw'].float().mean()), out_crop, out_frame])
        save('train_128x416_hs1', loss=np.array([float(loss)]), **{k: np.array([float(v)]) for k, v in terms.items()},
             grad_norms=gnorm, grad_proj=proj, flow_f_out=out['flow_f_out'], flow_b_out=out['flow_b_out'],
             occ_fw=out['occ_fw'].to(torch.uint8), occ_bw=out['occ_bw'].to(torch.uint8), regime=regime, **bias)
        print({k: float(v) for k, v in terms.items()}, 'grad norm sum', gnorm.sum())
        print('regime: mean |flow| %.2f px, occluded %.1f %%, photometric samples outside the crop %.1f %%, outside the frame %.1f %%'
              % (regime[0], 100 * regime[1], 100 * regime[2], 100 * regime[3]))
        net.zero_grad()
        b16 = {k: (v.bfloat16().float() if k in ('im1', 'im2', 'im1_raw', 'im2_raw') else v) for k, v in batch.items()}
        out16 = net(b16)
        sum(out16[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')).backward()
        proj16 = _weights.grad_projections({n: params[n].grad for n in names})
        cos = (proj * proj16).sum(1) / (np.linalg.norm(proj, axis=1) * np.linalg.norm(proj16, axis=1))
        print('reference vs reference-with-bf16-rounded-frames: gradient cosine min %.5f median %.5f' % (cos.min(), np.median(cos)))
        z = dict(np.load(os.path.join(HERE, 'train_128x416_hs1.npz')))
        z['grad_proj_bf16_frames'] = proj16
        z['loss_bf16_frames'] = np.array([float(out16[k].mean()) for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')])
        np.savez_compressed(os.path.join(HERE, 'train_128x416_hs1.npz'), **z)
    finally:
        pwc.WarpingLayer_no_div.forward = old_fwd
        mu.upsample2d_flow_as = old_up


def main():
    torch.set_num_threads(8)
    upflow, pwc, tools, Corr_pyTorch = import_reference()
    which = sys.argv[1:] or ['corr', 'warp', 'upsample', 'normalize', 'sgu', 'occ', 'census', 'losses', 'eval', 'net', 'net384', 'neths1', 'train', 'traj', 'trainhs1']
    if 'corr' in which:
        golden_corr(Corr_pyTorch)
    if 'warp' in which:
        golden_warp(pwc, tools)
    if 'upsample' in which:
        golden_upsample(pwc)
    if 'normalize' in which:
        golden_normalize(upflow)
    if 'sgu' in which:
        golden_sgu_blend(upflow, tools)
    if 'occ' in which:
        golden_occ(tools)
    if 'census' in which:
        golden_census()
    if 'losses' in which:
        golden_losses(upflow, tools)
    if 'eval' in which:
        golden_eval(tools)
    if 'net' in which:
        golden_net(upflow, pwc, tools)
    if 'net384' in which:
        golden_net_headline(upflow, pwc, tools)
    if 'neths1' in which:
        golden_net_realistic(upflow, pwc, tools)
    if 'train' in which:
        golden_train(upflow, pwc, tools)
    if 'traj' in which:
        golden_train_trajectory(upflow, pwc, tools)
    if 'trainhs1' in which:
        golden_train_realistic(upflow, pwc, tools)


if __name__ == '__main__':
    main()
