#!/usr/bin/env python3
"""Where does the bf16 TRAINING mode's distance to the reference come from?  (at realistic motion the gradient
cosines are min 0.910 / median 0.985 against 0.983 / 0.998 for the reference's own step with bf16-rounded frames; "nothing localises it".)

The config-3-style step of tests/test_hip_train.py (`train_128x416_hs1`: full-scale heads, mean |flow| 11.4 px, a crop at the frame's
corner) is run in the bf16 matrix-core mode with ONE part of the network at a time moved back to fp32 (its inputs cast to fp32 on the way
in — under autograd fp32 tensors take the PyTorch-ROCm path — and its outputs cast back to bf16 for the rest), and every variant is
scored against the reference's golden step: forward-flow EPE, loss terms, gradient norms, gradient direction (cosines of the 64 seeded
projections per parameter).  Variants:
    bf16                the training mode as it ships
    fp32                the parity mode
    pyramid             feature pyramid in fp32 (frames not rounded), features rounded to bf16 once at its outputs
    pyramid+1x1         ... and the 1x1 projections
    decoder             per level: cost volume + flow estimator + context network in fp32 (features arrive in bf16)
    sgu                 the SGU module (stack, blend) in fp32
    pyramid+decoder / pyramid+sgu / all-but-pyramid
    fp32, round_pyr     control: everything fp32, only the pyramid's output features rounded to bf16 once
    fp16                train_conv_dtype = 'fp16' (no loss scaling)
    python tools/precision_localise_train.py > profiles/r06_train_precision_localise.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from upflow_pytorch_amd import synthetic  # noqa: E402
from upflow_pytorch_amd.model.upflow import UPFlow_net  # noqa: E402
from upflow_pytorch_amd.model import upflow as mu  # noqa: E402

FLAGS = {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
         'norm_moments_across_images': False, 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}


def load_golden(name):
    z = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def f32(t):
    return t.float() if torch.is_tensor(t) and t.is_floating_point() else t


def b16(t):
    return t.to(torch.bfloat16) if torch.is_tensor(t) and t.dtype == torch.float32 else t


def patch(net, parts):
    """Move `parts` of the bf16-mode network to fp32 by wrapping their entry points."""
    if 'pyramid' in parts:
        fpe = net.feature_pyramid_extractor
        orig = fpe.forward
        frames = {}
        # the bf16 mode hands the pyramid X.to(bf16): keep the fp32 frames aside and feed those
        net._pyr_frames = frames

        def pyr(x, outs=None, pitched=False, _orig=orig):
            src = frames.get('X', x)
            return [b16(p) for p in _orig(f32(src))]
        fpe.forward = pyr
        # capture the fp32 frames before the cast
        orig_f2 = net._forward_2_frame_v3

        def f2(x1_raw, x2_raw, if_loss, cdt, _o=orig_f2):
            frames['X'] = torch.cat([x1_raw, x2_raw], 0)
            return _o(x1_raw, x2_raw, if_loss, cdt)
        net._forward_2_frame_v3 = f2
    if 'round_pyr' in parts:
        # control: everything fp32, the pyramid's OUTPUT features rounded to bf16 once (the smallest feature-sized perturbation)
        fpe = net.feature_pyramid_extractor
        o = fpe.forward
        fpe.forward = (lambda x, outs=None, pitched=False, _o=o: [p.to(torch.bfloat16).float() for p in _o(x)])
    if '1x1' in parts:
        for seq in net.conv_1x1:
            o = seq.forward
            seq.forward = (lambda x, _o=o: b16(_o(f32(x))))
        # (UPFlow_net._forward_stacked calls fast_conv_seq(self.conv_1x1[level], Fm, ...): patch that entry too)
        orig_fcs = mu.fast_conv_seq
        ids = {id(s) for s in net.conv_1x1}

        def fcs(seq, x, cache, out=None, allow_hip=True, pitched=False, _o=orig_fcs):
            if id(seq) in ids:
                return b16(_o(seq, f32(x), cache, out=out, allow_hip=allow_hip, pitched=pitched))
            return _o(seq, x, cache, out=out, allow_hip=allow_hip, pitched=pitched)
        mu.fast_conv_seq = fcs
    if 'decoder' in parts:
        o = net._level_update
        net._level_update = (lambda Fn, Fwn, A, flow_up, add_to_flow=False, _o=o: _o(f32(Fn), f32(Fwn), f32(A), flow_up, add_to_flow=add_to_flow))
    if 'sgu' in parts:
        o = net.sgi_model.forward
        net.sgi_model.forward = (lambda flow_init, f1, f2, output_level_flow=None, batch_shift=0, _o=o:
                                 _o(flow_init, f32(f1), f32(f2), output_level_flow=output_level_flow, batch_shift=batch_shift))
        oc = net.sgi_model.output_conv
        net.sgi_model.output_conv = (lambda x, out=None, out8=None, pitched=False, _o=oc: _o(f32(net._pyr_frames.get('X', x)) if hasattr(net, '_pyr_frames') else f32(x)))


# ---- part 2: the fp32 step with bf16 ROUNDING injected at one class of tensors at a time (straight-through: x.to(bf16).float()) ----------
GROUPS = {'pyramid': 'feature_pyramid_extractor.', 'conv_1x1': 'conv_1x1.', 'estimator': 'flow_estimators.', 'context': 'context_networks.',
          'sgu_est': 'sgi_model.dense_estimator_mask.', 'sgu_stem': 'sgi_model.upsample_output_conv.'}


def rnd(t):
    return t.to(torch.bfloat16).float() if torch.is_tensor(t) and t.dtype == torch.float32 else t


def inject(net, classes):
    """Install rounding at the named tensor classes of the fp32-mode network; returns an undo list."""
    undo = []
    import torch.nn as nn

    def hook_out(mod):
        h = mod.register_forward_hook(lambda m, i, o: rnd(o))
        undo.append(h.remove)
    for c in classes:
        if c.startswith('weights:'):
            pre = GROUPS[c[8:]]
            for n, p_ in net.named_parameters():
                if n.startswith(pre):
                    p_.data = rnd(p_.data)
        elif c == 'images':
            o = net._forward_2_frame_v3
            net._forward_2_frame_v3 = (lambda a, b, if_loss, cdt, _o=o: _o(rnd(a), rnd(b), if_loss, cdt))
        elif c == 'pyr_out':
            fpe = net.feature_pyramid_extractor
            o = fpe.forward
            fpe.forward = (lambda x, outs=None, pitched=False, _o=o: [rnd(p) for p in _o(x)])
        elif c == 'pyr_hidden':
            for st in net.feature_pyramid_extractor.convs:
                hook_out(st[0])
        elif c == '1x1_out':
            ids = {id(q) for q in net.conv_1x1}
            o = mu.fast_conv_seq
            mu.fast_conv_seq = (lambda seq, x, cache, out=None, allow_hip=True, pitched=False, _o=o:
                                rnd(_o(seq, x, cache, out=out, allow_hip=allow_hip, pitched=pitched)) if id(seq) in ids else _o(seq, x, cache, out=out, allow_hip=allow_hip, pitched=pitched))
            undo.append(lambda _o=o: setattr(mu, 'fast_conv_seq', _o))
        elif c in ('warp_dec', 'warp_sgu'):
            wl = net.warping_layer if c == 'warp_dec' else net.sgi_model.warping_layer
            o = wl.forward
            wl.forward = (lambda x, flow, batch_shift=0, _o=o: rnd(_o(x, flow, batch_shift)))
        elif c == 'normalize':
            o = mu.network_tools.normalize_features
            mu.network_tools.normalize_features = classmethod(lambda cls, fl, *a, _o=o, **k: [rnd(t) for t in _o(fl, *a, **k)])
            undo.append(lambda _o=o: setattr(mu.network_tools, 'normalize_features', _o))
        elif c == 'corr':
            o = net._corr_leaky
            net._corr_leaky = (lambda a, b, _o=o: rnd(_o(a, b)))
        elif c in ('est_hidden', 'sgu_hidden'):
            st = net.flow_estimators if c == 'est_hidden' else net.sgi_model.dense_estimator_mask
            for nme in st._NAMES:
                hook_out(getattr(st, nme))
        elif c == 'ctx_hidden':
            for q in list(net.context_networks.convs)[:6]:
                hook_out(q)
        elif c == 'sgu_stem_hidden':
            for q in net.sgi_model.upsample_output_conv:
                hook_out(q)
        elif c == 'head_res':
            hook_out(net.flow_estimators.conv_last)
        elif c == 'head_fine':
            hook_out(net.context_networks.convs[6])
        elif c == 'sgu_x_out':
            hook_out(net.sgi_model.dense_estimator_mask.conv_last)
        elif c in ('flow_in_est', 'flow_in_ctx'):
            mod = net.flow_estimators if c == 'flow_in_est' else net.context_networks

            def pre(m, args):
                x = args[0]
                return (torch.cat([x[:, :-2], rnd(x[:, -2:])], 1),) + tuple(args[1:])
            h = mod.register_forward_pre_hook(pre)
            undo.append(h.remove)
        else:
            raise ValueError(c)
    return undo


INJECT_CLASSES = ['images', 'weights:pyramid', 'weights:conv_1x1', 'weights:estimator', 'weights:context', 'weights:sgu_est', 'weights:sgu_stem',
                  'pyr_hidden', 'pyr_out', '1x1_out', 'warp_dec', 'warp_sgu', 'normalize', 'corr', 'flow_in_est', 'est_hidden', 'head_res',
                  'flow_in_ctx', 'ctx_hidden', 'head_fine', 'sgu_stem_hidden', 'sgu_hidden', 'sgu_x_out']


def run_inject(classes):
    conf = UPFlow_net.config()
    d = dict(FLAGS)
    d.update(synthetic.TRAIN_FLAGS)
    d['train_conv_dtype'] = 'fp32'
    conf.update(d, verbose=False)
    net = conf()
    net.load_state_dict(synthetic.make_state_dict(0, head_scale=1.0))
    net = net.cuda().train()
    undo = inject(net, classes)
    try:
        batch = {k: v.cuda() for k, v in synthetic.make_train_batch(**synthetic.TRAIN_HS1).items()}
        batch['if_loss'] = True
        out = net(batch)
        terms = {k: out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')}
        sum(terms.values()).backward()
    finally:
        for u in undo:
            u()
    names = sorted(n for n, _ in net.named_parameters())
    params = dict(net.named_parameters())
    return out, terms, names, {n: params[n].grad for n in names}


def run(mode, parts):
    conf = UPFlow_net.config()
    d = dict(FLAGS)
    d.update(synthetic.TRAIN_FLAGS)
    d['train_conv_dtype'] = mode
    conf.update(d, verbose=False)
    net = conf()
    net.load_state_dict(synthetic.make_state_dict(0, head_scale=1.0))
    net = net.cuda().train()
    saved_fcs = mu.fast_conv_seq
    try:
        patch(net, parts)
        batch = {k: v.cuda() for k, v in synthetic.make_train_batch(**synthetic.TRAIN_HS1).items()}
        batch['if_loss'] = True
        out = net(batch)
        terms = {k: out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')}
        sum(terms.values()).backward()
    finally:
        mu.fast_conv_seq = saved_fcs
    names = sorted(n for n, _ in net.named_parameters())
    params = dict(net.named_parameters())
    return out, terms, names, {n: params[n].grad for n in names}


def main():
    g = load_golden('train_128x416_hs1')
    ref, r16 = g['grad_proj'].numpy(), g['grad_proj_bf16_frames'].numpy()
    floor = (ref * r16).sum(1) / (np.linalg.norm(ref, axis=1) * np.linalg.norm(r16, axis=1))
    print('# tools/precision_localise_train.py on %s — golden train_128x416_hs1 (mean |flow| %.1f px); the reference with bf16-rounded frames: cosine min %.4f median %.4f'
          % (torch.cuda.get_device_name(0), float(g['regime'][0]), floor.min(), np.median(floor)))
    print('%-22s %9s %9s %10s %10s %9s %9s   %s' % ('fp32 parts', 'flow EPE', 'loss rel', 'gnorm max', 'gnorm med', 'cos min', 'cos med', 'worst-cosine parameter'))
    variants = [('bf16', ()), ('fp32', None), ('fp32', ('round_pyr',)), ('fp16', ()), ('bf16', ('pyramid',)), ('bf16', ('pyramid', '1x1')), ('bf16', ('decoder',)), ('bf16', ('sgu',)),
                ('bf16', ('pyramid', '1x1', 'decoder')), ('bf16', ('pyramid', '1x1', 'sgu')), ('bf16', ('1x1', 'decoder', 'sgu'))]
    for mode, parts in variants:
        label = 'fp32 (parity mode)' if parts is None else ('%s (as shipped)' % mode if not parts else ('fp32, ' if mode == 'fp32' else '') + '+'.join(parts))
        try:
            out, terms, names, grads = run(mode, parts or ())
        except Exception as e:
            print('%-22s failed: %s: %s' % (label, type(e).__name__, str(e)[:120]))
            continue
        e = float((out['flow_f_out'].detach().float().cpu() - g['flow_f_out']).pow(2).sum(1).sqrt().mean())
        lrel = max(abs(float(v.detach()) - float(g[k])) / max(1.0, abs(float(g[k]))) for k, v in terms.items())
        got = np.array([float(grads[n].norm()) for n in names])
        want = g['grad_norms'].numpy()
        rel = np.abs(got - want) / np.maximum(want, 1e-3)
        gp = synthetic.grad_projections(grads)
        cos = (gp * ref).sum(1) / np.maximum(np.linalg.norm(gp, axis=1) * np.linalg.norm(ref, axis=1), 1e-30)
        print('%-22s %9.4f %9.5f %10.4f %10.4f %9.4f %9.4f   %s' % (label, e, lrel, rel.max(), np.median(rel), cos.min(), np.median(cos), names[int(cos.argmin())]), flush=True)
    print('# part 2: the fp32 step with bf16 rounding injected at ONE class of tensors (the weights of a group / the outputs of a group\'s hidden layers / an operator\'s output / the flow channels of an input)')
    for cls in [[c] for c in INJECT_CLASSES] + [INJECT_CLASSES, [c for c in INJECT_CLASSES if not c.startswith('weights:')], [c for c in INJECT_CLASSES if c.startswith('weights:')]]:
        label = cls[0] if len(cls) == 1 else ('ALL' if len(cls) == len(INJECT_CLASSES) else ('ALL activations' if not cls[0].startswith('weights:') or cls[0] == 'images' else 'ALL weights'))
        try:
            out, terms, names, grads = run_inject(cls)
        except Exception as e:
            print('%-22s failed: %s: %s' % (label, type(e).__name__, str(e)[:160]))
            continue
        e = float((out['flow_f_out'].detach().float().cpu() - g['flow_f_out']).pow(2).sum(1).sqrt().mean())
        lrel = max(abs(float(v.detach()) - float(g[k])) / max(1.0, abs(float(g[k]))) for k, v in terms.items())
        got = np.array([float(grads[n].norm()) for n in names])
        rel = np.abs(got - g['grad_norms'].numpy()) / np.maximum(g['grad_norms'].numpy(), 1e-3)
        gp = synthetic.grad_projections(grads)
        cos = (gp * ref).sum(1) / np.maximum(np.linalg.norm(gp, axis=1) * np.linalg.norm(ref, axis=1), 1e-30)
        print('%-22s %9.4f %9.5f %10.4f %10.4f %9.4f %9.4f   %s' % (label, e, lrel, rel.max(), np.median(rel), cos.min(), np.median(cos), names[int(cos.argmin())]), flush=True)


if __name__ == '__main__':
    main()
