"""Whole-network CPU restatement of UPFlow_net's inference forward (TEST INFRASTRUCTURE).

Functional, state_dict-driven: convolutions are ATen `conv2d` on CPU (third-party arithmetic, the
same the reference calls through nn.Conv2d), every hot-path operator comes from oracle/ops.py.
Follows model/upflow.py:370-392 (forward), :494-533 (forward_2_frame_v3), :535-573
(decode_level_res), :71-89 (sgu_model.forward) with the flags of test.py:22-30.

`corr='unfold'` routes the cost volume through the reference's fallback algorithm
(utils/pytorch_correlation.py:27-50) — that is BASELINE config 1 and the cpu_baseline of bench.py.
"""
import torch
import torch.nn.functional as F

from . import ops

_DIL = [1, 2, 4, 8, 16, 1, 1]          # ContextNetwork_v2_: model/pwc_modules.py:401-409


def _conv(sd, key, x, stride=1, dilation=1, relu=True):
    """`conv` factory of model/pwc_modules.py:10-49: Conv2d(pad=((k-1)*dil)//2) [+ LeakyReLU(0.1)]."""
    w = sd[key + '.0.weight']
    b = sd[key + '.0.bias']
    k = w.shape[-1]
    y = F.conv2d(x, w, b, stride=stride, padding=((k - 1) * dilation) // 2, dilation=dilation)
    return F.leaky_relu(y, 0.1) if relu else y


def _dense(sd, prefix, x):
    """FlowEstimatorDense_v2.forward (model/pwc_modules.py:279-286) and the SGU's private copy
    (model/upflow.py:53-60): five dense 3x3 convs, new features concatenated IN FRONT."""
    for n in ('conv1', 'conv2', 'conv3', 'conv4', 'conv5'):
        x = torch.cat([_conv(sd, '%s.%s' % (prefix, n), x), x], dim=1)
    return x, _conv(sd, prefix + '.conv_last', x, relu=False)


def _pyramid(sd, x):
    """FeatureExtractor.forward, model/pwc_modules.py:136-142 (coarsest first)."""
    feats = []
    for l in range(6):
        x = _conv(sd, 'feature_pyramid_extractor.convs.%d.0' % l, x, stride=2)
        x = _conv(sd, 'feature_pyramid_extractor.convs.%d.1' % l, x)
        feats.append(x)
    return feats[::-1]


def _sgu(sd, flow_init, f1, f2, mask_mode, output_level_flow=None):
    """sgu_model.forward, model/upflow.py:71-89."""
    if flow_init.shape[2:] != f1.shape[2:]:
        flow_init = ops.flow_upsample(flow_init, f1.shape[2], f1.shape[3], True)
    f2w = ops.warp(f2, flow_init, mask_mode)
    _, x_out = _dense(sd, 'sgi_model.dense_estimator_mask', torch.cat([f1, f2w], 1))
    return ops.sgu_blend(flow_init, x_out, output_level_flow)[1]


def _corr(a, b, corr):
    return ops.corr81_unfold(a, b) if corr == 'unfold' else ops.corr81(a, b)


def forward(sd, im1, im2, mask_mode='literal', corr='direct', sgu=True, return_levels=False):
    """-> dict(flow_f_out, flow_b_out, occ_fw, occ_bw[, flows])   (model/upflow.py:384-392)."""
    p1 = _pyramid(sd, im1)
    p2 = _pyramid(sd, im2)
    B, _, h0, w0 = p1[0].shape
    flow_f = torch.zeros(B, 2, h0, w0)
    flow_b = torch.zeros(B, 2, h0, w0)
    flows = []
    for level in range(5):                                               # output_level = 4, :512
        x1, x2 = p1[level], p2[level]
        a1 = _conv(sd, 'conv_1x1.%d' % level, x1)
        a2 = _conv(sd, 'conv_1x1.%d' % level, x2)
        h, w = x1.shape[2:]
        up_f = ops.flow_upsample(flow_f, h, w, True)                     # :536-537
        up_b = ops.flow_upsample(flow_b, h, w, True)
        if level == 0:
            x2w, x1w = x2, x1
        else:
            if sgu:
                up_f = _sgu(sd, up_f, a1, a2, mask_mode)                 # :544-545
                up_b = _sgu(sd, up_b, a2, a1, mask_mode)
            x2w = ops.warp(x2, up_f, mask_mode)                          # :546-547
            x1w = ops.warp(x1, up_b, mask_mode)
        n1, n2w = ops.normalize_pair(x1, x2w)                            # :550-555
        n2, n1w = ops.normalize_pair(x2, x1w)
        c1 = F.leaky_relu(_corr(n1, n2w, corr), 0.1)                     # :557-564
        c2 = F.leaky_relu(_corr(n2, n1w, corr), 0.1)
        feat1, res1 = _dense(sd, 'flow_estimators', torch.cat([c1, a1, up_f], 1))
        feat2, res2 = _dense(sd, 'flow_estimators', torch.cat([c2, a2, up_b], 1))
        fine = []
        for feat, up, res in ((feat1, up_f, res1), (feat2, up_b, res2)):
            x = torch.cat([feat, up + res], 1)                           # :567-570
            for i in range(7):
                x = _conv(sd, 'context_networks.convs.%d' % i, x, dilation=_DIL[i], relu=(i < 6))
            fine.append(x)
        flow_f = up_f + (res1 + fine[0])                                 # :571-573, :519-520
        flow_b = up_b + (res2 + fine[1])
        flows.append((flow_f, flow_b))
    H, W = im1.shape[2:]
    out_f = ops.flow_upsample(flow_f, H, W, True)                        # :522-523
    out_b = ops.flow_upsample(flow_b, H, W, True)
    if sgu:                                                              # :526-530
        def oc(x):
            x = _conv(sd, 'sgi_model.upsample_output_conv.0', x)
            x = _conv(sd, 'sgi_model.upsample_output_conv.1', x, stride=2)
            x = _conv(sd, 'sgi_model.upsample_output_conv.2', x)
            return _conv(sd, 'sgi_model.upsample_output_conv.3', x, stride=2)
        g1, g2 = oc(im1), oc(im2)
        out_f = _sgu_final(sd, flow_f, g1, g2, mask_mode, out_f)
        out_b = _sgu_final(sd, flow_b, g2, g1, mask_mode, out_b)
    occ_fw, occ_bw = ops.occ_check(out_f, out_b)                         # :386
    out = {'flow_f_out': out_f, 'flow_b_out': out_b, 'occ_fw': occ_fw, 'occ_bw': occ_bw}
    if return_levels:
        out['flows'] = flows[::-1]
    return out


def _sgu_final(sd, flow, f1, f2, mask_mode, output_level_flow):
    return _sgu(sd, flow, f1, f2, mask_mode, output_level_flow=output_level_flow)
