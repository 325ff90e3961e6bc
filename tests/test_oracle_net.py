"""Whole-network oracle (oracle/net.py) against the reference's outputs; BASELINE config 1
(256x256 pair on CPU through the reference's fallback correlation algorithm).  CPU only."""
import json
import os

import pytest
import torch

import oracle
from oracle import net as onet
import _weights
from conftest import load_golden, GOLDEN

META = json.load(open(os.path.join(GOLDEN, 'net_meta.json')))


def test_weight_recipe_is_stable():
    assert _weights.state_dict_sha256(_weights.make_state_dict(0)) == META['weights_sha256']
    assert _weights.state_dict_sha256(_weights.make_state_dict(0, head_scale=0.1)) == META['weights_sha256_hs']
    assert _weights.state_dict_sha256(_weights.make_state_dict(0, head_scale=1.0)) == META['weights_sha256_hs1']


@pytest.mark.parametrize('name,H,W,cid', [('net_64x128', 64, 128, 1), ('net_256x256', 256, 256, 1)])
def test_free_running_robust_mask(name, H, W, cid):
    """P3b: exact-predicate mask on both sides -> free-running EPE <= 1e-4."""
    sd = _weights.make_state_dict(0, head_scale=0.1)
    im1, im2 = _weights.make_smooth_images(cid, 1, H, W)
    g = load_golden(name + '_robust')
    with torch.no_grad():
        out = onet.forward(sd, im1, im2, mask_mode='robust', corr='unfold' if H == 256 else 'direct')
    assert oracle.epe(out['flow_f_out'], g['flow_f_out']) <= 1e-4
    assert oracle.epe(out['flow_b_out'], g['flow_b_out']) <= 1e-4
    assert (out['occ_fw'] != g['occ_fw'].float()).float().mean() <= 2e-3
    assert (out['occ_bw'] != g['occ_bw'].float()).float().mean() <= 2e-3


def test_free_running_robust_mask_headline_resolution():
    """The oracle at BASELINE config 2's resolution (384x1280) against the reference's output: EPE <= 1e-4."""
    import numpy as np
    sd = _weights.make_state_dict(0, head_scale=0.1)
    im1, im2 = _weights.make_smooth_images(2, 1, 384, 1280)
    g = load_golden('net_384x1280_robust')
    with torch.no_grad():
        out = onet.forward(sd, im1, im2, mask_mode='robust')
    assert oracle.epe(out['flow_f_out'], g['flow_f_out']) <= 1e-4
    occ = torch.from_numpy(np.unpackbits(g['occ_fw'].numpy())[:384 * 1280].reshape(1, 1, 384, 1280)).float()
    assert (out['occ_fw'] != occ).float().mean() <= 2e-3
    s, a = float(out['flow_b_out'].double().sum()), float(out['flow_b_out'].double().abs().sum())
    assert abs(a - float(g['flow_b_checksum'][1])) <= 1e-4 * 384 * 1280 * 2          # mean |d| <= 1e-4 on the backward flow too
    assert abs(s - float(g['flow_b_checksum'][0])) <= 1e-4 * 384 * 1280 * 2


@pytest.mark.parametrize('name,H,W,cids', [('net_256x256_hs1_robust', 256, 256, (1,)), ('net_384x1280_hs1_robust', 384, 1280, (2, 12))])
def test_free_running_realistic_motion(name, H, W, cids):
    """Round 4: the oracle against the reference's output with FULL-SCALE heads — mean |flow| 10.6 px
    (p99 21) at 256x256, 15.6 px (p99 34, max 56) at 384x1280: border masks, the +-4 search range and the SGU warp outside
    the sub-pixel regime.  Bar: 1e-4 px, or 3x the reference's own sensitivity to 1e-7 input noise where that is larger."""
    import numpy as np
    sd = _weights.make_state_dict(0, head_scale=1.0)
    ims = [_weights.make_smooth_images(c, 1, H, W) for c in cids]
    im1, im2 = torch.cat([a for a, _ in ims]), torch.cat([b for _, b in ims])
    g = load_golden(name)
    B = len(cids)
    assert abs(float(g['mean_flow_px'][0]) - META[name]['mean_flow_px']) < 1e-6 and META[name]['mean_flow_px'] > 10.0
    with torch.no_grad():
        out = onet.forward(sd, im1, im2, mask_mode='robust')
    bar = max(1e-4, 3 * META[name]['self_sensitivity_epe'])
    e = oracle.epe(out['flow_f_out'], g['flow_f_out'])
    print('%s oracle vs reference EPE %.3g px (mean |flow| %.3g px, bar %.3g)' % (name, e, META[name]['mean_flow_px'], bar))
    assert e <= bar
    fb = out['flow_b_out'] if H * W <= 256 * 256 else out['flow_b_out'][:, :, ::4, ::4]
    assert oracle.epe(fb, g['flow_b_out']) <= bar
    occ = torch.from_numpy(np.unpackbits(g['occ_fw'].numpy())[:B * H * W].reshape(B, 1, H, W)).float()
    assert (out['occ_fw'] != occ).float().mean() <= 2e-3


def test_free_running_literal_mask_vs_noise_floor():
    """P3a: literal `mask >= 1.0` semantics.  The reference is chaotic against itself here (its own
    output moves by META[...self_sensitivity] px under 1e-7 input noise), so the bar is that floor."""
    sd = _weights.make_state_dict(0, head_scale=0.1)
    im1, im2 = _weights.make_smooth_images(1, 1, 64, 128)
    g = load_golden('net_64x128_literal')
    with torch.no_grad():
        out = onet.forward(sd, im1, im2, mask_mode='literal')
    floor = META['net_64x128_literal_self_sensitivity_epe']
    e = oracle.epe(out['flow_f_out'], g['flow_f_out'])
    print('literal free-running EPE %.3g (reference self-sensitivity %.3g)' % (e, floor))
    assert e <= 3 * floor


def test_teacher_forced_trace():
    """P2: every hot-op call of the recorded reference forward replayed on its recorded inputs."""
    import _trace
    from oracle import ops

    class P:
        corr81 = staticmethod(ops.corr81)
        warp = staticmethod(ops.warp)
        flow_upsample = staticmethod(ops.flow_upsample)
        normalize_pair = staticmethod(ops.normalize_pair)
        sgu_blend = staticmethod(ops.sgu_blend)
    r = _trace.replay(P)
    print({k: max(v) for k, v in r['errs'].items()}, r['mask_mismatch'], r['final_epe'], r['n_events'])
    assert r['mask_mismatch'] == 0
    assert r['final_epe'] <= 1e-4
    for op, v in r['errs'].items():
        assert max(v) <= 2e-5, op
