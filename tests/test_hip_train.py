"""Train-mode parity (BASELINE config 3 at a small crop): loss terms and per-parameter gradient norms of
the HIP path against the reference's (tests/golden/train_128x192.npz), robust mask on both sides."""
import numpy as np
import pytest
import torch

import oracle
import _weights
from conftest import load_golden

pytestmark = pytest.mark.gpu
FLAGS = {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
         'norm_moments_across_images': False, 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}


def build():
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    conf = UPFlow_net.config()
    d = dict(FLAGS)
    d.update(_weights.TRAIN_FLAGS)
    conf.update(d, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
    return net.cuda().train()


def test_train_losses_and_gradients_match_reference():
    g = load_golden('train_128x192')
    net = build()
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
    batch['if_loss'] = True
    out = net(batch)
    terms = {k: out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')}
    for k, v in terms.items():
        want = float(g[k])
        print(k, float(v), want)
        assert abs(float(v) - want) <= 2e-4 * max(1.0, abs(want)), k
    assert oracle.epe(out['flow_f_out'].detach().cpu(), g['flow_f_out']) <= 1e-4
    sum(terms.values()).backward()
    names = sorted(n for n, _ in net.named_parameters())
    params = dict(net.named_parameters())
    got = np.array([float(params[n].grad.norm()) for n in names])
    want = g['grad_norms'].numpy()
    rel = np.abs(got - want) / np.maximum(want, 1e-3)
    print('max rel grad-norm error %.3g (param %s)' % (rel.max(), names[int(rel.argmax())]))
    assert (got > 0).all(), 'every parameter must receive a gradient'
    assert rel.max() <= 2e-2


def test_trainer_step_updates_every_parameter():
    """Trainer.step = forward + loss + backward + Adam(amsgrad) on the HIP path: finite loss terms, every
    parameter moves.  (No "loss goes down" claim: with random MSRA weights the self-distillation term makes
    the first steps of the reference's own recipe noisy.)"""
    from upflow_pytorch_amd.train import Trainer
    net = build()
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    tr = Trainer(net, lr=1e-4, distributed=False)
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
    stats = tr.step(batch)
    assert set(stats) == {'loss', 'photo_loss', 'smooth_loss', 'census_loss', 'msd_loss'}
    assert all(np.isfinite(v) for v in stats.values())
    moved = [n for n, p in net.named_parameters() if not torch.equal(p.detach(), before[n])]
    assert len(moved) == 80
