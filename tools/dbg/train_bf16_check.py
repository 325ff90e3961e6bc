import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
import _weights
from conftest import load_golden
from upflow_pytorch_amd.model.upflow import UPFlow_net
FLAGS = {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False, 'norm_moments_across_images': False, 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}
g = load_golden('train_128x192')
for mode in ('fp32', 'bf16', 'fp16'):
    conf = UPFlow_net.config(); d = dict(FLAGS); d.update(_weights.TRAIN_FLAGS); d['train_conv_dtype'] = mode; conf.update(d, verbose=False)
    net = conf(); net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1)); net = net.cuda().train()
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}; batch['if_loss'] = True
    out = net(batch)
    terms = {k: out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')}
    sum(terms.values()).backward()
    names = sorted(n for n, _ in net.named_parameters()); params = dict(net.named_parameters())
    got = np.array([float(params[n].grad.norm()) for n in names]); want = g['grad_norms'].numpy()
    rel = np.abs(got - want) / np.maximum(want, 1e-3)
    print(mode, {k: (round(float(v), 5), round(float(g[k]), 5)) for k, v in terms.items()}, 'grad-norm rel err: max %.3g median %.3g (worst %s)' % (rel.max(), np.median(rel), names[int(rel.argmax())]),
          'flow EPE vs fp32 golden %.4g' % float((out['flow_f_out'].detach().cpu() - g['flow_f_out']).pow(2).sum(1).sqrt().mean()))
