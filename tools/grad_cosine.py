"""Training gradients of the HIP path vs the reference's recorded ones (golden train_128x192): per-parameter cosine similarity and relative norm, for tuning the tolerances of tests/test_hip_train.py."""
import sys, os
sys.path.insert(0, 'tests')
import numpy as np, torch
import _weights
from conftest import load_golden
import test_hip_train as T
from upflow_pytorch_amd.model.upflow import UPFlow_net
g = load_golden('train_128x192')
for mode in ('bf16', 'fp16'):
    conf = UPFlow_net.config(); d = dict(T.FLAGS); d.update(_weights.TRAIN_FLAGS); d['train_conv_dtype'] = mode
    conf.update(d, verbose=False); net = conf(); net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1)); net = net.cuda().train()
    batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}; batch['if_loss'] = True
    out = net(batch); sum(out[k].mean() for k in ('photo_loss', 'smooth_loss', 'census_loss', 'msd_loss')).backward()
    names = sorted(n for n, _ in net.named_parameters()); params = dict(net.named_parameters())
    cos, worst = T.grad_direction_check({n: params[n].grad for n in names}, g)
    order = np.argsort(cos)
    print(mode, 'median cos %.6f' % np.median(cos), 'worst bias err %.3g' % worst)
    for i in order[:12]:
        print('   %.5f  %s  (norm %.3g)' % (cos[i], names[i], float(params[names[i]].grad.norm())))
    print('   weights only: min %.5f' % min(cos[i] for i, n in enumerate(names) if n.endswith('weight')))
