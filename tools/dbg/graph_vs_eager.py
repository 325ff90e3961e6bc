"""graph vs eager parameter drift after 5 steps in bf16 mode, with the round-2 training features toggled."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from upflow_pytorch_amd import synthetic as _weights
from upflow_pytorch_amd.model.upflow import UPFlow_net
from upflow_pytorch_amd.train import Trainer
from upflow_pytorch_amd.model.pwc_modules import _DenseStack

FLAGS = {'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
         'norm_moments_across_images': False, 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}

def make(sinks, inbuf):
    conf = UPFlow_net.config()
    d = dict(FLAGS); d.update(_weights.TRAIN_FLAGS); d['train_conv_dtype'] = 'bf16'
    conf.update(d, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
    net.shared_grad_sinks = sinks
    for m in net.modules():
        if isinstance(m, _DenseStack):
            m._no_train_buffer = not inbuf
    return net

batch = {k: v.cuda() for k, v in _weights.make_train_batch().items()}
for sinks in (False, True):
    for inbuf in (False, True):
        res = []
        for graph in (False, False, True):
            tr = Trainer(make(sinks, inbuf), lr=1e-4, device=torch.device('cuda', 0), distributed=False, graph=graph)
            for _ in range(5):
                st = tr.step(batch)
            res.append(({n: p.detach().clone() for n, p in tr.raw_net.named_parameters()}, st))
        diff = lambda u, v: (max(float((u[n] - v[n]).abs().max()) for n in u), sum(float((u[n] - v[n]).abs().sum()) for n in u) / sum(u[n].numel() for n in u))
        print('sinks %d inbuf %d: graph-eager worst %.3g mean %.3g | eager-eager worst %.3g mean %.3g | loss %.6f %.6f' %
              ((sinks, inbuf) + diff(res[0][0], res[2][0]) + diff(res[0][0], res[1][0]) + (res[0][1]['loss'], res[2][1]['loss'])), flush=True)
