#!/usr/bin/env python3
"""Which parameter gradients of an fp32 (PyTorch-ROCm convolutions) training step are not laid out like their parameter?
torch.optim.Adam(fused=True) walks parameter and gradient memory linearly: a channels-last weight gradient from MIOpen would be
applied to the wrong elements (round 4: the fp32 step at config 3's full size diverged at step 1 with fused=True)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from upflow_pytorch_amd.train import synthetic_train_batch
from test_hip_train import _config3_trainer

for mode in ('fp32', 'bf16'):
    tr = _config3_trainer(mode, False)
    batch = dict(synthetic_train_batch(4, device='cuda'))
    batch['if_loss'] = True
    out = tr.net(batch)
    loss, parts = tr.loss_manager.compute_loss(out)
    loss.backward()
    bad = [(n, tuple(p.shape), p.stride(), p.grad.stride()) for n, p in tr.raw_net.named_parameters()
           if p.grad is not None and (p.grad.stride() != p.stride() or not p.grad.is_contiguous())]
    print(mode, 'parameters whose gradient is laid out differently:', len(bad))
    for b in bad[:12]:
        print('   ', b)
    del tr
    torch.cuda.empty_cache()
