"""Deterministic synthetic UPFlow_net weights and frame pairs (bench.py, smoke(), the golden generator, the tests).

The reference publishes no checkpoint (`/root/reference/.MISSING_LARGE_BLOBS:3`), so every
whole-network fixture uses MSRA-style weights (`model/pwc_modules.py:52-69`: kaiming-normal conv
weights, zero bias) drawn from a per-tensor CPU generator keyed by the parameter's position in the
sorted key list.  14 MB of weights are therefore never committed: only this recipe and a SHA-256.
"""
import hashlib
import math

import torch

# (key prefix, [(cin, cout, k)]) in reference construction order: `model/upflow.py:343-359`.
_EST = [(115, 128), (243, 128), (371, 96), (467, 64), (531, 32), (563, 2)]
_CTX = [(565, 128), (128, 128), (128, 128), (128, 96), (96, 64), (64, 32), (32, 2)]
_PYR = [3, 16, 32, 64, 96, 128, 196]
_SGU = [(64, 32), (96, 32), (128, 32), (160, 16), (176, 8), (184, 3)]
_SGU_OUT = [(3, 16), (16, 16), (16, 32), (32, 32)]


def param_shapes():
    """name -> shape for the 80 tensors of SURVEY.md §8(a12)."""
    shapes = {}
    for l in range(6):
        ci, co = _PYR[l], _PYR[l + 1]
        shapes[f'feature_pyramid_extractor.convs.{l}.0.0.weight'] = (co, ci, 3, 3)
        shapes[f'feature_pyramid_extractor.convs.{l}.0.0.bias'] = (co,)
        shapes[f'feature_pyramid_extractor.convs.{l}.1.0.weight'] = (co, co, 3, 3)
        shapes[f'feature_pyramid_extractor.convs.{l}.1.0.bias'] = (co,)
    names = ['conv1', 'conv2', 'conv3', 'conv4', 'conv5', 'conv_last']
    for n, (ci, co) in zip(names, _EST):
        shapes[f'flow_estimators.{n}.0.weight'] = (co, ci, 3, 3)
        shapes[f'flow_estimators.{n}.0.bias'] = (co,)
    for i, (ci, co) in enumerate(_CTX):
        shapes[f'context_networks.convs.{i}.0.weight'] = (co, ci, 3, 3)
        shapes[f'context_networks.convs.{i}.0.bias'] = (co,)
    for i, ci in enumerate([196, 128, 96, 64, 32]):
        shapes[f'conv_1x1.{i}.0.weight'] = (32, ci, 1, 1)
        shapes[f'conv_1x1.{i}.0.bias'] = (32,)
    for n, (ci, co) in zip(names, _SGU):
        shapes[f'sgi_model.dense_estimator_mask.{n}.0.weight'] = (co, ci, 3, 3)
        shapes[f'sgi_model.dense_estimator_mask.{n}.0.bias'] = (co,)
    for i, (ci, co) in enumerate(_SGU_OUT):
        shapes[f'sgi_model.upsample_output_conv.{i}.0.weight'] = (co, ci, 3, 3)
        shapes[f'sgi_model.upsample_output_conv.{i}.0.bias'] = (co,)
    return shapes


def make_state_dict(seed=0, head_scale=1.0, bias_std=0.0):
    """MSRA-like weights: N(0, 2/fan_in) conv kernels, biases zero (or N(0, bias_std²)).

    `head_scale` shrinks the 2-/3-channel output heads so the synthetic network produces flows of
    a few pixels instead of hundreds (keeps the warps inside the image, like a trained model).
    """
    sd = {}
    for idx, (name, shape) in enumerate(sorted(param_shapes().items())):
        g = torch.Generator().manual_seed(seed * 100003 + idx)
        if name.endswith('weight'):
            fan_in = shape[1] * shape[2] * shape[3]
            w = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
            if shape[0] in (2, 3):
                w = w * head_scale
            sd[name] = w
        else:
            sd[name] = torch.randn(shape, generator=g) * bias_std if bias_std > 0 else torch.zeros(shape)
    return sd


def state_dict_sha256(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].contiguous().numpy().tobytes())
    return h.hexdigest()


def make_images(config_id, B, H, W):
    """Synthetic frame pair of SURVEY.md §8(d): seed 1000+config_id, U(0,1) - 0.45."""
    g = torch.Generator().manual_seed(1000 + config_id)
    im1 = torch.rand(B, 3, H, W, generator=g) - 0.45
    im2 = torch.rand(B, 3, H, W, generator=g) - 0.45
    return im1, im2


def make_smooth_images(config_id, B, H, W, shift=2):
    """A low-pass random texture and a copy shifted by `shift` px: a pair with a real flow."""
    g = torch.Generator().manual_seed(3000 + config_id)
    base = torch.rand(B, 3, H // 4 + 4, W // 4 + 4, generator=g)
    big = torch.nn.functional.interpolate(base, size=(H + 16, W + 16), mode='bicubic', align_corners=True)
    im1 = big[:, :, 8:8 + H, 8:8 + W] - 0.45
    im2 = big[:, :, 8:8 + H, 8 - shift:8 - shift + W] - 0.45
    return im1.contiguous(), im2.contiguous()


def make_train_batch(B=2, crop_hw=(128, 192), raw_hw=(160, 256), seed=0, start_xy=None):
    """KITTI-style training batch: crops, the un-cropped frames and the crop offset `start`
    (scripts/ex_runner.py:146-147).  Smooth texture moved by 2 px so the photometric loss is meaningful.
    start_xy: crop offset (x, y) inside the un-cropped frame (default: centred) — near the border, flows that leave the crop also leave
    the frame, which is where tools.boundary_dilated_warp clamps (utils/tools.py:351-499)."""
    g = torch.Generator().manual_seed(4000 + seed)
    H, W = raw_hw
    h, w = crop_hw
    base = torch.rand(B, 3, H // 8 + 2, W // 8 + 2, generator=g)
    big = torch.nn.functional.interpolate(base, size=(H + 8, W + 8), mode='bicubic', align_corners=True) - 0.45
    im1 = big[:, :, 4:4 + H, 4:4 + W].contiguous()
    im2 = big[:, :, 4:4 + H, 2:2 + W].contiguous()
    sy, sx = (H - h) // 2, (W - w) // 2
    if start_xy is not None:
        sx, sy = int(start_xy[0]), int(start_xy[1])
    start = torch.tensor([sx, sy], dtype=torch.float32).view(1, 2, 1, 1).repeat(B, 1, 1, 1)
    return {'im1': im1[:, :, sy:sy + h, sx:sx + w].contiguous(), 'im2': im2[:, :, sy:sy + h, sx:sx + w].contiguous(),
            'im1_raw': im1, 'im2_raw': im2, 'start': start}


# the realistic-motion training vector (tests/golden/train_128x416_hs1.npz): full-scale heads (head_scale 1: flows
# of ~10 px), a crop whose corner sits 3 px / 2 px from the frame's, so that the boundary-dilated warp samples outside the crop AND is
# clamped at the frame border, and the occlusion masks are non-trivial
TRAIN_HS1 = dict(B=2, crop_hw=(128, 416), raw_hw=(144, 448), seed=3, start_xy=(3, 2))


TRAIN_FLAGS = {'photo_loss_census_weight': 1, 'multi_scale_distillation_weight': 1, 'multi_scale_distillation_style': 'upup',
               'multi_scale_distillation_occ': True, 'smooth_order_1_weight': 1, 'photo_loss_type': 'abs_robust',
               'photo_loss_delta': 0.4, 'photo_loss_use_occ': False, 'if_use_boundary_warp': True}


def grad_projections(named_grads, ndir=64, seed=9000):
    """Direction-sensitive fingerprint of a set of per-parameter gradients: for the parameters in sorted-name order, the
    inner products of the flattened gradient with `ndir` seeded N(0,1) directions (one CPU generator per parameter index),
    [P, ndir] float64.  Two gradients with the same norm but a different direction have different fingerprints — the cosine of
    two fingerprints estimates the cosine of the gradients (Johnson-Lindenstrauss) — so tests compare these, not norms.
    The golden generator applies it to the reference's gradients, the tests to the build's."""
    import numpy as np
    names = sorted(named_grads)
    out = np.zeros((len(names), ndir), dtype=np.float64)
    for i, n in enumerate(names):
        g = named_grads[n].detach().double().cpu().reshape(-1)
        gen = torch.Generator().manual_seed(seed + i)
        for lo in range(0, ndir, 8):                                         # (8 directions at a time: bounded memory)
            d = torch.randn(min(8, ndir - lo), g.numel(), generator=gen, dtype=torch.float64)
            out[i, lo:lo + d.shape[0]] = (d @ g).numpy()
    return out
