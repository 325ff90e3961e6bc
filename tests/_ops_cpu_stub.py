"""TEST-ONLY stand-in for the operators of upflow_pytorch_amd.ops on a box without a GPU.

The product has no CPU path (ops raise on CPU tensors).  To cover the N>1 code that the driver runs with RCCL —
UPFlow_net's training forward, train.Trainer, parallel.ddp_wrap, bench.py's rank plumbing — with two `gloo`
processes on CPU, install() replaces the operator entry points the TRAINING forward calls by the oracle's
differentiable restatements (oracle/ops.py).  Only tests/ may do this; nothing here is importable from the package."""
import torch

from oracle import ops as oops


def _warp(x, flow, mask_mode='literal', batch_shift=0):
    mode = {None: None, 'none': None, 0: None, 'literal': 'literal', 1: 'literal', 'robust': 'robust', 2: 'robust'}[mask_mode]
    if batch_shift:
        x = torch.roll(x, shifts=-int(batch_shift), dims=0)      # item n samples x[(n + shift) % B]
    return oops.warp(x, flow.float(), mode)


def _sgu_blend(flow_init, x_out, output_level_flow=None, want_inter=True):
    return oops.sgu_blend(flow_init, x_out, output_level_flow)


def _census_distance(gray1, gray2, max_distance=3):
    """oracle.census_distance on grey images (the product passes grey, utils/loss.py:52-55 of the reference converts)."""
    import torch.nn.functional as F
    patch = 2 * max_distance + 1
    n = patch * patch

    def ternary(gray):
        weight = torch.eye(n, dtype=gray.dtype).view(n, 1, patch, patch)
        t = F.conv2d(gray, weight, padding=max_distance) - gray
        return t / torch.sqrt(0.81 + t ** 2)
    d = (ternary(gray1) - ternary(gray2)) ** 2
    return torch.sum(d / (0.1 + d), 1, keepdim=True)


def install():
    from upflow_pytorch_amd import ops
    ops.warp = _warp
    ops.flow_upsample = lambda x, h, w, if_rate=True: oops.flow_upsample(x.float(), int(h), int(w), if_rate)
    ops.sgu_blend = _sgu_blend
    ops.normalize = lambda x: oops.normalize_pair(x, x)[0]
    ops.corr81 = lambda a, b, slope=0.0: (oops.corr81(a, b) if not slope else torch.nn.functional.leaky_relu(oops.corr81(a, b), slope))

    def corr81_forward_raw(a, b, out=None, leaky_slope=0.0):
        r = ops.corr81(a, b, leaky_slope)
        return r if out is None else out.copy_(r)
    ops.corr81_forward_raw = corr81_forward_raw
    ops.occ_check = lambda ff, fb, a1=0.1, a2=0.5: oops.occ_check(ff.detach().float(), fb.detach().float(), a1, a2)
    ops.census_distance = _census_distance
    ops.boundary_warp = oops.boundary_warp
    ops.robust_loss_sums = oops.robust_loss_sums
    ops.smooth_edge1 = oops.smooth_edge1
    # Correlation's autograd Function calls corr81_forward_raw / corr81_backward_raw explicitly
    ops.corr81_backward_raw = lambda a, b, go: oops.corr81_backward(a, b, go)
    return ops
