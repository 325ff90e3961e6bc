"""Training-loss helpers mirroring `/root/reference/utils/loss.py` (class `loss_functions`).
Plain torch ops (loss side, not the hot path: SURVEY.md §2 row 14)."""
import os
import torch
import torch.nn.functional as F


class _GreyFunction(torch.autograd.Function):
    """0.2989 r + 0.5870 g + 0.1140 b (utils/loss.py:53-55), evaluated left to right like the reference, as one autograd node:
    the backward is one broadcast multiply instead of the five nodes of the spelled-out expression."""

    @staticmethod
    def forward(ctx, image):
        if image.is_cuda and image.dtype == torch.float32 and not os.environ.get('UPF_NO_FUSED_GREY'):
            from .. import ops
            if hasattr(ops, 'grey'):
                return ops.grey(image)                # (the same expression, one launch: csrc/loss.hip grey_kernel)
        r, g, b = image[:, 0:1], image[:, 1:2], image[:, 2:3]
        return 0.2989 * r + 0.5870 * g + 0.1140 * b

    @staticmethod
    def backward(ctx, gy):
        return gy * _grey_weights(gy)


_VALID = {}
_ZEROS = {}
_GREY_W = {}


def _grey_weights(like):
    """[1,3,1,1] constant on the device (made once, outside any graph capture: a host-to-device copy cannot be captured)."""
    key = (like.dtype, like.device)
    w = _GREY_W.get(key)
    if w is None:
        w = _GREY_W[key] = torch.tensor([0.2989, 0.5870, 0.1140], dtype=like.dtype, device=like.device).view(1, 3, 1, 1)
    return w


def cached_constants():
    """Every constant tensor of this module's caches: a trainer that captured a hipGraph keeps them alive (their addresses are
    baked into the graph; the caches evict when they grow past 64 shapes — ADVICE r3)."""
    return list(_GREY_W.values()) + list(_VALID.values()) + list(_ZEROS.values())


def _census_valid(mask, max_distance):
    """ones inside, zeros in the max_distance-pixel border (utils/loss.py:58-60): a constant of the shape, built once."""
    key = (tuple(mask.shape), mask.dtype, mask.device, max_distance)
    v = _VALID.get(key)
    if v is None:
        if len(_VALID) > 64:
            _VALID.clear()       # (a captured trainer holds its own references: cached_constants(), train.Trainer._capture)
        inner = torch.ones(mask.shape[0], mask.shape[1], mask.shape[2] - 2 * max_distance, mask.shape[3] - 2 * max_distance,
                           dtype=mask.dtype, device=mask.device)
        v = _VALID[key] = F.pad(inner, [max_distance] * 4)
    return v


def _zeros_like_cached(t):
    key = (tuple(t.shape), t.dtype, t.device)
    z = _ZEROS.get(key)
    if z is None:
        if len(_ZEROS) > 64:
            _ZEROS.clear()       # (see _census_valid)
        z = _ZEROS[key] = torch.zeros_like(t)
    return z


class loss_functions():

    @classmethod
    def photo_loss_function(cls, diff, mask, q, charbonnier_or_abs_robust, if_use_occ, averge=True):
        """utils/loss.py:16-48."""
        red = (lambda t: t.mean()) if averge else (lambda t: t.sum())
        if charbonnier_or_abs_robust:
            if if_use_occ:
                p = (diff ** 2 + 1e-6).pow(q) * mask
                return red(p) / (red(mask) * 2 + 1e-6)
            return red((diff ** 2 + 1e-8).pow(q))
        d = (diff.abs() + 0.01).pow(q)
        if if_use_occ:
            return torch.sum(d * mask) / (torch.sum(mask) * 2 + 1e-6)
        return red(d)

    @classmethod
    def census_loss_torch(cls, img1, img1_warp, mask, q, charbonnier_or_abs_robust, if_use_occ, averge=True, max_distance=3):
        """Soft census (ternary) transform distance, utils/loss.py:50-91."""
        # ONE fused HIP launch (csrc/misc.hip: upf_census_forward / _backward) instead of the reference's two 49-channel
        # identity convolutions and ~10 element-wise passes over [B,49,H,W] tensors (utils/loss.py:52-67); same fp32
        # arithmetic, deterministic gather backward.  GPU only, like every operator of this package (oracle/ops.py keeps
        # the reference's spelling as the test oracle).
        from .. import ops

        def grey(image):
            return _GreyFunction.apply(image.float())
        dist = ops.census_distance(grey(img1), grey(img1_warp), max_distance).to(img1.dtype)
        valid = _census_valid(mask, max_distance)
        if (not charbonnier_or_abs_robust) and if_use_occ and dist.is_cuda and dist.dtype == torch.float32:
            # sum((|d| + 0.01)^q * m) / (sum(m) * 2 + 1e-6), utils/loss.py:28-31, as the one-launch reduction of csrc/loss.hip
            # (ops.robust_loss_sums with y = 0) instead of nine element-wise / reduction launches each way
            s, s_m = ops.robust_loss_sums(dist, _zeros_like_cached(dist), mask * valid, q=q, eps=0.01)
            return s / (s_m * 2 + 1e-6)
        if (not charbonnier_or_abs_robust) and (not if_use_occ) and dist.is_cuda and dist.dtype == torch.float32:
            # mean((|d| + 0.01)^q), utils/loss.py:32-33 — the same one-launch deterministic reduction without a mask.  (Round 4: ATen's
            # multi-block `mean()` zeroes its semaphores with a memset node, the node class that was seen mis-ordered inside
            # replayed hipGraphs on this ROCm (api.hip: zero_fill_u64_kernel): inside a captured training step whose allocation
            # pattern had shifted it returned 9.8e3 for inputs whose mean is 2.0 — tools/frozen_graph_probe.py.)
            s, _ = ops.robust_loss_sums(dist, _zeros_like_cached(dist), None, q=q, eps=0.01)
            return s / dist.numel() if averge else s
        return cls.photo_loss_function(diff=dist, mask=mask * valid, q=q, charbonnier_or_abs_robust=charbonnier_or_abs_robust,
                                       if_use_occ=if_use_occ, averge=averge)

    @classmethod
    def flow_smooth_delta(cls, flow, if_second_order=False):
        """utils/loss.py:93-111."""
        def grad(x):
            return x[:, :, :, 1:] - x[:, :, :, :-1], x[:, :, 1:] - x[:, :, :-1]
        dx, dy = grad(flow)
        loss = dx.abs().mean() + dy.abs().mean()
        if if_second_order:
            dx2, dxdy = grad(dx)
            dydx, dy2 = grad(dy)
            loss = loss + dx2.abs().mean() + dxdy.abs().mean() + dydx.abs().mean() + dy2.abs().mean()
        return loss

    @classmethod
    def edge_aware_smoothness_per_pixel(cls, img, pred):
        """utils/loss.py:113-134."""
        def dx(t):
            return t[:, :, :-1, :] - t[:, :, 1:, :]

        def dy(t):
            return t[:, :, :, :-1] - t[:, :, :, 1:]
        wx = torch.exp(-dx(img).abs().mean(1, keepdim=True))
        wy = torch.exp(-dy(img).abs().mean(1, keepdim=True))
        return (dx(pred).abs() * wx).mean() + (dy(pred).abs() * wy).mean()
