"""Training-loss helpers mirroring `/root/reference/utils/loss.py` (class `loss_functions`).
Plain torch ops (loss side, not the hot path: SURVEY.md §2 row 14)."""
import torch
import torch.nn.functional as F


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
            r, g, b = torch.split(image.float(), 1, 1)
            return 0.2989 * r + 0.5870 * g + 0.1140 * b
        dist = ops.census_distance(grey(img1), grey(img1_warp), max_distance).to(img1.dtype)
        inner = torch.ones(mask.shape[0], mask.shape[1], mask.shape[2] - 2 * max_distance, mask.shape[3] - 2 * max_distance,
                           dtype=mask.dtype, device=mask.device)
        valid = F.pad(inner, [max_distance] * 4)
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
