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
        patch = 2 * max_distance + 1
        n = patch * patch

        def ternary(image):
            r, g, b = torch.split(image, 1, 1)
            gray = 0.2989 * r + 0.5870 * g + 0.1140 * b
            weight = torch.eye(n, dtype=gray.dtype, device=gray.device).view(n, 1, patch, patch)
            t = torch.conv2d(gray, weight, bias=None, stride=[1, 1], padding=[max_distance, max_distance]) - gray
            return t / torch.sqrt(0.81 + t ** 2)

        def hamming(t1, t2):
            d = (t1 - t2) ** 2
            return torch.sum(d / (0.1 + d), 1, keepdim=True)

        dist = hamming(ternary(img1), ternary(img1_warp))
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
