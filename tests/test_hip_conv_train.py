"""The convolution under autograd on the matrix cores (ops.ConvTrainFunction: forward + data gradient on csrc/conv3x3.hip,
weight gradient on csrc/conv_wgrad.hip, LeakyReLU / bias gradients fused or one launch each) against fp32 torch autograd
evaluated on the SAME 16-bit-rounded operands — every layer geometry of the decoder (model/pwc_modules.py:250-286, :396-412):
Cin 115..565, Cout 2..128, dilation 1/2/4/8/16, 1x1, with and without activation."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # (B, Cin, Cout, H, W, k, dilation, slope)
    (2, 115, 128, 16, 32, 3, 1, 0.1), (1, 243, 128, 12, 40, 3, 1, 0.1), (2, 565, 128, 8, 16, 3, 1, 0.1), (2, 563, 2, 16, 24, 3, 1, 0.0),
    (1, 128, 128, 24, 40, 3, 2, 0.1), (1, 128, 128, 20, 48, 3, 4, 0.1), (1, 128, 96, 26, 40, 3, 8, 0.1), (1, 96, 64, 40, 64, 3, 16, 0.1),
    (2, 64, 32, 9, 16, 3, 1, 0.1), (2, 32, 2, 7, 8, 3, 1, 0.0), (2, 196, 32, 6, 24, 1, 1, 0.1), (4, 32, 32, 64, 208, 1, 1, 0.1),
    (4, 64, 32, 32, 104, 3, 1, 0.1), (1, 184, 3, 10, 32, 3, 1, 0.0), (3, 5, 7, 5, 8, 3, 1, 0.1),
    # ragged widths (the coarse levels of the 256x832 training crops: 52, 26, 13 pixels), odd sizes, every dilation
    (2, 115, 128, 16, 52, 3, 1, 0.1), (2, 243, 128, 8, 26, 3, 1, 0.1), (2, 196, 32, 4, 13, 1, 1, 0.1), (1, 565, 128, 4, 13, 3, 1, 0.1),
    (1, 128, 128, 16, 52, 3, 2, 0.1), (1, 128, 128, 9, 26, 3, 4, 0.1), (1, 128, 96, 17, 52, 3, 8, 0.1), (1, 96, 64, 33, 52, 3, 16, 0.1),
    (3, 7, 5, 5, 9, 3, 1, 0.1), (1, 33, 31, 7, 37, 3, 1, 0.0),
]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_conv_train_matches_fp32_autograd(case, dtype):
    from upflow_pytorch_amd import ops
    B, Cin, Cout, H, W, k, d, slope = case
    g = torch.Generator().manual_seed(sum(case[:7]))
    x = torch.randn(B, Cin, H, W, generator=g).to(dtype).cuda().requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).cuda().requires_grad_(True)
    b = (torch.randn(Cout, generator=g) * 0.1).cuda().requires_grad_(True)
    gy = torch.randn(B, Cout, H, W, generator=g).to(dtype).cuda()
    assert ops.conv_train_supported(x, w, 1, d) and ops.conv_wgrad_supported(x, w, 1, d)
    y = ops.conv_train(x, w, b, d, slope)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), gy)
    # fp32 reference on the operands the kernels saw (x, w rounded to 16 bits)
    xr = x.detach().float()
    wr = w.detach().to(dtype).float()
    pre = F.conv2d(xr, wr, b.detach(), padding=d * (k - 1) // 2, dilation=d)
    yr = F.leaky_relu(pre, slope) if slope else pre
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert (y.detach().float() - yr).abs().max() <= 2 * eps * max(1.0, float(yr.abs().max()))
    # gradient through the activation with the kernel's own mask (sign of its 16-bit output), rounded like the kernel does
    gpre = (gy.float() * torch.where(y.detach().float() > 0, 1.0, slope)).to(dtype).float() if slope else gy.float()
    gx_ref = torch.nn.grad.conv2d_input(xr.shape, wr, gpre, padding=d * (k - 1) // 2, dilation=d)
    gw_ref = torch.nn.grad.conv2d_weight(xr, wr.shape, gpre, padding=d * (k - 1) // 2, dilation=d)
    gb_ref = gpre.sum((0, 2, 3))
    assert (gx.float() - gx_ref).abs().max() <= 2 * eps * max(1.0, float(gx_ref.abs().max()))
    assert gw.dtype == torch.float32 and gw.shape == w.shape
    assert (gw - gw_ref).abs().max() <= 2e-4 * max(1.0, float(gw_ref.abs().max())), float((gw - gw_ref).abs().max())
    assert (gb - gb_ref).abs().max() <= 1e-4 * max(1.0, float(gb_ref.abs().max()))
    # deterministic
    gx2, gw2, gb2 = torch.autograd.grad(ops.conv_train(x, w, b, d, slope), (x, w, b), gy)
    assert torch.equal(gw, gw2) and torch.equal(gx, gx2) and torch.equal(gb, gb2)


@pytest.mark.parametrize('case', [(2, 16, 32, 16, 52, 3, 1, 0.1, 1), (2, 96, 64, 8, 26, 3, 1, 0.1, 1), (1, 196, 32, 4, 13, 1, 1, 0.1, 1),
                                  (2, 3, 16, 32, 64, 3, 1, 0.1, 2), (2, 16, 32, 17, 30, 3, 1, 0.1, 2), (1, 32, 32, 9, 21, 3, 1, 0.0, 1)])
def test_conv_train_ragged_and_strided_layers(case):
    """Ragged-width levels (weight gradient through PyTorch-ROCm) and the stride-2 layers of the pyramid / SGU guidance
    (forward on the MFMA kernel, gradients through PyTorch-ROCm): same check as above."""
    from upflow_pytorch_amd import ops
    B, Cin, Cout, H, W, k, d, slope, stride = case
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(sum(case[:7]))
    x = torch.randn(B, Cin, H, W, generator=g).to(dtype).cuda().requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).cuda().requires_grad_(True)
    b = (torch.randn(Cout, generator=g) * 0.1).cuda().requires_grad_(True)
    assert ops.conv_train_supported(x, w, stride, d)
    y = ops.conv_train(x, w, b, d, slope, stride)
    gy = torch.randn(y.shape, generator=g).to(dtype).cuda()
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), gy)
    xr, wr = x.detach().float(), w.detach().to(dtype).float()
    pad = d * (k - 1) // 2
    pre = F.conv2d(xr, wr, b.detach(), stride=stride, padding=pad, dilation=d)
    yr = F.leaky_relu(pre, slope) if slope else pre
    eps = 2.0 ** -8
    assert y.shape == yr.shape and (y.detach().float() - yr).abs().max() <= 2 * eps * max(1.0, float(yr.abs().max()))
    gpre = (gy.float() * torch.where(y.detach().float() > 0, 1.0, slope)).to(dtype).float() if slope else gy.float()
    gx_ref = torch.nn.grad.conv2d_input(xr.shape, wr, gpre, stride=stride, padding=pad, dilation=d)
    gw_ref = torch.nn.grad.conv2d_weight(xr, wr.shape, gpre, stride=stride, padding=pad, dilation=d)
    assert (gx.float() - gx_ref).abs().max() <= 3 * eps * max(1.0, float(gx_ref.abs().max()))
    assert (gw - gw_ref).abs().max() <= 2e-2 * max(1.0, float(gw_ref.abs().max()))      # (fp32 master weights vs their 16-bit rounding in the dgrad)
    assert (gb - gpre.sum((0, 2, 3))).abs().max() <= 1e-4 * max(1.0, float(gpre.sum((0, 2, 3)).abs().max()))


def test_conv_train_unsupported_shapes_are_reported():
    from upflow_pytorch_amd import ops
    x = torch.zeros(1, 8, 8, 5, dtype=torch.bfloat16, device='cuda')           # rows shorter than 8 pixels
    w = torch.zeros(4, 8, 3, 3, device='cuda')
    assert not ops.conv_train_supported(x, w, 1, 1)
    x = torch.zeros(1, 8, 8, 16, dtype=torch.bfloat16, device='cuda')
    assert not ops.conv_train_supported(x.float(), w, 1, 1)
    assert ops.conv_train_supported(x, w, 1, 1) and ops.conv_train_supported(x, w, 2, 1) and not ops.conv_train_supported(x, w, 2, 2)
    assert ops.conv_wgrad_supported(x, w, 1, 1) and not ops.conv_wgrad_supported(x, w, 2, 1) and not ops.conv_wgrad_supported(x, w, 1, 3)
    assert ops.conv_wgrad_supported(torch.zeros(1, 8, 8, 12, dtype=torch.bfloat16, device='cuda'), w, 1, 1)
