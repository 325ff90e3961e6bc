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
    """Ragged-width levels (the ragged launch of the multi-level weight gradient) and the stride-2 layers of the pyramid / SGU
    guidance (gradients through their space-to-depth form on the stride-1 kernels): same check as above."""
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


# ---- the fused activation-gradient pass, the multi-level weight gradient, the in-buffer dense stack, the parameter sinks ----
@pytest.mark.parametrize('shape', [(2, 32, 16, 52), (1, 7, 5, 9), (3, 64, 8, 32), (2, 2, 4, 13)])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_act_grad_matches_torch(shape, dtype):
    """dst = (src + add) * (y > 0 ? 1 : slope) on channel slices of wider buffers + the bias partial sums of the rounded
    result (upf_act_grad / upf_conv_bias_grad_finish) vs the same arithmetic in torch."""
    from upflow_pytorch_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    wide = lambda: torch.randn(B, C + 5, H, W, generator=g).to(dtype).cuda()
    src, add, y = wide()[:, 2:2 + C], wide()[:, 1:1 + C], wide()[:, 5:]
    for use_add in (False, True):
        for use_y in (False, True):
            dst, part = ops.act_grad(src, y if use_y else None, 0.1, add=add if use_add else None, want_bias=True)
            ref = src.float()
            if use_add:
                ref = (ref + add.float()).to(dtype).float()
            if use_y:
                ref = torch.where(y.float() > 0, ref, ref * 0.1)
            ref = ref.to(dtype)
            assert torch.equal(dst, ref)
            gb = ops.conv_bias_grad_finish([part], C)
            gb_ref = ref.double().sum((0, 2, 3))
            assert (gb.double() - gb_ref).abs().max() <= 1e-4 * max(1.0, float(gb_ref.abs().max()))
    # in place, and the sums alone
    buf = torch.cat([src, src], 1).contiguous()
    want, _ = ops.act_grad(src, y, 0.1)
    ops.act_grad(buf[:, C:], y, 0.1, dst=buf[:, C:])
    assert torch.equal(buf[:, C:], want) and torch.equal(buf[:, :C], src)
    none, part = ops.act_grad(src, dst=False, want_bias=True)
    assert none is None and (ops.conv_bias_grad_finish([part, part], C) - 2 * src.float().sum((0, 2, 3))).abs().max() <= 1e-3 * max(1.0, float(src.float().abs().sum((0, 2, 3)).max()))


@pytest.mark.parametrize('cfg', [(115, 128, 3, 1), (128, 96, 3, 8), (196, 32, 1, 1), (40, 70, 3, 2)])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_wgrad_multi_level(cfg, dtype):
    """One weight gradient over five uses of different sizes (aligned and ragged pyramid levels, channel slices of wider
    buffers) == the sum of the per-use fp32 references; deterministic."""
    from upflow_pytorch_amd import ops
    Cin, Cout, k, d = cfg
    g = torch.Generator().manual_seed(Cin + Cout + d)
    sizes = [(2, 32, 104), (2, 16, 52), (2, 8, 26), (2, 4, 13), (2, 64, 208), (1, 24, 40), (1, 9, 16)]      # seven uses: two launches of <= 6
    uses, ref = [], 0
    for (B, H, W) in sizes:
        xw = torch.randn(B, Cin + 3, H, W, generator=g).to(dtype).cuda()
        gw_ = (torch.randn(B, Cout + 2, H, W, generator=g) * 0.25).to(dtype).cuda()
        x, gy = xw[:, 3:], gw_[:, :Cout]
        uses.append((x, gy))
        ref = ref + torch.nn.grad.conv2d_weight(x.float(), (Cout, Cin, k, k), gy.float(), padding=d * (k - 1) // 2, dilation=d)
    got = ops.conv_wgrad_multi(uses, Cin, Cout, k, d)
    assert got.shape == ref.shape and (got - ref).abs().max() <= 2e-4 * max(1.0, float(ref.abs().max())), float((got - ref).abs().max())
    assert torch.equal(got, ops.conv_wgrad_multi(uses, Cin, Cout, k, d))
    one = ops.conv_wgrad_multi(uses[:1], Cin, Cout, k, d)
    ref1 = torch.nn.grad.conv2d_weight(uses[0][0].float(), (Cout, Cin, k, k), uses[0][1].float(), padding=d * (k - 1) // 2, dilation=d)
    assert (one - ref1).abs().max() <= 2e-4 * max(1.0, float(ref1.abs().max()))


@pytest.mark.parametrize('cfg', [(3, 16, 3, 1, 4), (12, 16, 3, 1, 8), (16, 16, 3, 2, 4), (16, 32, 3, 1, 8), (32, 32, 1, 1, 2), (16, 3, 3, 1, 4), (16, 16, 3, 1, 6)])
def test_wgrad_of_narrow_layers(cfg):
    """Layers with <= 32 input and output channels (the feature pyramid's first levels, the heads): aligned and ragged widths,
    channel slices of wider buffers, several batch sizes == the per-image fp32 reference; deterministic."""
    from upflow_pytorch_amd import ops
    Cin, Cout, k, d, B = cfg
    g = torch.Generator().manual_seed(Cin * 100 + Cout + B)
    for sizes in ([(64, 104)], [(24, 40), (12, 26), (6, 13)]):
        uses, ref = [], 0
        for (H, W) in sizes:
            xw = torch.randn(B, Cin + 5, H, W, generator=g).bfloat16().cuda()
            gw_ = (torch.randn(B, Cout + 2, H, W, generator=g) * 0.25).bfloat16().cuda()
            x, gy = xw[:, 2:2 + Cin], gw_[:, 1:1 + Cout]
            uses.append((x, gy))
            ref = ref + torch.nn.grad.conv2d_weight(x.float(), (Cout, Cin, k, k), gy.float(), padding=d * (k - 1) // 2, dilation=d)
        got = ops.conv_wgrad_multi(uses, Cin, Cout, k, d)
        assert (got - ref).abs().max() <= 2e-4 * max(1.0, float(ref.abs().max())), (sizes, float((got - ref).abs().max()))
        assert torch.equal(got, ops.conv_wgrad_multi(uses, Cin, Cout, k, d))


@pytest.mark.parametrize('cfg', [(115, 128, 3, 1), (563, 2, 3, 1), (196, 300, 1, 1), (32, 32, 3, 4)])
def test_wgrad_multi_with_fused_bias_finish(cfg):
    """upf_conv_wgrad_multi_bias: the reduction launch also finishes the bias gradient — the weight gradient is bit-identical to
    the launch without it, the bias gradient bit-identical to upf_conv_bias_grad_finish over the same first-stage buffers."""
    from upflow_pytorch_amd import ops
    Cin, Cout, k, d = cfg
    g = torch.Generator().manual_seed(7 + Cin + Cout)
    uses, parts = [], []
    for (B, H, W) in [(2, 32, 104), (2, 16, 52), (2, 8, 26), (2, 4, 13)]:
        x = torch.randn(B, Cin, H, W, generator=g).bfloat16().cuda()
        gy = (torch.randn(B, Cout, H, W, generator=g) * 0.25).bfloat16().cuda()
        uses.append((x, gy))
        parts.append(ops.act_grad(gy, dst=False, want_bias=True)[1])
    gw, gb = ops.conv_wgrad_multi(uses, Cin, Cout, k, d, bias_parts=parts)
    assert torch.equal(gw, ops.conv_wgrad_multi(uses, Cin, Cout, k, d))
    assert torch.equal(gb, ops.conv_bias_grad_finish(parts, Cout))
    ref = sum(gy.float().sum((0, 2, 3)) for _, gy in uses)
    assert (gb - ref).abs().max() <= 1e-3 * max(1.0, float(ref.abs().max()))
    with pytest.raises(ops.UpflowHipError):
        ops.conv_wgrad_multi(uses, Cin, Cout, k, d, bias_parts=parts * 3)


def _dense_stack(ch_in, dev):
    from upflow_pytorch_amd.model.pwc_modules import FlowEstimatorDense_v2
    torch.manual_seed(ch_in)
    m = FlowEstimatorDense_v2(ch_in).to(dev)
    for p in m.parameters():
        if p.dim() == 1:
            torch.nn.init.normal_(p, std=0.05)
    return m


# (B, Cin, Cout, H, W): one 16-channel chunk (wide epilogue), two / four / one channel blocks per workgroup, 16-row tiles,
# split-K, ragged rows (even and odd widths) in the tiled and the split-K kernel, input and outputs as odd channel slices
GATED = [(2, 2, 32, 16, 64), (2, 34, 64, 16, 32), (8, 66, 96, 64, 160), (1, 40, 32, 24, 64), (4, 40, 32, 128, 256), (1, 290, 128, 64, 64),
         (2, 130, 96, 16, 52), (2, 226, 64, 8, 13), (2, 20, 32, 12, 26), (2, 450, 128, 4, 13), (1, 7, 3, 9, 40)]


@pytest.mark.parametrize('case', GATED)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_gated_convolution_equals_convolution_then_act_grad(case, dtype):
    """upf_conv_forward_gated (the mask / residual arithmetic of upf_act_grad in the convolution's epilogue) is BIT-identical to
    upf_conv_forward followed by upf_act_grad in place, for every kernel variant the data-gradient convolutions take — with and
    without the addend, and with every operand a channel slice of a wider buffer."""
    from upflow_pytorch_amd import ops
    B, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    rnd = lambda *s: torch.randn(*s, generator=g)
    x = rnd(B, Cin + 3, H, W).to(dtype).cuda()[:, 1:1 + Cin]
    w = (rnd(Cout, Cin, 3, 3) / (3.0 * Cin ** 0.5)).to(dtype).cuda()
    packed = ops.conv3x3_pack(w)
    bias = (rnd(Cout) * 0.1).cuda()
    act = rnd(B, Cout + 2, H, W).to(dtype).cuda()[:, 2:]
    act[:, :, ::3, ::5] = 0                                              # exact zeros take the slope
    add = rnd(B, Cout + 4, H, W).to(dtype).cuda()[:, 3:3 + Cout]
    for use_add in (False, True):
        a = add if use_add else None
        two = torch.full((B, Cout + 2, H, W), 7.0, dtype=dtype, device='cuda')
        ops.conv3x3_forward_raw(x, packed, bias, two[:, 1:1 + Cout], 1, 0.0, 1, 3)
        ops.act_grad(two[:, 1:1 + Cout], act, 0.1, add=a, dst=two[:, 1:1 + Cout])
        one = torch.full((B, Cout + 2, H, W), 7.0, dtype=dtype, device='cuda')
        ops.conv3x3_forward_gated_raw(x, packed, bias, one[:, 1:1 + Cout], a, act, 0.1)
        assert torch.equal(one, two), (case, use_add, float((one.float() - two.float()).abs().max()))
    # the addend alone (no mask) = the 16-bit tensor add
    ops.conv3x3_forward_raw(x, packed, bias, two[:, 1:1 + Cout], 1, 0.0, 1, 3)
    ops.conv3x3_forward_gated_raw(x, packed, bias, one[:, 1:1 + Cout], add, None, 0.1)
    assert torch.equal(one[:, 1:1 + Cout], two[:, 1:1 + Cout] + add)
    with pytest.raises(ops.UpflowHipError):
        ops.conv3x3_forward_gated_raw(x, packed, bias, one[:, 1:1 + Cout], None, act[:, :, :, :W - 1], 0.1)
    with pytest.raises(ops.UpflowHipError):
        ops.conv3x3_forward_gated_raw(x, packed, bias, one[:, 1:1 + Cout], None, None, 0.1)


@pytest.mark.parametrize('geom', [(2, 16, 24), (2, 8, 26), (1, 4, 13), (2, 32, 104)])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_dense_stack_in_buffer_matches_layerwise_autograd(geom, dtype):
    """ops.DenseStackTrainFunction (one buffer, stacked data-gradient convolutions, no concatenations) vs the layer-by-layer
    autograd schedule (ConvTrainFunction + torch.cat) and vs fp32 torch autograd of the reference's module structure
    (model/pwc_modules.py:250-286) on the same rounded inputs."""
    from upflow_pytorch_amd import ops
    B, H, W = geom
    m = _dense_stack(115, 'cuda')
    g = torch.Generator().manual_seed(sum(geom))
    mk = lambda c, dt: torch.randn(B, c, H, W, generator=g).to(dt).cuda().requires_grad_(True)
    c, A, flow = mk(81, dtype), mk(32, dtype), mk(2, torch.float32)
    g_buf = (torch.randn(B, m._n_total + 2, H, W, generator=g) * 0.1).to(dtype).cuda()
    g_out = torch.randn(B, 2, H, W, generator=g).to(dtype).cuda()
    params = list(m.parameters())

    assert m.train_in_buffer_ok([c, A, flow])
    buf, out = m.forward_train([c, A, flow], flow_tail=flow)
    assert buf.shape == (B, m._n_total + 2, H, W) and out.shape == (B, 2, H, W)
    grads = torch.autograd.grad((buf, out), [c, A, flow] + params, (g_buf, g_out))

    # the same stack, layer by layer (16-bit tensor adds between the layers)
    m._no_train_buffer = True
    x = torch.cat([c, A, flow.to(dtype)], 1)
    x5, out2 = m(x)
    tail = (flow + out2.float()).to(dtype)
    buf2 = torch.cat([x5, tail], 1)
    assert torch.equal(out, out2) and torch.equal(buf, buf2)                     # forward: the same kernels on the same values
    grads2 = torch.autograd.grad((buf2, out2), [c, A, flow] + params, (g_buf, g_out))
    m._no_train_buffer = False

    # fp32 reference of the module structure
    import torch.nn.functional as F
    xr = torch.cat([c.detach().float(), A.detach().float(), flow.detach().to(dtype).float()], 1).requires_grad_(True)
    pr = [p.detach().clone().requires_grad_(True) for p in params]
    h = xr
    for i in range(5):
        h = torch.cat([F.leaky_relu(F.conv2d(h, pr[2 * i].to(dtype).float(), pr[2 * i + 1], padding=1), 0.1), h], 1)
    outr = F.conv2d(h, pr[10].to(dtype).float(), pr[11], padding=1)
    flow_r = flow.detach().clone().requires_grad_(True)
    bufr = torch.cat([h, flow_r + outr], 1)
    gr = torch.autograd.grad((bufr, outr), [xr, flow_r] + pr, (g_buf.float(), g_out.float()))
    gx_ref = gr[0]
    refs = [gx_ref[:, :81], gx_ref[:, 81:113], gx_ref[:, 113:] + gr[1]] + list(gr[2:])
    rel = lambda u, v: float((u.float() - v).norm() / v.norm().clamp_min(1e-20))
    lim = 0.06 if dtype == torch.bfloat16 else 0.008          # six chained layers with 16-bit activations vs fp32
    for i, (a, b, r) in enumerate(zip(grads, grads2, refs)):
        assert a.shape == r.shape and a.dtype == ([c, A, flow] + params)[i].dtype
        assert rel(a, r) <= lim, (i, rel(a, r), rel(b, r))
        # the in-buffer schedule is at least as close to fp32 as the layer-wise one (fp32 accumulation across consumers)
        assert rel(a, r) <= 1.1 * rel(b, r) + 1e-4, (i, rel(a, r), rel(b, r))
    again = torch.autograd.grad(m.forward_train([c, A, flow], flow_tail=flow), [c, A, flow] + params, (g_buf, g_out))
    assert all(torch.equal(a, b) for a, b in zip(grads, again))
    # the data-gradient operands packed one by one (concatenation + pack per slice) instead of in one launch: the same bits
    ops.DenseStackTrainFunction.no_stack_pack = True
    ops.train_caches_clear()
    try:
        one_by_one = torch.autograd.grad(m.forward_train([c, A, flow], flow_tail=flow), [c, A, flow] + params, (g_buf, g_out))
    finally:
        ops.DenseStackTrainFunction.no_stack_pack = False
        ops.train_caches_clear()
    assert all(torch.equal(a, b) for a, b in zip(grads, one_by_one))
    # the schedule with the separate mask / residual passes (round 4): the same data and weight gradients bit for bit; the bias
    # gradients are the same sums in another order
    ops.DenseStackTrainFunction.no_gated_dgrad = True
    try:
        sep = torch.autograd.grad(m.forward_train([c, A, flow], flow_tail=flow), [c, A, flow] + params, (g_buf, g_out))
    finally:
        ops.DenseStackTrainFunction.no_gated_dgrad = False
    for i, (a, b) in enumerate(zip(grads, sep)):
        if i >= 3 and a.dim() == 1:
            assert (a - b).abs().max() <= 1e-4 * max(1.0, float(b.abs().max())), i
        else:
            assert torch.equal(a, b), i


def test_shared_conv_grads_defers_to_one_contraction():
    """Inside ops.shared_conv_grads the five uses of a shared decoder (pyramid levels of different sizes) hand their
    (x, g) pairs to the parameters' sinks; the gradients equal the per-use ones summed by autograd (fp32 summation order
    aside), inputs' gradients are identical, and nothing is left in the registry afterwards."""
    from upflow_pytorch_amd import ops
    from upflow_pytorch_amd.model.pwc_modules import ContextNetwork_v2_
    dt = torch.bfloat16
    est = _dense_stack(115, 'cuda')
    ctxn = ContextNetwork_v2_(est._n_total + 2).cuda()
    sizes = [(2, 4, 13), (2, 8, 26), (2, 16, 52), (2, 32, 104), (2, 64, 208)]
    g = torch.Generator().manual_seed(3)
    data = [[torch.randn(B, ch, H, W, generator=g).to(d).cuda().requires_grad_(True) for ch, d in ((81, dt), (32, dt), (2, torch.float32))] for B, H, W in sizes]
    params = list(est.parameters()) + list(ctxn.parameters())

    def run(shared):
        convs = [m for m in list(est.modules()) + list(ctxn.modules()) if isinstance(m, torch.nn.Conv2d)]
        total = 0
        with ops.shared_conv_grads(convs if shared else []):
            for c, A, flow in data:
                buf, res = est.forward_train([c, A, flow], flow_tail=flow)
                fine = ctxn(buf)
                total = total + (res.float() + fine.float()).square().mean()
        assert not ops._GATES
        flat = [t for lvl in data for t in lvl]
        return torch.autograd.grad(total, flat + params)

    a, b = run(True), run(False)
    n_in = 3 * len(sizes)
    for x, y in zip(a[:n_in], b[:n_in]):
        assert torch.equal(x, y)
    for x, y in zip(a[n_in:], b[n_in:]):
        assert (x - y).abs().max() <= 1e-5 * max(1.0, float(y.abs().max())), float((x - y).abs().max())
    a2 = run(True)
    assert all(torch.equal(x, y) for x, y in zip(a, a2))


@pytest.mark.parametrize('shape', [(2, 16, 32, 64), (1, 3, 6, 20), (2, 5, 12, 26), (1, 32, 64, 208)])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_space_to_depth_and_stride2_gradients(shape, dtype):
    """upf_space_to_depth2 == F.pixel_unshuffle / F.pixel_shuffle (bit copies), and the stride-2 layer's gradients through its
    space-to-depth form (upf_conv_wgrad_s2d, upf_conv_pack_weights_f32(dgrad = 2)) vs fp32 autograd of the strided convolution
    on the same rounded operands (model/pwc_modules.py:95, model/upflow.py:53-55)."""
    from upflow_pytorch_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(dtype).cuda()
    xs = ops.space_to_depth2(x)
    assert torch.equal(xs, F.pixel_unshuffle(x, 2))
    assert torch.equal(ops.space_to_depth2(xs, inverse=True), x)
    if W // 2 < 8:
        return
    Cout = 24
    w = (torch.randn(Cout, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda().requires_grad_(True)
    b = (torch.randn(Cout, generator=g) * 0.1).cuda().requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    gy = torch.randn(B, Cout, H // 2, W // 2, generator=g).to(dtype).cuda()
    assert ops._s2d_ok(xg, w, 2, 1)
    y = ops.conv_train(xg, w, b, 1, 0.1, 2)
    gx, gw, gb = torch.autograd.grad(y, (xg, w, b), gy)
    xr, wr = x.float(), w.detach().to(dtype).float()
    gpre = (gy.float() * torch.where(y.detach().float() > 0, 1.0, 0.1)).to(dtype).float()
    gx_ref = torch.nn.grad.conv2d_input(xr.shape, wr, gpre, stride=2, padding=1)
    gw_ref = torch.nn.grad.conv2d_weight(xr, wr.shape, gpre, stride=2, padding=1)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert (gx.float() - gx_ref).abs().max() <= 2 * eps * max(1.0, float(gx_ref.abs().max()))
    assert gw.shape == w.shape and (gw - gw_ref).abs().max() <= 2e-4 * max(1.0, float(gw_ref.abs().max())), float((gw - gw_ref).abs().max())
    assert (gb - gpre.sum((0, 2, 3))).abs().max() <= 1e-4 * max(1.0, float(gpre.sum((0, 2, 3)).abs().max()))


def test_stacked_dgrad_pack_equals_concatenate_then_pack():
    """upf_conv_pack_stacked_dgrad (every data-gradient operand of a dense stack in one launch, the concatenation as an index
    computation) writes bit for bit what torch.cat + upf_conv_pack_weights_f32(dgrad = 1) wrote per slice."""
    from upflow_pytorch_amd import ops
    g = torch.Generator().manual_seed(5)
    cin0, f = 21, [40, 33, 24, 16, 8]
    nt = cin0 + sum(f)
    lo = [sum(f[k + 1:]) for k in range(5)]
    hi = {k: lo[k] + f[k] for k in range(5)}
    hi[5] = 0
    w = {k: torch.randn(f[k], nt - hi[k], 3, 3, generator=g).cuda() for k in range(5)}
    w[5] = torch.randn(3, nt, 3, 3, generator=g).cuda()
    order = [5, 4, 3, 2, 1, 0]
    for dtype in (torch.bfloat16, torch.float16):
        slices = [(lo[k], f[k], pos) for pos, k in enumerate(order[1:], start=1)] + [(nt - cin0, cin0, 6)]
        got = ops._stack_dgrad_packs([w[j] for j in order], [hi[j] for j in order], slices, dtype)
        for (c0, width, npos), t in zip(slices, got):
            want = ops._stacked_dgrad_pack([w[j] for j in order[:npos]], lo, [hi[j] for j in order[:npos]], c0, width, dtype)
            assert torch.equal(t[:want.numel()], want), (c0, width, npos)
        assert ops._stack_dgrad_packs([w[j] for j in order], [hi[j] for j in order], slices, dtype) is got      # cached
    ops.train_caches_clear()
    with pytest.raises(ops.UpflowHipError):                                     # a slice outside a consumer's input
        ops._stack_dgrad_packs([w[5], w[4]], [0, hi[4]], [(lo[4], f[4], 2)], torch.bfloat16)
    ops.train_caches_clear()


def test_pack_weights_multi_equals_the_single_packs():
    """upf_conv_pack_weights_f32_multi (every layer's operands of a training step in one launch) writes bit for bit what
    upf_conv_pack_weights_f32 writes per layer: forward, data-gradient and space-to-depth data-gradient forms, 3x3 and 1x1,
    more jobs than one kernel-argument table holds (56)."""
    import ctypes
    from upflow_pytorch_amd import _lib, ops
    g = torch.Generator().manual_seed(11)
    shapes = [(16, 3, 3), (16, 16, 3), (32, 16, 3), (128, 115, 3), (2, 563, 3), (32, 196, 1), (96, 371, 3), (128, 565, 3), (3, 184, 3), (64, 96, 3)]
    jobs = []
    for rep in range(7):                                   # 70 jobs -> two launches
        for (co, ci, k) in shapes:
            w = (torch.randn(co, ci, k, k, generator=g) * 0.1).cuda()
            mode = (rep + co) % 3 if k == 3 else (rep + co) % 2
            jobs.append((w, mode))
    singles, outs = [], []
    for w, mode in jobs:
        co, ci, k, _ = w.shape
        n = _lib.lib().upf_conv_packed_bytes(co, 4 * ci, 3) if mode == 2 else _lib.lib().upf_conv_packed_bytes(co if mode else ci, ci if mode else co, k)
        a = torch.full((n // 2,), float('nan'), dtype=torch.bfloat16, device='cuda')
        _lib.call('upf_conv_pack_weights_f32', _lib.ptr(w), _lib.ptr(a), ci, co, k, _lib.UPF_BF16, mode, _lib.stream_ptr(w.device))
        singles.append(a)
        outs.append(torch.full((n // 2,), float('nan'), dtype=torch.bfloat16, device='cuda'))
    n = len(jobs)
    wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w, _ in jobs])
    op = (ctypes.c_void_p * n)(*[t.data_ptr() for t in outs])
    ci = (ctypes.c_int * n)(*[w.shape[1] for w, _ in jobs])
    co = (ctypes.c_int * n)(*[w.shape[0] for w, _ in jobs])
    ks = (ctypes.c_int * n)(*[w.shape[2] for w, _ in jobs])
    dg = (ctypes.c_int * n)(*[m for _, m in jobs])
    _lib.call('upf_conv_pack_weights_f32_multi', wp, op, ci, co, ks, dg, n, _lib.UPF_BF16, _lib.stream_ptr(torch.device('cuda', 0)))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(singles, outs)):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (i, jobs[i][0].shape, jobs[i][1])


def test_conv_prepack_fills_the_caches_the_layers_read():
    """ops.conv_prepack: after a first step has told the packers which forms each parameter needs, one launch makes all of them
    for the NEW parameter versions, and conv_train then packs nothing itself (same results as without the prepack)."""
    from upflow_pytorch_amd import ops
    g = torch.Generator().manual_seed(12)
    ws = [(torch.randn(32, 16, 3, 3, generator=g) * 0.1).cuda().requires_grad_(True), (torch.randn(16, 32, 3, 3, generator=g) * 0.1).cuda().requires_grad_(True)]
    bs = [torch.zeros(32).cuda().requires_grad_(True), torch.zeros(16).cuda().requires_grad_(True)]
    x = torch.randn(2, 16, 16, 32, generator=g).bfloat16().cuda().requires_grad_(True)

    def step():
        y = ops.conv_train(ops.conv_train(x, ws[0], bs[0], 1, 0.1, 1), ws[1], bs[1], 1, 0.1, 1)
        gx, = torch.autograd.grad(y.float().sum(), x)
        return y.detach().clone(), gx.clone()
    ops.train_caches_clear()
    want = step()                                           # registers the wanted forms
    with torch.no_grad():
        for w in ws:
            w.mul_(1.5)                                     # an optimiser step: new versions
    ref = step()
    with torch.no_grad():
        for w in ws:
            w.mul_(1.0)                                     # bump the versions, same values
    calls = {'n': 0}
    real = ops._conv_pack_from_master

    def counting(*a, **k):
        calls['n'] += 1
        return real(*a, **k)
    ops._conv_pack_from_master = counting
    try:
        ops.conv_prepack(ws)
        got = step()
    finally:
        ops._conv_pack_from_master = real
    assert calls['n'] == 0                                  # every operand came out of the one prepack launch
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and not torch.equal(got[0], want[0])
