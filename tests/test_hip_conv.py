"""Matrix-core 3x3 convolution (csrc/conv3x3.hip) against torch's conv2d on the same bf16/fp16-rounded
operands (fp32 reference arithmetic), incl. channel-slice inputs/outputs, dilations, ragged Cin/Cout."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # B, Cin, Cout, H, W, dilation
    (1, 32, 32, 8, 32, 1), (2, 115, 128, 24, 40, 1), (1, 243, 128, 16, 64, 1), (1, 371, 96, 9, 24, 1), (1, 467, 64, 8, 8, 1),
    (1, 563, 2, 12, 40, 1), (1, 565, 128, 16, 32, 1), (1, 128, 128, 24, 48, 2), (1, 128, 128, 24, 48, 4), (1, 128, 96, 40, 64, 8), (1, 96, 64, 40, 64, 16), (2, 96, 64, 24, 80, 16),
    (2, 64, 32, 17, 56, 1), (1, 184, 3, 8, 16, 1), (1, 7, 5, 3, 8, 1), (4, 96, 32, 96, 320, 1),
    # rows that are not 16-byte aligned (W % 8 != 0, odd W, odd channel-slice offsets) and Cout > 128
    (2, 115, 128, 6, 20, 1), (1, 565, 96, 6, 20, 1), (2, 64, 196, 6, 20, 1), (1, 96, 64, 6, 20, 16), (1, 128, 128, 12, 26, 4),
    (1, 40, 33, 7, 13, 1), (1, 35, 2, 5, 9, 1), (2, 48, 64, 24, 52, 2), (1, 16, 160, 9, 75, 1), (1, 196, 196, 8, 24, 1),
    # row-phase layers whose phases hold 6 / 3 / 12 rows (the 6- and 4-row tiles of round 3), odd heights (unequal phases)
    (1, 128, 128, 48, 64, 4), (1, 128, 96, 48, 64, 8), (1, 96, 64, 48, 64, 16), (1, 96, 64, 96, 64, 16), (1, 128, 96, 24, 64, 8),
    (1, 128, 96, 45, 64, 8), (1, 64, 64, 23, 40, 2), (1, 96, 64, 47, 45, 16), (2, 128, 128, 96, 32, 8), (1, 128, 96, 13, 27, 4)]


# launch heuristics under which every case runs: the defaults (coarse grids -> split-K kernel), every grid
# through the tiled kernel with 4 / 2 / 1 channel blocks per workgroup by Cout (and 16-row tiles for Cout <= 32),
# every grid through 32-channel slabs over blockIdx.y, the split-K kernel wherever it applies (with the row-phase
# layers on their 8-row tiles), and 64-channel workgroups with 8-row tiles everywhere
MODES = {'auto': {}, 'tiled': {'force_sk': 0, 'small_grid': 0, 'rpw4_min': 0},
         'slabs': {'force_sk': 0, 'force_mtw': 1, 'rpw4_min': 1 << 30}, 'splitk': {'force_sk': 1, 'ph_fit': 0},
         'mtw2_th8': {'force_sk': 0, 'force_mtw': 2, 'ph_fit': 0}}


@pytest.fixture(params=sorted(MODES))
def conv_mode(request):
    from upflow_pytorch_amd import ops
    prev = {k: ops.conv_set_option(k, v) for k, v in MODES[request.param].items()}
    yield request.param
    for k, v in prev.items():
        ops.conv_set_option(k, v)


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_conv3x3_matches_conv2d(case, dtype, conv_mode):
    from upflow_pytorch_amd import ops
    B, Cin, Cout, H, W, d = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).to(dtype).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dtype).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    want = F.leaky_relu(F.conv2d(x.float(), w.float(), b, padding=d, dilation=d), 0.1)
    # channel-slice operands of wider buffers, like the dense estimator's concat buffer
    xbuf = torch.zeros(B, Cin + 5, H, W, dtype=dtype, device='cuda')
    xbuf[:, 5:] = x
    ybuf = torch.full((B, Cout + 3, H, W), 7.0, dtype=dtype, device='cuda')
    xv = xbuf[:, 5:]                        # 5*H*W elements in: 16-byte aligned for some cases, not for others
    assert ops.conv3x3_supported(xv, Cout, d)
    packed = ops.conv3x3_pack(w)
    ops.conv3x3_forward_raw(xv, packed, b, ybuf[:, 2:2 + Cout], dilation=d, leaky_slope=0.1)
    got = ybuf[:, 2:2 + Cout].float()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    tol = eps * float(want.abs().max()) + 1e-3
    assert (got - want).abs().max() <= tol, float((got - want).abs().max())
    assert bool((ybuf[:, :2] == 7).all()) and bool((ybuf[:, 2 + Cout:] == 7).all()), 'wrote outside its channel slice'
    # no activation
    ops.conv3x3_forward_raw(xv, packed, b, ybuf[:, 2:2 + Cout], dilation=d, leaky_slope=0.0)
    want0 = F.conv2d(x.float(), w.float(), b, padding=d, dilation=d)
    assert (ybuf[:, 2:2 + Cout].float() - want0).abs().max() <= eps * float(want0.abs().max()) + 1e-3


def test_conv3x3_rejects_unsupported():
    from upflow_pytorch_amd import ops
    x = torch.zeros(1, 8, 8, 6, dtype=torch.bfloat16, device='cuda')        # rows shorter than one 8-pixel group
    assert not ops.conv3x3_supported(x, 8, 1)
    x = torch.zeros(1, 8, 8, 16, dtype=torch.bfloat16, device='cuda')
    assert ops.conv3x3_supported(x.float(), 8, 1)                              # (fp32: the split-precision kernel, round 4)
    assert ops.conv3x3_supported(torch.zeros(1, 8, 8, 6, device='cuda'), 8, 1)   # ... which takes any size
    assert not ops.conv3x3_supported(x, 8, 17) and not ops.conv3x3_supported(x, 8, 2, 2)
    w = ops.conv3x3_pack(torch.zeros(8, 8, 3, 3, dtype=torch.bfloat16, device='cuda'))
    with pytest.raises(RuntimeError):
        ops.conv3x3_forward_raw(x, w, torch.zeros(8, device='cuda'), torch.zeros(1, 8, 8, 16, dtype=torch.bfloat16, device='cuda'), dilation=17)


@pytest.mark.parametrize('case', [(1, 3, 16, 64, 128), (2, 16, 32, 32, 64), (1, 32, 64, 24, 40), (1, 64, 96, 17, 24), (2, 16, 16, 48, 64),
                                  (2, 128, 196, 12, 40), (1, 96, 128, 24, 52), (1, 128, 196, 12, 26), (1, 32, 64, 13, 27)])
def test_conv3x3_stride2(case, conv_mode):
    from upflow_pytorch_amd import ops
    B, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).bfloat16().cuda()
    b = torch.randn(Cout, generator=g).cuda()
    want = F.leaky_relu(F.conv2d(x.float(), w.float(), b, padding=1, stride=2), 0.1)
    assert ops.conv3x3_supported(x, Cout, 1, 2)
    ho, wo = ops.conv3x3_out_hw(H, W, 2)
    assert (ho, wo) == tuple(want.shape[2:])
    y = torch.empty(B, Cout, ho, wo, dtype=torch.bfloat16, device='cuda')
    ops.conv3x3_forward_raw(x, ops.conv3x3_pack(w), b, y, dilation=1, leaky_slope=0.1, stride=2)
    assert (y.float() - want).abs().max() <= 2.0 ** -8 * float(want.abs().max()) + 1e-3


@pytest.mark.parametrize('case', [(2, 32, 32, 24, 40), (1, 64, 32, 16, 64), (1, 96, 32, 9, 16), (1, 128, 32, 12, 40), (4, 32, 32, 96, 320),
                                  (2, 196, 32, 6, 20), (1, 128, 32, 12, 26), (1, 16, 32, 7, 13), (1, 64, 200, 6, 20)])
def test_conv1x1(case, conv_mode):
    from upflow_pytorch_amd import ops
    B, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().cuda()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).bfloat16().cuda()
    b = torch.randn(Cout, generator=g).cuda()
    want = F.leaky_relu(F.conv2d(x.float(), w.float(), b), 0.1)
    assert ops.conv3x3_supported(x, Cout, 1, 1, 1)
    y = torch.empty(B, Cout, H, W, dtype=torch.bfloat16, device='cuda')
    ops.conv3x3_forward_raw(x, ops.conv3x3_pack(w), b, y, dilation=1, leaky_slope=0.1, stride=1, kernel_size=1)
    assert (y.float() - want).abs().max() <= 2.0 ** -8 * float(want.abs().max()) + 1e-3


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', [  # Cin, Cout, H, W, k, dilation, stride, LeakyReLU
    (196, 196, 2, 4, 3, 1, 1, True), (128, 196, 3, 7, 3, 1, 2, True), (565, 128, 1, 2, 3, 1, 1, True), (128, 96, 2, 5, 3, 8, 1, True),
    (96, 64, 4, 6, 3, 16, 1, True), (196, 32, 2, 4, 1, 1, 1, False), (563, 2, 1, 1, 3, 1, 1, False), (35, 7, 3, 3, 3, 2, 1, True)])
def test_rows_shorter_than_8_pixels_leave_miopen_too(case, dtype):
    """pwc_modules.fast_conv_seq on 16-bit tensors whose rows are shorter than the 16-bit kernel's 8-pixel tile (the coarsest levels
    of small inputs): the same contraction through the split-precision kernel on fp32 copies (_PackedConv3x3.__call__) — equal to
    F.conv2d on the same rounded operands up to the final rounding to 16 bits, written into a channel slice, and bit-reproducible
    (MIOpen's fp16 kernels, which these levels used until round 4, are not: tools/pipe_stress_small.py)."""
    from upflow_pytorch_amd.model import pwc_modules as pm
    Cin, Cout, H, W, k, d, s, relu = case
    torch.manual_seed(sum(case))
    seq = pm.conv(Cin, Cout, kernel_size=k, stride=s, dilation=d, isReLU=relu).cuda().to(dtype)
    x = torch.randn(2, Cin, H, W, device='cuda').to(dtype)
    ho, wo = (H - 1) // s + 1, (W - 1) // s + 1
    buf = torch.full((2, Cout + 5, ho, wo), 3.0, device='cuda', dtype=dtype)
    cache = {}
    with torch.no_grad():
        y = pm.fast_conv_seq(seq, x, cache, out=buf[:, 2:2 + Cout])
        y2 = pm.fast_conv_seq(seq, x, cache).clone()
        want = F.conv2d(x.float(), seq[0].weight.float(), seq[0].bias.float(), stride=s, padding=d * (k - 1) // 2, dilation=d)
        if relu:
            want = F.leaky_relu(want, 0.1)
    assert len(cache) == 1 and next(iter(cache.values())).packed32 is not None          # (the hand-written path ran, not nn.Conv2d)
    assert bool((buf[:, :2] == 3).all()) and bool((buf[:, 2 + Cout:] == 3).all())
    assert torch.equal(y, y2)
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    assert float((y.float() - want).abs().max()) <= ulp * float(want.abs().max()) + 1e-6
