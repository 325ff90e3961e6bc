"""Convolution with operands in the channel-octet layout (csrc/conv_c8.hip: LDS-DMA staging, two LDS buffers, C8 epilogue)
against torch's conv2d on the same 16-bit-rounded operands: every input / output layout combination, every workgroup shape
the launcher picks (Cout 2 ... 128, coarse and fine grids), dilations 1 ... 16, ragged channel counts."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# B, C8-slice channels (carried by the octets, a multiple of 8 after padding), tail channels, Cout, H, W, dilation, y_c8
CASES = [
    (1, 32, 83, 128, 16, 32, 1, True), (2, 160, 83, 128, 24, 64, 1, True), (1, 288, 83, 96, 16, 32, 1, True), (1, 384, 83, 64, 8, 32, 1, True),
    (1, 448, 83, 32, 16, 32, 1, True), (1, 480, 83, 2, 16, 32, 1, False), (1, 480, 85, 128, 16, 64, 1, True),
    (1, 64, 0, 32, 16, 32, 1, True), (1, 96, 0, 32, 32, 64, 1, True), (1, 160, 0, 16, 16, 32, 1, True), (1, 176, 0, 8, 16, 32, 1, True),
    (1, 184, 0, 3, 16, 32, 1, False), (1, 32, 0, 2, 8, 16, 1, False), (1, 64, 0, 32, 9, 24, 1, True), (2, 40, 7, 33, 7, 8, 1, True),
    (1, 128, 0, 128, 24, 32, 2, True), (1, 128, 0, 128, 48, 64, 4, True), (1, 128, 0, 96, 48, 64, 8, True), (1, 96, 0, 64, 48, 64, 16, True),
    (1, 96, 0, 64, 96, 32, 16, True), (1, 128, 0, 96, 24, 64, 8, True), (1, 128, 0, 96, 45, 64, 8, True), (1, 64, 0, 64, 23, 40, 2, True),
    (8, 128, 0, 128, 48, 160, 4, True), (8, 448, 83, 32, 96, 320, 1, True), (8, 64, 0, 32, 96, 320, 1, True),
]
MODES = {'auto': {}, 'mtw1': {'force_mtw': 1}, 'mtw2': {'force_mtw': 2}, 'th8': {'ph_fit': 0}}


@pytest.fixture(params=sorted(MODES))
def conv_mode(request):
    from upflow_pytorch_amd import ops
    prev = {k: ops.conv_set_option(k, v) for k, v in MODES[request.param].items()}
    yield request.param
    for k, v in prev.items():
        ops.conv_set_option(k, v)


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_conv_c8_matches_conv2d(case, dtype, conv_mode):
    from upflow_pytorch_amd import ops
    B, C8c, C2, Cout, H, W, d, y_c8 = case
    if conv_mode != 'auto' and B == 8:
        pytest.skip('large case: default launch shape only')
    g = torch.Generator().manual_seed(sum(case[:7]))
    Cin = C8c + C2
    x = torch.randn(B, Cin, H, W, generator=g).to(dtype).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dtype).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    want = F.leaky_relu(F.conv2d(x.float(), w.float(), b, padding=d, dilation=d), 0.1)
    # the layer's input channels [0, C8c) live in a C8 buffer (as an octet SLICE of a wider one, behind 2 foreign octets and
    # padded to whole octets with garbage-free zeros), channels [C8c, Cin) in an NCHW buffer behind 5 foreign planes
    n8 = (C8c + 7) // 8
    buf8 = torch.full((B, n8 + 3, H, W, 8), 3.0, dtype=dtype, device='cuda')
    buf8[:, 2:2 + n8] = ops.to_c8(x[:, :C8c])
    x8 = buf8[:, 2:2 + n8]
    x2 = None
    if C2:
        buf2 = torch.full((B, C2 + 5, H, W), 3.0, dtype=dtype, device='cuda')
        buf2[:, 5:] = x[:, C8c:]
        x2 = buf2[:, 5:]
        if (5 * H * W * 2) % 16:
            pytest.skip('tail slice not 16-byte aligned for this shape')
    c8_map = list(range(C8c)) + [-1] * (n8 * 8 - C8c)
    packed = ops.conv_c8_pack(w, c8_map, list(range(C8c, Cin)))
    assert ops.conv_c8_supported(H, W, dtype, Cout, d, 3, True, C2 > 0, y_c8)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    tol = eps * float(want.abs().max()) + 1e-3
    if y_c8:
        no = (Cout + 7) // 8
        ybuf = torch.full((B, no + 2, H, W, 8), 7.0, dtype=dtype, device='cuda')
        ops.conv_c8_forward_raw(x8, x2, packed, b, ybuf[:, 1:1 + no], dilation=d, leaky_slope=0.1)
        got = ops.from_c8(ybuf[:, 1:1 + no]).float()
        assert (got[:, :Cout] - want).abs().max() <= tol, float((got[:, :Cout] - want).abs().max())
        assert bool((got[:, Cout:] == 0).all()), 'the channels that pad the last octet must be zeros'
        assert bool((ybuf[:, :1] == 7).all()) and bool((ybuf[:, 1 + no:] == 7).all()), 'wrote outside its octet slice'
    else:
        ybuf = torch.full((B, Cout + 8, H, W), 7.0, dtype=dtype, device='cuda')
        ops.conv_c8_forward_raw(x8, x2, packed, b, ybuf[:, 8:8 + Cout], dilation=d, leaky_slope=0.1)
        got = ybuf[:, 8:8 + Cout].float()
        assert (got - want).abs().max() <= tol, float((got - want).abs().max())
        assert bool((ybuf[:, :8] == 7).all()), 'wrote outside its channel slice'


@pytest.mark.parametrize('case', [(2, 32, 32, 24, 40), (1, 64, 32, 16, 64), (1, 196, 32, 8, 24), (8, 32, 32, 96, 320), (1, 96, 20, 9, 16)])
def test_conv1x1_nchw_to_c8(case):
    from upflow_pytorch_amd import ops
    B, Cin, Cout, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().cuda()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).bfloat16().cuda()
    b = torch.randn(Cout, generator=g).cuda()
    want = F.leaky_relu(F.conv2d(x.float(), w.float(), b), 0.1)
    no = (Cout + 7) // 8
    y = ops.c8_empty(B, Cout, H, W, torch.bfloat16, 'cuda')
    ops.conv_c8_forward_raw(None, x, ops.conv3x3_pack(w), b, y, dilation=1, leaky_slope=0.1, kernel_size=1)
    got = ops.from_c8(y).float()
    assert (got[:, :Cout] - want).abs().max() <= 2.0 ** -8 * float(want.abs().max()) + 1e-3
    assert no * 8 == Cout or bool((got[:, Cout:] == 0).all())


def test_conv_c8_rejects_unsupported():
    from upflow_pytorch_amd import ops
    x8 = ops.c8_empty(1, 32, 8, 12, torch.bfloat16, 'cuda')           # W % 8 != 0
    w = ops.conv_c8_pack(torch.zeros(8, 32, 3, 3, dtype=torch.bfloat16, device='cuda'), list(range(32)))
    with pytest.raises(RuntimeError):
        ops.conv_c8_forward_raw(x8, None, w, torch.zeros(8, device='cuda'), ops.c8_empty(1, 8, 8, 12, torch.bfloat16, 'cuda'))
    x8 = ops.c8_empty(1, 32, 8, 16, torch.bfloat16, 'cuda')
    with pytest.raises(RuntimeError):                                  # dilation 2 needs a C8 output
        ops.conv_c8_forward_raw(x8, None, w, torch.zeros(8, device='cuda'), torch.zeros(1, 8, 8, 16, dtype=torch.bfloat16, device='cuda'), dilation=2)
