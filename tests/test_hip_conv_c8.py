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
    w = ops.conv_c8_pack(torch.zeros(8, 32, 3, 3, dtype=torch.bfloat16, device='cuda'), list(range(32)))
    x2 = torch.zeros(1, 32, 8, 12, dtype=torch.bfloat16, device='cuda')          # an NCHW input with unaligned rows (W % 8 != 0, no pitch)
    w2 = ops.conv_c8_pack(torch.zeros(8, 32, 3, 3, dtype=torch.bfloat16, device='cuda'), (), list(range(32)))
    with pytest.raises(RuntimeError):
        ops.conv_c8_forward_raw(None, x2, w2, torch.zeros(8, device='cuda'), ops.c8_empty(1, 8, 8, 12, torch.bfloat16, 'cuda'))
    # (octet inputs of any width are fine since round 5: tests/test_hip_pitched.py)
    x8 = ops.c8_empty(1, 32, 8, 16, torch.bfloat16, 'cuda')
    with pytest.raises(RuntimeError):                                  # dilation 2 needs a C8 output
        ops.conv_c8_forward_raw(x8, None, w, torch.zeros(8, device='cuda'), torch.zeros(1, 8, 8, 16, dtype=torch.bfloat16, device='cuda'), dilation=2)


@pytest.mark.parametrize('mask_mode', ['literal', 'robust', None])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_warp_c8_equals_the_nchw_warp(mask_mode, dtype):
    """upf_warp_forward_c8 (octet tensors, 16-byte gathers) == upf_warp_forward_strided bit for bit, incl. batch_shift, flows
    that leave the frame, NaN flows, and octet slices of wider buffers."""
    from upflow_pytorch_amd import ops
    g = torch.Generator().manual_seed(77)
    B, C, H, W = 4, 32, 24, 40
    x = torch.randn(B, C, H, W, generator=g).to(dtype).cuda()
    flow = (torch.randn(B, 2, H, W, generator=g) * 6).cuda()
    flow[0, :, 0, :3] = float('nan')
    flow[1, 0, 2, 5] = float('inf')
    flow[2] = 0
    want = torch.empty_like(x)
    ops.warp_into(x, flow, want, mask_mode, batch_shift=2)
    buf = torch.full((B, 11, H, W, 8), 5.0, dtype=dtype, device='cuda')
    buf[:, 2:6] = ops.to_c8(x)
    ops.warp_c8_into(buf[:, 2:6], flow, buf[:, 6:10], mask_mode, batch_shift=2)
    got = ops.from_c8(buf[:, 6:10])
    if mask_mode is None:       # (no mask: a tap outside the frame contributes value * 0 = +-0; the sign of that zero may differ)
        assert torch.equal(got.float(), want.float())
    else:
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert bool((buf[:, :2] == 5).all()) and bool((buf[:, 10:] == 5).all())


def test_whole_net_c8_levels_are_bit_identical_to_nchw():
    """The channel-octet SGU stack / context network of the large pyramid levels (UPFlow_net._forward_stacked_fast, c8_level_ok)
    change the layout, not the arithmetic or its summation order: the whole config-2 forward is bit-identical with `_no_c8`."""
    import _weights
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    from upflow_pytorch_amd.model.pwc_modules import c8_level_ok
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False, 'norm_moments_across_images': False,
                 'if_sgu_upsample': True}, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0))
    net = net.cuda().bfloat16().eval()
    im1, im2 = _weights.make_images(2, 2, 384, 1280)
    assert c8_level_ok(4, 96, 320, torch.bfloat16) and not c8_level_ok(4, 24, 80, torch.bfloat16)
    net._no_c8_est = True              # (the estimator's octet form has another K order: its own test below)
    from upflow_pytorch_amd.model import pwc_modules
    monkey_narrow, monkey_merge = pwc_modules._NO_NARROW[0], pwc_modules.MERGE_TAIL[0]
    pwc_modules._NO_NARROW[0] = True   # (so has the 16-channel instruction of the <= 16-channel layers: test_conv_c8_narrow_layers)
    pwc_modules.MERGE_TAIL[0] = False  # (... and the merged narrow tail of the octet stacks, round 6: test_merged_tail_* below)
    try:
        with torch.no_grad():
            a = net({'im1': im1.cuda(), 'im2': im2.cuda(), 'if_loss': False})
            net._no_c8 = True
            b = net({'im1': im1.cuda(), 'im2': im2.cuda(), 'if_loss': False})
    finally:
        pwc_modules._NO_NARROW[0], pwc_modules.MERGE_TAIL[0] = monkey_narrow, monkey_merge
    for k in ('flow_f_out', 'flow_b_out', 'occ_fw', 'occ_bw'):
        assert torch.equal(a[k], b[k]), k
    assert torch.isfinite(a['flow_f_out']).all() and float(a['flow_f_out'].abs().mean()) > 0


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(8, 32, 96, 320), (2, 64, 48, 160), (2, 96, 24, 80), (1, 196, 8, 24)])
def test_corr81_norm_c8_is_the_nchw_cost_volume_in_octet_order(shape, dt):
    """upf_corr81_norm_forward_c8 writes the SAME 81 values as upf_corr81_norm_forward (bit for bit), in the octet order of
    ops.corr81_c8_channel_map, zeros in the 7 padding positions; every tile geometry of the kernel (C = 32 ... 196)."""
    from upflow_pytorch_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(C)
    f1 = (torch.randn(B, C, H, W, generator=g) * 2 + 0.3).to(dt).cuda()
    f2 = (torch.randn(B, C, H, W, generator=g) * 2 - 0.1).to(dt).cuda()
    want = ops.corr81_norm_forward_raw(f1, f2, leaky_slope=0.1)
    buf8 = torch.full((B, 14, H, W, 8), float('nan'), dtype=dt, device='cuda')       # (a slice of a wider buffer)
    ops.corr81_norm_forward_c8(f1, f2, buf8[:, 2:13], leaky_slope=0.1)
    got = buf8[:, 2:13].permute(0, 1, 4, 2, 3).reshape(B, 88, H, W)
    m = ops.corr81_c8_channel_map()
    assert sorted(c for c in m if c >= 0) == list(range(81)) and len(m) == 88
    for p, c in enumerate(m):
        if c >= 0:
            assert torch.equal(got[:, p], want[:, c]), (p, c)
        else:
            assert float(got[:, p].float().abs().max()) == 0.0
    assert torch.isnan(buf8[:, :2].float()).all() and torch.isnan(buf8[:, 13:].float()).all()      # nothing outside the 11 octets


def test_flow_update_c8():
    from upflow_pytorch_amd import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randn(4, 2, 24, 40, generator=g).cuda()
    b = torch.randn(4, 2, 24, 40, generator=g).bfloat16().cuda()
    c = torch.randn(4, 2, 24, 40, generator=g).bfloat16().cuda()
    for bb, cc in ((None, None), (b, None), (b, c)):
        buf8 = torch.full((4, 3, 24, 40, 8), float('nan'), dtype=torch.bfloat16, device='cuda')
        ops.flow_update_c8(a, bb, cc, buf8[:, 1:2])
        want = ops.flow_update(a, bb, cc, out=torch.empty(4, 2, 24, 40, dtype=torch.bfloat16, device='cuda'))
        got = buf8[:, 1].permute(0, 3, 1, 2)
        assert torch.equal(got[:, :2], want) and float(got[:, 2:].float().abs().max()) == 0.0
        assert torch.isnan(buf8[:, 0].float()).all() and torch.isnan(buf8[:, 2].float()).all()


@pytest.mark.parametrize('dt,size', [(torch.bfloat16, (384, 1280)), (torch.float16, (448, 1024))])
def test_whole_net_with_the_estimator_in_octets(dt, size):
    """The flow estimator of the two fine levels in the channel-octet layout (cost volume, 1x1 features and flows written as
    octets; UPFlow_net._level_c8) is the same arithmetic with another K order inside the matrix-core sums: the whole forward
    stays within the 16-bit envelope of the NCHW path — compared with the fp32 forward it is as close as the NCHW path is."""
    import _weights
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False, 'norm_moments_across_images': False,
                 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0))
    net = net.cuda().eval()
    im1, im2 = _weights.make_images(2, 4, *size)
    with torch.no_grad():
        ref = net({'im1': im1.cuda(), 'im2': im2.cuda(), 'if_loss': False})
        net = net.to(dt)
        calls = []
        orig = net._level_c8
        net._level_c8 = lambda *a, **k: (calls.append(a[0]), orig(*a, **k))[1]
        a = net({'im1': im1.cuda(), 'im2': im2.cuda(), 'if_loss': False})
        net._no_c8_est = True
        b = net({'im1': im1.cuda(), 'im2': im2.cuda(), 'if_loss': False})
    assert len(calls) == 2                                                   # the two fine levels took the octet path
    for k in ('flow_f_out', 'flow_b_out'):
        ea = float((a[k].float() - ref[k]).pow(2).sum(1).sqrt().mean())
        eb = float((b[k].float() - ref[k]).pow(2).sum(1).sqrt().mean())
        eab = float((a[k].float() - b[k].float()).pow(2).sum(1).sqrt().mean())
        print('%s %s: EPE vs fp32: octets %.5f px, NCHW %.5f px; octets vs NCHW %.5f px' % (k, dt, ea, eb, eab))
        assert torch.isfinite(a[k]).all()
        assert ea <= 1.25 * eb + 1e-4 and eab <= 1.5 * eb + 1e-4


def test_conv3x3_stride2_nchw_to_c8():
    from upflow_pytorch_amd import ops
    g = torch.Generator().manual_seed(9)
    B, Cin, Cout, H, W = 2, 32, 32, 48, 64
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).bfloat16().cuda()
    b = torch.randn(Cout, generator=g).cuda()
    want = F.leaky_relu(F.conv2d(x.float(), w.float(), b, padding=1, stride=2), 0.1)
    y = ops.c8_empty(B, Cout, H // 2, W // 2, torch.bfloat16, 'cuda')
    ops.conv_c8_forward_raw(None, x, ops.conv3x3_pack(w), b, y, dilation=1, leaky_slope=0.1, stride=2)
    assert (ops.from_c8(y).float() - want).abs().max() <= 2.0 ** -8 * float(want.abs().max()) + 1e-3
    ref = torch.empty(B, Cout, H // 2, W // 2, dtype=torch.bfloat16, device='cuda')
    ops.conv3x3_forward_raw(x, ops.conv3x3_pack(w), b, ref, 1, 0.1, 2)
    assert torch.equal(ops.from_c8(y), ref)            # the same kernel with another epilogue: bit-identical to the NCHW output


def test_corr81_norm_c8_timed_helper_runs_the_same_kernel():
    """bench.py's roofline probe (upf_corr81_norm_forward_c8_timed): positive event durations, and the buffer it leaves equals
    what upf_corr81_norm_forward_c8 writes."""
    from upflow_pytorch_amd import ops
    g = torch.Generator().manual_seed(4)
    f1 = torch.randn(4, 32, 48, 160, generator=g).bfloat16().cuda()
    f2 = torch.randn(4, 32, 48, 160, generator=g).bfloat16().cuda()
    a = ops.c8_empty(4, 88, 48, 160, torch.bfloat16, 'cuda')
    b = ops.c8_empty(4, 88, 48, 160, torch.bfloat16, 'cuda')
    avg, mn = ops.corr81_norm_forward_c8_timed(f1, f2, a, 0.1, nrep=5)
    ops.corr81_norm_forward_c8(f1, f2, b, 0.1)
    assert 0 < mn <= avg < 1e4 and torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('case', [(8, 184, 3, 96, 320, False), (8, 176, 8, 96, 320, True), (2, 160, 16, 48, 160, True), (8, 568, 2, 96, 320, False),
                                  (2, 32, 2, 24, 40, False), (1, 64, 12, 20, 24, True), (2, 40, 16, 8, 32, True), (2, 72, 5, 33, 48, False)])
def test_conv_c8_narrow_layers(case, dt):
    """upf_conv_forward_c8_narrow (Cout <= 16 on the 16-output-channel matrix instruction) vs F.conv2d on the same rounded operands
    and vs the 32-channel kernel (same products; fp32 sums grouped 32 instead of 16 input channels per instruction): octet
    inputs with padding positions, NCHW and octet outputs, 16- and 8-row tiles, heights that are not whole tiles."""
    from upflow_pytorch_amd import ops
    B, Cin, Cout, H, W, y_c8 = case
    g = torch.Generator().manual_seed(Cin * 17 + Cout)
    n_oct = (Cin + 7) // 8
    x = torch.randn(B, Cin, H, W, generator=g).to(dt).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dt).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    x8 = ops.to_c8(x)
    if n_oct * 8 != Cin:
        x8[..., Cin - (n_oct - 1) * 8:][:, -1] = 3.0                        # finite junk in the padding positions of the last octet
    cmap = list(range(Cin)) + [-1] * (n_oct * 8 - Cin)
    want = F.leaky_relu(F.conv2d(x.float(), w.float(), b, padding=1), 0.1)
    ya = ops.c8_empty(B, Cout, H, W, dt, 'cuda') if y_c8 else torch.empty(B, Cout, H, W, dtype=dt, device='cuda')
    yb = torch.empty_like(ya)
    ops.conv_c8_forward_narrow_raw(x8, ops.conv_c8_pack16(w, cmap), b, ya, 0.1)
    ops.conv_c8_forward_raw(x8, None, ops.conv_c8_pack(w, cmap), b, yb, 1, 0.1)
    got = (ops.from_c8(ya, Cout) if y_c8 else ya).float()
    ref = (ops.from_c8(yb, Cout) if y_c8 else yb).float()
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    assert (got - want).abs().max() <= eps * float(want.abs().max()) + 1e-3
    assert (got - ref).abs().max() <= 2 * eps * float(want.abs().max())      # at most one rounding step (of the largest binade) apart
    assert float((got != ref).float().mean()) < 0.02                         # ... and that rarely
    if y_c8 and Cout % 8:
        assert float(ya[:, -1, :, :, Cout % 8:].float().abs().max()) == 0.0  # padding channels of the last output octet: exact zeros


# ----------------------------------------------------------------------------------------------- the merged narrow tail (round 6)
def _stack_reference(stack, x, dt):
    """The dense stack layer by layer through conv2d in fp32 on 16-bit-rounded activations (what every layer-per-pass path computes up
    to fp32 summation order) -> (list of the conv outputs as dt tensors, x_out fp32)."""
    feats = x
    outs = []
    for name in stack._NAMES:
        c = getattr(stack, name)[0]
        y = F.leaky_relu(F.conv2d(feats.float(), c.weight.float(), c.bias.float(), padding=1), 0.1).to(dt)
        outs.append(y)
        feats = torch.cat([y, feats], 1)
    c = stack.conv_last[0]
    return outs, F.conv2d(feats.float(), c.weight.float(), c.bias.float(), padding=1)


@pytest.mark.parametrize('kind', ['sgu', 'est'])
@pytest.mark.parametrize('shape', [(2, 48, 160), (1, 16, 32), (1, 23, 77), (8, 96, 320), (1, 9, 24)])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_merged_tail_of_a_dense_stack_matches_the_layer_per_pass_form(kind, shape, dt):
    """_PackedTailC8: conv_m + the shared-input partials of the later layers in ONE pass, the later layers finished from the partial
    (upf_conv_forward_c8_split / _narrow_init) — against the same stack with every layer its own pass and against conv2d.  Same fp32
    sums in another order: the 16-bit outputs agree to a rounding step where a sum lands next to a rounding boundary."""
    from upflow_pytorch_amd import ops
    from upflow_pytorch_amd.model import pwc_modules
    from upflow_pytorch_amd.model.pwc_modules import FlowEstimatorDense_v2, _PackedTailC8
    B, H, W = shape
    if kind == 'est' and B * H * W > 2 * 48 * 160 and dt == torch.float16:
        pytest.skip('large estimator case: bf16 only')
    torch.manual_seed(B * 1000 + H + W)
    if kind == 'sgu':
        stack = FlowEstimatorDense_v2(64, f_channels=(32, 32, 32, 16, 8), out_channel=3)       # model/upflow.py:24-60
        assert _PackedTailC8.plan([32, 32, 32, 16, 8, 3]) == (3, [4, 5])
    else:
        stack = FlowEstimatorDense_v2(120, f_channels=(128, 128, 96, 64, 32), out_channel=2)   # pwc_modules.py:250-286 (input padded to octets)
        assert _PackedTailC8.plan([128, 128, 96, 64, 32, 2]) == (2, [5])
    for m in stack.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
            torch.nn.init.normal_(m.bias, std=0.1)
    stack = stack.cuda().to(dt).eval()
    x = torch.randn(B, stack._ch_in, H, W, device='cuda').to(dt)
    with torch.no_grad():
        ref_outs, ref_last = _stack_reference(stack, x, dt)
    n_in = stack._ch_in // 8
    nconv = sum(stack._f) // 8
    res = {}
    for merge in (False, True):
        buf8 = torch.full((B, nconv + n_in + 1, H, W, 8), float('nan'), dtype=dt, device='cuda')
        buf8[:, nconv:nconv + n_in] = ops.to_c8(x)
        stack._no_merge_tail = not merge
        with torch.no_grad():
            out = stack.forward_in_buffer_c8(buf8)
        assert torch.isnan(buf8[:, nconv + n_in:].float()).all()                          # nothing written past the buffer's octets
        res[merge] = (ops.from_c8(buf8[:, :nconv]).float(), out.float())
    holders = stack.__dict__['_packed8'][None][0]
    assert isinstance(holders[-1], _PackedTailC8) and len(holders) == 7
    ref_last = ref_last.detach()
    scale = float(ref_last.abs().max())
    # merged vs layer-per-pass: the conv outputs that precede the tail are the same launches -> identical; the tail's within a rounding step
    # (conv_m itself: the same sum on the 32-channel instruction — the layer-per-pass form of a <= 16-channel layer uses the 16-channel
    # one, which adds 32 input channels per instruction instead of 16)
    tail = holders[-1]
    nt = sum(tail.widths[tail.m:5])                              # buffer order: [conv5 | conv4 | ...]
    assert torch.equal(res[True][0][:, nt:], res[False][0][:, nt:])
    all_ref = torch.cat(ref_outs[::-1], 1).float()
    eps = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    for got in (res[True][0], res[False][0]):
        assert (got - all_ref).abs().max() <= 2 * eps * float(all_ref.abs().max()) + 1e-3
    for got in (res[True][1], res[False][1]):
        assert (got - ref_last).abs().max() <= 4 * eps * scale + 1e-3, float((got - ref_last).abs().max())
    # the merged form is not further from conv2d than the layer-per-pass form (mean error, 10 % slack)
    e_m = float((res[True][1] - ref_last).abs().mean()); e_s = float((res[False][1] - ref_last).abs().mean())
    assert e_m <= 1.1 * e_s + 1e-5, (e_m, e_s)


def test_whole_net_with_merged_tails_stays_within_the_16_bit_envelope():
    """The default schedule (merged narrow tails in the octet stacks of the fine levels) against the layer-per-pass schedule and against
    the fp32 forward: as close to fp32 as the layer-per-pass schedule is."""
    import _weights
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    from upflow_pytorch_amd.model import pwc_modules
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False, 'norm_moments_across_images': False,
                 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0))
    net = net.cuda().eval()
    im1, im2 = _weights.make_images(2, 4, 384, 1280)
    with torch.no_grad():
        ref = net({'im1': im1.cuda(), 'im2': im2.cuda(), 'if_loss': False})
        net = net.bfloat16()
        calls = []
        orig = pwc_modules._PackedTailC8.main_pass
        pwc_modules._PackedTailC8.main_pass = lambda self, *a, **k: (calls.append(self.cmain), orig(self, *a, **k))[1]
        try:
            a = net({'im1': im1.cuda(), 'im2': im2.cuda(), 'if_loss': False})
        finally:
            pwc_modules._PackedTailC8.main_pass = orig
        prev = pwc_modules.MERGE_TAIL[0]
        pwc_modules.MERGE_TAIL[0] = False
        try:
            b = net({'im1': im1.cuda(), 'im2': im2.cuda(), 'if_loss': False})
        finally:
            pwc_modules.MERGE_TAIL[0] = prev
    assert sorted(calls) == [16, 16, 16, 96, 96]               # SGU stack (conv4's pass) at the two fine levels + the final up-sampling; estimator (conv3's pass) at the two fine levels
    for k in ('flow_f_out', 'flow_b_out'):
        ea = float((a[k].float() - ref[k]).pow(2).sum(1).sqrt().mean())
        eb = float((b[k].float() - ref[k]).pow(2).sum(1).sqrt().mean())
        eab = float((a[k].float() - b[k].float()).pow(2).sum(1).sqrt().mean())
        print('%s: EPE vs fp32: merged tails %.5f px, layer per pass %.5f px; merged vs layer per pass %.5f px' % (k, ea, eb, eab))
        assert torch.isfinite(a[k]).all()
        assert ea <= 1.25 * eb + 1e-4 and eab <= 1.5 * eb + 1e-4


# ----------------------------------------------------------------------------------------------- two convolutions, one launch (round 6)
PAIR_CASES = [  # B, Cin, C1, C2, H, W
    (2, 3, 16, 16, 48, 96), (1, 3, 16, 16, 37, 70), (1, 3, 16, 16, 33, 131), (1, 16, 32, 32, 24, 80), (2, 16, 32, 32, 19, 46), (1, 16, 32, 32, 188, 621),
    (1, 3, 32, 32, 20, 40), (1, 16, 16, 16, 21, 64), (1, 8, 16, 24, 16, 32), (1, 12, 32, 20, 17, 34), (1, 4, 16, 16, 18, 40), (1, 5, 16, 12, 18, 40), (1, 1, 16, 16, 9, 24), (8, 3, 16, 16, 384, 1280), (8, 16, 32, 32, 192, 640),
]


@pytest.mark.parametrize('case', PAIR_CASES)
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('y_c8,strides', [(False, (1, 2)), (True, (1, 2)), (False, (2, 1))])
def test_conv_pair_matches_the_two_layer_composition(case, dt, y_c8, strides):
    """upf_conv_pair_forward (csrc/conv_pair.hip): two 3x3 + LeakyReLU layers with strides (1, 2) (the SGU guidance stem) or (2, 1) (a stage of
    the feature pyramid) in one launch against conv2d x 2 with the intermediate rounded to the 16-bit type, and against this library's
    two launches; ragged and odd sizes (row-pitched inputs, the last pixel pair straddling W), NCHW and octet outputs, every instantiated
    (Cin, C1) class."""
    from upflow_pytorch_amd import ops
    B, Cin, C1, C2, H, W = case
    if B == 8 and (dt == torch.float16 or (strides == (1, 2) and y_c8 != (Cin == 16))):
        pytest.skip('full-size case: one combination per stride order')
    sa, sb = strides
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).to(dt).cuda()
    wa = (torch.randn(C1, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dt).cuda()
    wb = (torch.randn(C2, C1, 3, 3, generator=g) * (2.0 / (C1 * 9)) ** 0.5).to(dt).cuda()
    ba, bb = torch.randn(C1, generator=g).cuda() * 0.1, torch.randn(C2, generator=g).cuda() * 0.1
    mid = F.leaky_relu(F.conv2d(x.float(), wa.float(), ba, padding=1, stride=sa), 0.1).to(dt)
    want = F.leaky_relu(F.conv2d(mid.float(), wb.float(), bb, padding=1, stride=sb), 0.1)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    xin = ops.empty_nchw((B, Cin, H, W), dt, 'cuda')              # (row-pitched at ragged widths, NaN in the padding)
    if xin.stride(2) != W:
        xin.as_strided((B, Cin, H, xin.stride(2)), xin.stride()).fill_(float('nan'))
    xin.copy_(x)
    if ops.nchw_pitch(xin) % 2:
        pytest.skip('odd contiguous width: the pair kernel needs an even pitch (the model falls back to two launches)')
    pa, pb = ops.conv_pair_pack(wa, wb)
    if y_c8:
        buf = torch.full((B, (C2 + 7) // 8 + 2, Ho, Wo, 8), 3.0, dtype=dt, device='cuda')
        y = buf[:, 1:1 + (C2 + 7) // 8]
        ops.conv_pair_forward_raw(xin, pa, ba, 0.1, pb, bb, 0.1, y, strides)
        got = ops.from_c8(y, C2).float()
        assert bool((buf[:, 0] == 3).all()) and bool((buf[:, -1] == 3).all())
        if C2 % 8:
            assert float(ops.from_c8(y)[:, C2:].float().abs().max()) == 0.0      # the padding channels of the last octet are zeros
    else:
        big = torch.full((B, C2 + 3, Ho, Wo), 5.0, dtype=dt, device='cuda')
        ops.conv_pair_forward_raw(xin, pa, ba, 0.1, pb, bb, 0.1, big[:, 2:2 + C2], strides)
        got = big[:, 2:2 + C2].float()
        assert bool((big[:, :2] == 5).all()) and bool((big[:, 2 + C2:] == 5).all())
    eps = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    scale = float(want.abs().max())
    # (one rounding step of the output + the rare intermediate element that rounds the other way: both forms carry those)
    assert (got - want).abs().max() <= 3 * eps * scale + 1e-3, float((got - want).abs().max())
    assert float((got - want).abs().mean()) <= 0.3 * eps * scale
    # ... and against the two launches of this library (same products, another summation order inside the first layer)
    m2 = torch.empty((B, C1) + ((H, W) if sa == 1 else (Ho, Wo)), dtype=dt, device='cuda')
    ops.conv3x3_forward_raw(x, ops.conv3x3_pack(wa), ba, m2, 1, 0.1, sa, 3)
    y2 = torch.empty(B, C2, Ho, Wo, dtype=dt, device='cuda')
    ops.conv3x3_forward_raw(m2, ops.conv3x3_pack(wb), bb, y2, 1, 0.1, sb, 3)
    assert (got - y2.float()).abs().max() <= 3 * eps * scale + 1e-3
    assert float((got != y2.float()).float().mean()) <= 0.02                       # the overwhelming majority of the outputs: the same bits


def test_whole_net_with_the_fused_guidance_stem_stays_within_the_16_bit_envelope():
    """The SGU guidance stem as two fused launches (default) against four launches: the whole config-2 forward moves by less than the
    16-bit envelope, and is as close to the fp32 forward."""
    import _weights
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    from upflow_pytorch_amd.model import pwc_modules
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False, 'norm_moments_across_images': False,
                 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0))
    net = net.cuda().eval()
    im1, im2 = _weights.make_images(2, 4, 384, 1280)
    with torch.no_grad():
        ref = net({'im1': im1.cuda(), 'im2': im2.cuda(), 'if_loss': False})
        net = net.bfloat16()
        calls = []
        orig = pwc_modules._PackedConvPair.__call__
        pwc_modules._PackedConvPair.__call__ = lambda self, *a, **k: (calls.append(self.convs[0].in_channels), orig(self, *a, **k))[1]
        try:
            a = net({'im1': im1.cuda(), 'im2': im2.cuda(), 'if_loss': False})
        finally:
            pwc_modules._PackedConvPair.__call__ = orig
        prev = pwc_modules.FUSE_PAIRS[0]
        pwc_modules.FUSE_PAIRS[0] = False
        try:
            b = net({'im1': im1.cuda(), 'im2': im2.cuda(), 'if_loss': False})
        finally:
            pwc_modules.FUSE_PAIRS[0] = prev
    assert calls == [3, 16, 3, 16]                                  # feature pyramid stages 0 and 1 (strides 2, 1), then the guidance stem's halves (1, 2)
    for k in ('flow_f_out', 'flow_b_out'):
        ea = float((a[k].float() - ref[k]).pow(2).sum(1).sqrt().mean())
        eb = float((b[k].float() - ref[k]).pow(2).sum(1).sqrt().mean())
        eab = float((a[k].float() - b[k].float()).pow(2).sum(1).sqrt().mean())
        print('%s: EPE vs fp32: fused stem %.5f px, four launches %.5f px; fused vs four launches %.5f px' % (k, ea, eb, eab))
        assert ea <= 1.25 * eb + 1e-4 and eab <= 1.5 * eb + 1e-4


@pytest.mark.parametrize('case', [(2, 32, 32, 48, 160), (1, 64, 32, 23, 78), (8, 32, 32, 96, 320), (1, 96, 20, 9, 156)])
@pytest.mark.parametrize('dts', [(torch.bfloat16, torch.bfloat16), (torch.float16, torch.bfloat16), (torch.float16, torch.float16)])
def test_conv1x1_dual_destination_equals_two_launches(case, dts):
    """upf_conv1x1_forward_c8_dual: the 1x1 projection NCHW -> octets stored into two buffers by one launch == the two launches it
    replaces, bit for bit, in both buffers; nothing outside the destination octets is written; mixed types (fp16 features -> bf16 octets)."""
    from upflow_pytorch_amd import ops
    B, Cin, Cout, H, W = case
    dti, dto = dts
    g = torch.Generator().manual_seed(sum(case))
    x = ops.empty_nchw((B, Cin, H, W), dti, 'cuda')
    x.copy_(torch.randn(B, Cin, H, W, generator=g))
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).to(dti).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    packed = ops.conv_c8_pack(w, (), range(Cin))
    no = (Cout + 7) // 8
    want = ops.c8_empty(B, Cout, H, W, dto, 'cuda')
    ops.conv_c8_forward_raw(None, x, packed, b, want, dilation=1, leaky_slope=0.1, kernel_size=1)
    a = torch.full((B, no + 2, H, W, 8), 3.0, dtype=dto, device='cuda')
    c = torch.full((B, no + 3, H, W, 8), 5.0, dtype=dto, device='cuda')
    ops.conv1x1_c8_dual_raw(x, packed, b, a[:, 1:1 + no], c[:, 2:2 + no], 0.1)
    assert torch.equal(a[:, 1:1 + no].contiguous().view(torch.int16), want.view(torch.int16))
    assert torch.equal(c[:, 2:2 + no].contiguous().view(torch.int16), want.view(torch.int16))
    assert bool((a[:, 0] == 3).all()) and bool((a[:, -1] == 3).all()) and bool((c[:, :2] == 5).all()) and bool((c[:, -1] == 5).all())


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(2, 24, 80), (1, 13, 37), (8, 96, 320)])
def test_blend_stores_the_estimators_flow_slot_itself(shape, dt):
    """upf_sgu_blend_forward_flow16 == upf_sgu_blend_forward followed by upf_flow_update / upf_flow_update_c8 of its output, bit for bit, for
    both slot layouts (two NCHW planes of a wider buffer / one octet of a channel-octet buffer); nothing else in the buffers is touched."""
    from upflow_pytorch_amd import ops
    B, H, W = shape
    g = torch.Generator().manual_seed(B * 7 + H + W)
    flow = (torch.randn(B, 2, H, W, generator=g) * 3).cuda()
    x_out = torch.randn(B, 3, H, W, generator=g).to(dt).cuda()
    want_up = ops.sgu_blend(flow, x_out, None, want_inter=False)[1]
    big = torch.full((B, 7, H, W), 5.0, dtype=dt, device='cuda')
    up_a = ops.sgu_blend_flow16(flow, x_out, big[:, 3:5])
    ref = torch.full((B, 7, H, W), 5.0, dtype=dt, device='cuda')
    ops.flow_update(want_up, out=ref[:, 3:5])
    assert torch.equal(up_a, want_up) and torch.equal(big.view(torch.int16), ref.view(torch.int16))
    b8 = torch.full((B, 3, H, W, 8), 3.0, dtype=dt, device='cuda')
    up_b = ops.sgu_blend_flow16(flow, x_out, b8[:, 1:2])
    r8 = torch.full((B, 3, H, W, 8), 3.0, dtype=dt, device='cuda')
    ops.flow_update_c8(want_up, None, None, r8[:, 1:2])
    assert torch.equal(up_b, want_up) and torch.equal(b8.view(torch.int16), r8.view(torch.int16))
