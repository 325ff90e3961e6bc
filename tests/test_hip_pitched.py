"""Row-pitched NCHW tensors and ragged-width octet levels (round 5): KITTI's native 375x1242 frames make every
pyramid level ragged (W = 621, 311, 156, 78, 39, 20).  The inference schedule keeps its 16-bit NCHW tensors PITCHED there (rows
16-byte aligned, ops.empty_nchw) and runs the fine levels' dense stacks in the channel-octet layout like at W % 8 == 0.

Every pitched kernel must return the bits of its contiguous ("ragged", rounds 1-4) form — and must not depend on what the pitch
padding holds: the padding columns of every pitched INPUT below are filled with NaN."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DT = [torch.bfloat16, torch.float16]


def pitched_copy(t, fill=float('nan')):
    """A pitched twin of a contiguous [..., H, W] tensor whose padding columns hold `fill`."""
    from upflow_pytorch_amd import ops
    W = t.shape[-1]
    Wp = (W + 7) // 8 * 8
    base = torch.full(tuple(t.shape[:-1]) + (Wp,), fill, dtype=t.dtype, device=t.device)
    base[..., :W] = t
    v = base[..., :W]
    assert Wp == W or ops.nchw_pitch(v.reshape(-1, *v.shape[-3:]) if v.dim() > 4 else v) == Wp
    return v


def bits(t):
    return t.contiguous().view(torch.int16)


# ----------------------------------------------------------------------------------------------- convolution, NCHW -> NCHW
CONV_CASES = [
    # B, Cin, Cout, H, W, stride, dilation, k
    (2, 32, 32, 12, 39, 1, 1, 3), (1, 16, 16, 23, 78, 2, 1, 3), (1, 3, 16, 21, 50, 1, 1, 3), (2, 64, 96, 9, 20, 1, 1, 3),
    (1, 96, 128, 24, 78, 1, 2, 3), (1, 128, 32, 10, 156, 1, 1, 1), (2, 32, 64, 47, 156, 1, 4, 3), (1, 16, 32, 47, 311, 1, 1, 3),
    (1, 32, 32, 48, 311, 2, 1, 3), (1, 200, 64, 6, 20, 1, 1, 3), (4, 565, 128, 12, 39, 1, 1, 3), (1, 64, 2, 24, 78, 1, 1, 3),
    (1, 128, 128, 24, 78, 1, 16, 3),
]


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('which', ['xy', 'x', 'y'])
def test_conv_pitched_equals_contiguous(case, dtype, which):
    from upflow_pytorch_amd import ops
    B, Cin, Cout, H, W, S, d, k = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).to(dtype).cuda()
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(dtype).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    packed = ops.conv3x3_pack(w)
    Ho, Wo = ops.conv3x3_out_hw(H, W, S)
    want = torch.empty(B, Cout, Ho, Wo, dtype=dtype, device='cuda')
    ops.conv3x3_forward_raw(x, packed, b, want, d, 0.1, S, k)
    ref = F.leaky_relu(F.conv2d(x.float(), w.float(), b, stride=S, padding=d * (k - 1) // 2, dilation=d), 0.1)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert (want.float() - ref).abs().max() <= eps * float(ref.abs().max()) + 1e-3
    xin = pitched_copy(x) if 'x' in which else x
    yout = pitched_copy(torch.zeros_like(want), fill=7.0) if 'y' in which else torch.empty_like(want)
    ops.conv3x3_forward_raw(xin, packed, b, yout, d, 0.1, S, k)
    assert torch.equal(bits(yout), bits(want)), 'pitched convolution differs from the contiguous one'


def test_conv_pitched_as_channel_slices_of_wider_pitched_buffers():
    """x / y as channel slices of wider pitched buffers with a batch stride (the dense stacks' use), odd W."""
    from upflow_pytorch_amd import ops
    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, H, W = 2, 40, 24, 11, 39
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).bfloat16().cuda()
    b = torch.randn(Cout, generator=g).cuda()
    packed = ops.conv3x3_pack(w)
    want = torch.empty(B, Cout, H, W, dtype=torch.bfloat16, device='cuda')
    ops.conv3x3_forward_raw(x, packed, b, want, 1, 0.1)
    big = ops.empty_nchw((B, Cin + Cout + 8, H, W), torch.bfloat16, 'cuda')
    assert ops.nchw_pitch(big) == 40
    big.fill_(3.0)
    big[:, 8 + Cout:] = x
    ops.conv3x3_forward_raw(big[:, 8 + Cout:], packed, b, big[:, 8:8 + Cout], 1, 0.1)
    assert torch.equal(bits(big[:, 8:8 + Cout]), bits(want))
    assert bool((big[:, :8] == 3).all()), 'wrote outside its channel slice'


def test_a_column_crop_of_a_wider_live_buffer_is_rejected_as_an_output():
    """`big[..., :W]` has the strides of a pitched tensor, but its "padding" columns are live data that the whole-segment
    stores of the pitched kernels would overwrite: outputs must be contiguous rows or padded as ops.empty_nchw pads them."""
    from upflow_pytorch_amd import ops
    g = torch.Generator().manual_seed(9)
    B, C, H, W = 1, 32, 12, 37
    x = torch.randn(B, C, H, W, generator=g).bfloat16().cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).bfloat16().cuda()
    b = torch.zeros(C).cuda()
    packed = ops.conv3x3_pack(w)
    for Wbig in (W + 11, 64, W + 8):
        big = torch.full((B, C, H, Wbig), 5.0, dtype=torch.bfloat16, device='cuda')
        with pytest.raises(ops.UpflowHipError):
            ops.conv3x3_forward_raw(x, packed, b, big[..., :W], 1, 0.1)
        with pytest.raises(ops.UpflowHipError):
            ops.warp_into(x, torch.zeros(B, 2, H, W, device='cuda'), big[..., :W], 'literal')
        assert bool((big == 5).all())
    # as an INPUT such a view is fine (nothing is written), and the empty_nchw padding is accepted as an output
    big = torch.full((B, C, H, 64), float('nan'), dtype=torch.bfloat16, device='cuda')
    big[..., :W] = x
    want = torch.empty(B, C, H, W, dtype=torch.bfloat16, device='cuda')
    ops.conv3x3_forward_raw(x, packed, b, want, 1, 0.1)
    got = ops.empty_nchw((B, C, H, W), torch.bfloat16, 'cuda')
    ops.conv3x3_forward_raw(big[..., :W], packed, b, got, 1, 0.1)
    assert torch.equal(bits(got), bits(want))


# ----------------------------------------------------------------------------------------------- convolution, NCHW -> C8 and C8 -> *
@pytest.mark.parametrize('case', [(2, 32, 32, 12, 39, 1), (1, 64, 32, 23, 78, 1), (1, 196, 32, 6, 20, 1), (2, 32, 32, 47, 311, 1), (1, 96, 20, 9, 156, 1),
                                  (2, 32, 32, 47, 155, 3), (1, 32, 32, 188, 621, 3)])
@pytest.mark.parametrize('dtype', DT)
def test_conv_nchw_to_c8_at_ragged_widths(case, dtype):
    """The entries into the octet world at a ragged level — the 1x1 projection of the pyramid features, and (k = 3: stride 2) the
    last layer of the SGU guidance stem — read PITCHED NCHW rows and write octets: same bits as the NCHW kernel on the contiguous
    tensor, re-laid as octets."""
    from upflow_pytorch_amd import ops
    B, Cin, Cout, H, W, k = case
    S = 2 if k == 3 else 1
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).to(dtype).cuda()
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(dtype).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    Ho, Wo = ops.conv3x3_out_hw(H, W, S)
    ref = F.leaky_relu(F.conv2d(x.float(), w.float(), b, stride=S, padding=(k - 1) // 2), 0.1)
    y = ops.c8_empty(B, Cout, Ho, Wo, dtype, 'cuda')
    y.fill_(9.0)
    packed = ops.conv_c8_pack(w, (), range(Cin))
    ops.conv_c8_forward_raw(None, pitched_copy(x), packed, b, y, dilation=1, leaky_slope=0.1, kernel_size=k, stride=S)
    got = ops.from_c8(y)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert (got[:, :Cout].float() - ref).abs().max() <= eps * float(ref.abs().max()) + 1e-3
    assert Cout % 8 == 0 or bool((got[:, Cout:] == 0).all())
    # bit for bit: the same kernel on the frame embedded in a contiguous 8-aligned width (zero columns behind W are the zero padding)
    Wa = (W + 15) // 16 * 16
    xa = torch.zeros(B, Cin, H, Wa, dtype=dtype, device='cuda')
    xa[..., :W] = x
    Hoa, Woa = ops.conv3x3_out_hw(H, Wa, S)
    ya = ops.c8_empty(B, Cout, Hoa, Woa, dtype, 'cuda')
    ops.conv_c8_forward_raw(None, xa, packed, b, ya, dilation=1, leaky_slope=0.1, kernel_size=k, stride=S)
    assert torch.equal(bits(ops.from_c8(ya)[..., :Wo]), bits(got))
    with pytest.raises(RuntimeError):          # contiguous ragged rows are not 16-byte aligned: rejected, never mis-read
        ops.conv_c8_forward_raw(None, x, packed, b, y, dilation=1, leaky_slope=0.1, kernel_size=k, stride=S)


C8_CASES = [
    # B, C8 channels, Cout, H, W, dilation, y_c8
    (1, 64, 32, 16, 39, 1, True), (2, 160, 128, 24, 78, 1, True), (1, 96, 64, 47, 156, 1, True), (1, 128, 128, 48, 311, 2, True),
    (1, 128, 96, 24, 78, 8, True), (1, 96, 64, 47, 156, 16, True), (1, 64, 32, 94, 311, 1, True), (2, 448, 32, 47, 311, 1, True),
    (1, 184, 3, 16, 39, 1, False), (1, 568, 2, 47, 311, 1, False), (1, 176, 8, 47, 156, 1, True), (1, 160, 16, 24, 311, 1, True),
    (1, 128, 128, 47, 156, 4, True),
]


@pytest.mark.parametrize('case', C8_CASES)
@pytest.mark.parametrize('dtype', DT)
def test_conv_c8_at_ragged_widths_matches_conv2d(case, dtype):
    """Octet tensors have aligned rows for EVERY width (a pixel is one 16-byte entry): the C8 -> C8 / C8 -> NCHW layers — LDS-DMA
    staging, wide / narrow (16-channel MFMA) kernels, dilations — at the ragged level widths, against conv2d on the same operands."""
    from upflow_pytorch_amd import ops
    from upflow_pytorch_amd.model.pwc_modules import _PackedConvC8, conv
    B, C8c, Cout, H, W, d, y_c8 = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = torch.randn(B, C8c, H, W, generator=g).to(dtype).cuda()
    seq = conv(C8c, Cout, 3, 1, d, isReLU=True).to(dtype).cuda()
    with torch.no_grad():
        seq[0].weight.copy_((torch.randn(Cout, C8c, 3, 3, generator=g) * (2.0 / (C8c * 9)) ** 0.5).to(dtype))
        seq[0].bias.copy_(torch.randn(Cout, generator=g).to(dtype))
    want = F.leaky_relu(F.conv2d(x.float(), seq[0].weight.float(), seq[0].bias.float(), padding=d, dilation=d), 0.1)
    pc = _PackedConvC8(seq, range(C8c))                       # (the model's own dispatch: narrow kernel for Cout <= 16)
    x8 = ops.to_c8(x)
    y = ops.c8_empty(B, Cout, H, W, dtype, 'cuda') if y_c8 else torch.empty(B, Cout, H, W, dtype=dtype, device='cuda')
    pc(x8, None, y)
    got = (ops.from_c8(y)[:, :Cout] if y_c8 else y).float()
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert (got - want).abs().max() <= eps * float(want.abs().max()) + 1e-3, float((got - want).abs().max())
    # the same layer on the same data embedded in an 8-aligned width (zero columns behind W == the convolution's zero padding):
    # the ragged launch must produce the same bits in its W columns
    Wa = (W + 7) // 8 * 8
    xa = torch.zeros(B, C8c, H, Wa, dtype=dtype, device='cuda')
    xa[..., :W] = x
    ya = ops.c8_empty(B, Cout, H, Wa, dtype, 'cuda') if y_c8 else torch.empty(B, Cout, H, Wa, dtype=dtype, device='cuda')
    pc(ops.to_c8(xa), None, ya)
    gota = (ops.from_c8(ya)[:, :Cout] if y_c8 else ya)[..., :W]
    assert torch.equal(bits(gota), bits(ops.from_c8(y)[:, :Cout] if y_c8 else y)), 'ragged-width octet launch differs from the aligned embedding'


# ----------------------------------------------------------------------------------------------- warp
@pytest.mark.parametrize('shape', [(4, 32, 12, 39), (2, 64, 24, 78), (2, 32, 47, 311), (2, 96, 9, 20), (2, 16, 23, 156)])
@pytest.mark.parametrize('dtype', DT)
@pytest.mark.parametrize('mask_mode', ['literal', 'robust', None])
def test_warp_pitched_equals_contiguous(shape, dtype, mask_mode):
    from upflow_pytorch_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, C, H, W, generator=g).to(dtype).cuda()
    flow = (torch.randn(B, 2, H, W, generator=g) * 5).cuda()
    flow[0, :, 0, :3] = float('nan')
    flow[1] = 0
    want = torch.empty_like(x)
    ops.warp_into(x, flow, want, mask_mode, batch_shift=B // 2)
    for xin, yout in ((pitched_copy(x), pitched_copy(torch.zeros_like(x), fill=7.0)), (pitched_copy(x), torch.empty_like(x)),
                      (x, pitched_copy(torch.zeros_like(x), fill=7.0))):
        ops.warp_into(xin, flow, yout, mask_mode, batch_shift=B // 2)
        if mask_mode is None:      # (a tap outside the frame contributes value * 0 = +-0: compare values)
            assert torch.equal(yout.float(), want.float())
        else:
            assert torch.equal(bits(yout), bits(want))


# ----------------------------------------------------------------------------------------------- statistics + cost volume
@pytest.mark.parametrize('shape', [(2, 32, 47, 311), (2, 64, 24, 156), (4, 96, 24, 78), (2, 128, 12, 39), (2, 196, 6, 20), (8, 32, 94, 311), (2, 32, 47, 155)])
@pytest.mark.parametrize('dtype', DT)
def test_corr81_norm_on_pitched_features_equals_contiguous(shape, dtype):
    """Statistics (the contiguous form's summation order on gathered addresses) + cost volume on pitched feature pairs:
    NCHW output == the contiguous launch; octet output (any W: the PADW kernel) == the same values in octet order."""
    from upflow_pytorch_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    pair = (torch.randn(2, B, C, H, W, generator=g) * 1.7 + 0.3).to(dtype).cuda()
    want = ops.corr81_norm_forward_raw(pair[0], pair[1], leaky_slope=0.1)
    pp = pitched_copy(pair)
    got = ops.corr81_norm_forward_raw(pp[0], pp[1], leaky_slope=0.1)
    assert torch.equal(bits(got), bits(want))
    out8 = ops.c8_empty(B, 88 + 16, H, W, dtype, 'cuda')
    out8.fill_(5.0)
    ops.corr81_norm_forward_c8(pp[0], pp[1], out8[:, 1:12], leaky_slope=0.1)
    m = ops.corr81_c8_channel_map()
    flat = ops.from_c8(out8[:, 1:12])
    for pos, ch in enumerate(m):
        if ch >= 0:
            assert torch.equal(bits(flat[:, pos]), bits(want[:, ch])), (pos, ch)
        else:
            assert bool((flat[:, pos] == 0).all())
    assert bool((out8[:, :1] == 5).all()) and bool((out8[:, 12:] == 5).all()), 'wrote outside its 11 octets'
    if W % 8:
        with pytest.raises(RuntimeError):      # contiguous ragged rows: the octet form needs aligned rows
            ops.corr81_norm_forward_c8(pair[0], pair[1], out8[:, 1:12], leaky_slope=0.1)


# ----------------------------------------------------------------------------------------------- whole network
def _net(dtype=torch.bfloat16):
    import _weights
    from upflow_pytorch_amd.model.upflow import UPFlow_net
    conf = UPFlow_net.config()
    conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False, 'norm_moments_across_images': False,
                 'if_sgu_upsample': True, 'warp_mask_mode': 'robust'}, verbose=False)
    net = conf()
    net.load_state_dict(_weights.make_state_dict(0, head_scale=0.1))
    return net.cuda().to(dtype).eval()


@pytest.mark.parametrize('size', [(1, 375, 1242), (4, 375, 1242), (2, 188, 621)])
@pytest.mark.parametrize('dtype', DT)
def test_native_kitti_frames_pitched_octet_path_vs_the_ragged_path(size, dtype):
    """KITTI's native frame size, the workload of the reference's evaluation (test.py:40-47, batch 1).  Three schedules of the same
    forward:   ragged  = rounds 1-4: contiguous NCHW everywhere, unaligned-row kernels (`_no_pitch`, `_no_c8`);
               pitched = pitched NCHW buffers + octet SGU stacks / context network, flow estimator on NCHW planes (`_no_c8_est`);
               default = pitched + every dense stack of the fine levels on octets (the estimator's K order differs: fp32 sums to
                         summation order, like at W % 8 == 0).
    pitched == ragged BIT FOR BIT (layout changes, not arithmetic); default within the summation-order tolerance."""
    import _weights
    import oracle
    from upflow_pytorch_amd import ops
    from upflow_pytorch_amd.model import pwc_modules
    B, H, W = size
    im1, im2 = _weights.make_smooth_images(9, B, H, W)
    im1, im2 = im1.cuda(), im2.cuda()
    net = _net(dtype)
    with torch.no_grad():
        default = net({'im1': im1, 'im2': im2, 'if_loss': False})
        # bit identity needs the same tile shapes on both sides, as in test_whole_net_c8_levels_are_bit_identical_to_nchw: the
        # 16-channel matrix instruction of the <= 16-channel layers and the 16-row tiles of large grids (which the octet kernels
        # run on 16-channel chunks) sum in another order
        saved = (pwc_modules._NO_NARROW[0], ops.conv_c8_set_option('rpw4', 0), ops.conv_set_option('rpw4_min', 1 << 30), pwc_modules.MERGE_TAIL[0])
        pwc_modules._NO_NARROW[0] = True
        pwc_modules.MERGE_TAIL[0] = False        # (the merged narrow tail of the octet stacks, round 6: another summation order too)
        saved_pairs = pwc_modules.FUSE_PAIRS[0]
        pwc_modules.FUSE_PAIRS[0] = False        # (the fused guidance stem, round 6, reads pixel pairs: it needs an even row pitch, which a contiguous odd-width frame lacks)
        try:
            for m in net.modules():
                m.__dict__.pop('_packed8', None)
            net.__dict__.get('_fast_cache', {}).clear()
            net._no_c8_est = True
            pitched = net({'im1': im1, 'im2': im2, 'if_loss': False})
            net._no_c8_est = False
            net._no_pitch = net._no_c8 = True
            ragged = net({'im1': im1, 'im2': im2, 'if_loss': False})
        finally:
            pwc_modules._NO_NARROW[0] = saved[0]
            pwc_modules.MERGE_TAIL[0] = saved[3]
            pwc_modules.FUSE_PAIRS[0] = saved_pairs
            ops.conv_c8_set_option('rpw4', saved[1])
            ops.conv_set_option('rpw4_min', saved[2])
    for k in ('flow_f_out', 'flow_b_out', 'occ_fw', 'occ_bw'):
        assert torch.isfinite(default[k]).all()
        assert torch.equal(pitched[k], ragged[k]), (k, float((pitched[k] - ragged[k]).abs().max()))
    for k in ('flow_f_out', 'flow_b_out'):
        assert oracle.epe(default[k].cpu(), ragged[k].cpu()) <= 2e-2, (k, float((default[k] - ragged[k]).abs().max()))


def test_native_kitti_frames_take_the_octet_kernels():
    """At 375x1242, batch 4, the two fine levels (94x311, 47x156) and the final up-sampling run their dense stacks on octets —
    the schedule of config 2 (they used to fall back to NCHW everywhere)."""
    import _weights
    net = _net()
    im1, im2 = _weights.make_smooth_images(9, 4, 375, 1242)
    net._taps = []
    with torch.no_grad():
        net({'im1': im1.cuda(), 'im2': im2.cuda(), 'if_loss': False})
    names = [n for n, _ in net._taps]
    assert 'L4.buf8' in names and 'L3.buf8' in names and 'L2.buf' in names, names
    buf8 = dict(net._taps)['L4.buf8']
    assert buf8.dim() == 5 and tuple(buf8.shape[2:]) == (94, 311, 8)
    pair = dict(net._taps)['L4.pair']
    assert pair.stride(-2) == 312, 'the level buffers of a ragged level are row-pitched'


# ----------------------------------------------------------------------------------------------- mixed storage types (`pyramid_dtype`)
@pytest.mark.parametrize('shape', [(2, 32, 47, 311), (2, 64, 24, 80), (2, 128, 12, 39), (2, 196, 6, 20), (8, 32, 96, 320)])
@pytest.mark.parametrize('pitched', [False, True])
def test_corr81_norm_fp16_features_into_bf16_buffers(shape, pitched):
    """UPFlow_net.to_inference(bf16, pyramid_dtype=fp16): the cost volume reads fp16 features and rounds its fp32 sums ONCE to
    bf16 — NCHW and octet output; against the fp16 -> fp16 launch (same sums, rounded to fp16) within one bf16 rounding."""
    from upflow_pytorch_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    pair = (torch.randn(2, B, C, H, W, generator=g) * 1.3).half().cuda()
    ref = ops.corr81_norm_forward_raw(pair[0], pair[1], leaky_slope=0.1).float()
    pp = pitched_copy(pair) if pitched else pair
    out = torch.empty(B, 81, H, W, dtype=torch.bfloat16, device='cuda')
    ops.corr81_norm_forward_raw(pp[0], pp[1], out=out, leaky_slope=0.1)
    tol = 2.0 ** -8 * ref.abs() + 2.0 ** -10 * ref.abs() + 1e-6
    assert bool(((out.float() - ref).abs() <= tol).all()), float((out.float() - ref).abs().max())
    if W % 8 == 0 or pitched:
        out8 = ops.c8_empty(B, 88, H, W, torch.bfloat16, 'cuda')
        ops.corr81_norm_forward_c8(pp[0], pp[1], out8, leaky_slope=0.1)
        flat = ops.from_c8(out8)
        for pos, ch in enumerate(ops.corr81_c8_channel_map()):
            if ch >= 0:
                assert torch.equal(bits(flat[:, pos]), bits(out[:, ch])), (pos, ch)


@pytest.mark.parametrize('case', [(2, 32, 24, 80), (2, 64, 47, 156), (8, 32, 94, 311), (2, 196, 6, 20), (2, 128, 12, 39), (1, 96, 24, 78)])
def test_conv1x1_fp16_features_into_bf16_buffers(case):
    """The 1x1 projection of the fp16 pyramid features into bf16 buffers (NCHW slice and channel octets): fp16 products, fp32 sums,
    ONE rounding to bf16 — against conv2d on the same fp16 operands."""
    from upflow_pytorch_amd import ops
    B, Cin, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).half().cuda()
    w = (torch.randn(32, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).half().cuda()
    b = torch.randn(32, generator=g).cuda()
    want = F.leaky_relu(F.conv2d(x.float(), w.float(), b), 0.1)
    xp = pitched_copy(x)
    packed = ops.conv3x3_pack(w)
    buf = torch.full((B, 40, H, W), 3.0, dtype=torch.bfloat16, device='cuda')
    ops.conv3x3_forward_raw(xp, packed, b, buf[:, 4:36], 1, 0.1, 1, 1)
    tol = 2.0 ** -8 * want.abs() + 1e-3
    assert bool(((buf[:, 4:36].float() - want).abs() <= tol).all()) and bool((buf[:, :4] == 3).all()) and bool((buf[:, 36:] == 3).all())
    y8 = ops.c8_empty(B, 32, H, W, torch.bfloat16, 'cuda')
    ops.conv_c8_forward_raw(None, xp, packed, b, y8, dilation=1, leaky_slope=0.1, kernel_size=1)
    got8 = ops.from_c8(y8)
    assert bool(((got8.float() - want).abs() <= tol).all())
    with pytest.raises(RuntimeError):          # only the 1x1 projection is offered with another output type
        ops.conv3x3_forward_raw(xp, ops.conv3x3_pack((torch.randn(32, Cin, 3, 3) * 0.05).half().cuda()), b, buf[:, 4:36], 1, 0.1, 1, 3)
