"""Split-precision convolution of the fp32 parity mode (csrc/conv_x3.hip: fp32 tensors, fp16 hi / lo operand halves, three
or four matrix-core products per operand pair, fp32 accumulation) against an fp64 convolution of the same fp32 operands — and
against the error torch's own fp32 convolution (MIOpen) makes on them, which is the yardstick: the kernel replaces MIOpen in the
mode whose bar is 1e-4 px end-point error vs the reference (model/pwc_modules.py:122-142, :250-286, :396-412 are fp32 there)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # B, Cin, Cout, H, W, k, dilation, stride
    (1, 32, 32, 8, 32, 3, 1, 1), (2, 115, 128, 24, 40, 3, 1, 1), (1, 243, 128, 16, 64, 3, 1, 1), (1, 371, 96, 9, 24, 3, 1, 1),
    (1, 563, 2, 12, 40, 3, 1, 1), (1, 565, 128, 16, 32, 3, 1, 1), (1, 128, 128, 24, 48, 3, 2, 1), (1, 128, 128, 24, 48, 3, 4, 1),
    (1, 128, 96, 40, 64, 3, 8, 1), (1, 96, 64, 40, 64, 3, 16, 1), (2, 64, 32, 17, 56, 3, 1, 1), (1, 184, 3, 8, 16, 3, 1, 1),
    (4, 96, 32, 96, 320, 3, 1, 1), (8, 115, 128, 48, 160, 3, 1, 1),
    # tiny and ragged images (the coarse pyramid levels of small inputs: MIOpen's territory until round 4), odd channel counts
    (2, 115, 128, 6, 20, 3, 1, 1), (1, 565, 96, 6, 20, 3, 1, 1), (2, 64, 196, 6, 20, 3, 1, 1), (1, 96, 64, 6, 20, 3, 16, 1),
    (1, 40, 33, 7, 13, 3, 1, 1), (1, 35, 2, 5, 9, 3, 1, 1), (1, 7, 5, 3, 8, 3, 1, 1), (1, 5, 3, 1, 1, 3, 1, 1), (2, 9, 4, 2, 3, 3, 2, 1),
    (1, 128, 96, 45, 64, 3, 8, 1), (1, 64, 64, 23, 40, 3, 2, 1), (1, 96, 64, 47, 45, 3, 16, 1), (1, 128, 96, 13, 27, 3, 4, 1),
    # stride 2 (feature pyramid, SGU guidance), 1x1 projections
    (1, 3, 16, 64, 128, 3, 1, 2), (2, 16, 32, 32, 64, 3, 1, 2), (1, 64, 96, 17, 24, 3, 1, 2), (1, 128, 196, 12, 26, 3, 1, 2),
    (1, 32, 64, 13, 27, 3, 1, 2), (1, 16, 16, 5, 7, 3, 1, 2), (1, 3, 16, 384, 1280, 3, 1, 2),
    (2, 32, 32, 24, 40, 1, 1, 1), (1, 196, 32, 6, 20, 1, 1, 1), (1, 128, 32, 12, 26, 1, 1, 1), (1, 16, 32, 7, 13, 1, 1, 1), (1, 64, 200, 3, 5, 1, 1, 1)]


def _run(x, w, b, d, s, k, slope, nprod, off=5, launch=None):
    from upflow_pytorch_amd import ops
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    ho, wo = ops.conv3x3_out_hw(H, W, s)
    xbuf = torch.zeros(B, Cin + off, H, W, device='cuda')
    xbuf[:, off:] = x
    ybuf = torch.full((B, Cout + 3, ho, wo), 7.0, device='cuda')
    xv = xbuf[:, off:]                                # a channel slice: 16-byte aligned rows for some (off, H, W), not for others
    assert ops.conv3x3_supported(xv, Cout, d, s, k)
    packed = ops.conv3x3_pack(w)
    prev = ops.CONV_X3_NPROD[0]
    ops.CONV_X3_NPROD[0] = nprod
    prev_sk = ops.conv_x3_set_option('sk_max_tiles', {'tiled': 0, 'splitk': 1 << 30}[launch]) if launch else None
    try:
        ops.conv3x3_forward_raw(xv, packed, b, ybuf[:, 2:2 + Cout], dilation=d, leaky_slope=slope, stride=s, kernel_size=k)
    finally:
        ops.CONV_X3_NPROD[0] = prev
        if launch:
            ops.conv_x3_set_option('sk_max_tiles', prev_sk)
    assert bool((ybuf[:, :2] == 7).all()) and bool((ybuf[:, 2 + Cout:] == 7).all()), 'wrote outside its channel slice'
    return ybuf[:, 2:2 + Cout].clone()


def test_fp16_matrix_instruction_keeps_subnormal_inputs():
    """The low halves of operands below 2^-3 are fp16 subnormals: the MFMA must multiply them un-flushed."""
    from upflow_pytorch_amd import ops
    assert ops.mfma_f16_denorm_probe(torch.device('cuda', 0))


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('nprod,launch', [(3, 'tiled'), (3, 'splitk'), (11, 'tiled')])
def test_conv_x3_is_fp32_class(case, nprod, launch):
    """`launch`: the 8 x 32-tile kernel everywhere / the split-K kernel wherever it is eligible (stride 1, >= 64 input channels)."""
    B, Cin, Cout, H, W, k, d, s = case
    if launch == 'splitk' and (s != 1 or (Cin + 15) // 16 < 4):
        pytest.skip('not eligible for the split-K kernel: same launch as "tiled"')
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).cuda()
    b = torch.randn(Cout, generator=g).cuda()
    pad = d * (k - 1) // 2
    want = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=pad, dilation=d, stride=s), 0.1)
    got = _run(x, w, b, d, s, k, 0.1, nprod, launch=launch)
    assert got.shape == want.shape and got.dtype == torch.float32
    scale = float(want.abs().max())
    err = float((got.double() - want).abs().max()) / scale
    ref32 = F.leaky_relu(F.conv2d(x, w, b, padding=pad, dilation=d, stride=s), 0.1)
    err32 = float((ref32.double() - want).abs().max()) / scale
    rms = float((got.double() - want).pow(2).mean().sqrt()) / float(want.pow(2).mean().sqrt())
    print('x%d %s %s: max err %.2e of max |y| (torch fp32: %.2e), rms %.2e' % (nprod, launch, case, err, err32, rms))
    # measured on MI355X (all cases): nprod 3: max <= 2.4e-6, rms <= 8.7e-7; nprod 11: max <= 1.5e-6, rms <= 5.4e-7; torch's own fp32
    # convolution (MIOpen) on the same operands: max <= 1.2e-6, rms <= 7.2e-7
    assert err <= (3.0e-6 if nprod == 3 else 2.0e-6), (err, err32)
    assert rms <= (1.1e-6 if nprod == 3 else 7e-7)
    # no activation
    got0 = _run(x, w, b, d, s, k, 0.0, nprod, launch=launch)
    want0 = F.conv2d(x.double(), w.double(), b.double(), padding=pad, dilation=d, stride=s)
    assert float((got0.double() - want0).abs().max()) / float(want0.abs().max()) <= 3.0e-6


@pytest.mark.parametrize('mag', [1e-4, 1e-2, 30.0, 3000.0])
def test_conv_x3_operand_magnitudes(mag):
    """Activations far from O(1).  Weights are rescaled per layer (a power of two in the packed operand), activations are not:
    an activation below 2^-3 has a SUBNORMAL low half (fp16 spacing 2^-24), i.e. an absolute representation error of up to
    2^-25 = 3e-8 whatever its size — fp32-class for O(1) activations, a documented floor for tiny ones (include/upflow_hip.h).
    Large activations (up to fp16's range) and a bias that dwarfs the products keep the relative error fp32-class."""
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(2, 64, 16, 40, generator=g) * mag).cuda()
    w = (torch.randn(48, 64, 3, 3, generator=g) * 0.05).cuda()
    b = torch.randn(48, generator=g).cuda() * mag
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    got = _run(x, w, b, 1, 1, 3, 0.0, 3)
    abs_err = float((got.double() - want).abs().max())
    err = abs_err / float(want.abs().max())
    floor = 2.0 ** -25 * float(w.abs().sum((1, 2, 3)).max())      # every activation off by the subnormal half-spacing, same sign
    print('|x| ~ %g: max err %.2e of max |y| (absolute %.2e, subnormal floor %.2e)' % (mag, err, abs_err, floor))
    assert err <= 2e-6 or abs_err <= floor
    if mag >= 1.0:
        assert err <= 2e-6


@pytest.mark.parametrize('launch', ['tiled', 'splitk'])
def test_conv_x3_is_deterministic_and_alignment_independent(launch):
    """Same bits whatever the alignment path (16-byte loads / element-wise loads) and from run to run, in both kernels."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 72, 12, 24, generator=g).cuda()
    w = (torch.randn(64, 72, 3, 3, generator=g) * 0.05).cuda()
    b = torch.zeros(64).cuda()
    a0 = _run(x, w, b, 1, 1, 3, 0.1, 3, off=4, launch=launch)     # 4*12*24*4 bytes in: aligned
    a1 = _run(x, w, b, 1, 1, 3, 0.1, 3, off=4, launch=launch)
    u = _run(x[:, :, :, :23].contiguous(), w, b, 1, 1, 3, 0.1, 3, off=5, launch=launch)        # W = 23: element-wise loads
    a23 = _run(torch.cat([x[:, :, :, :23], torch.zeros(2, 72, 12, 1, device='cuda')], 3), w, b, 1, 1, 3, 0.1, 3, off=4, launch=launch)
    assert torch.equal(a0, a1)
    assert torch.equal(u[:, :, :, :22], a23[:, :, :, :22])       # (column 22 sees the zero column either way)
    assert torch.equal(u[:, :, :, 22], a23[:, :, :, 22])


def test_dense_stack_in_fp32_buffers_matches_torch():
    """The concat-free dense-stack schedule (pwc_modules._DenseStack.forward_in_buffer) on fp32 buffers through the split-precision
    kernel == the reference's `x = cat([conv(x), x])` chain in torch fp32 (pwc_modules.py:279-286) to fp32 rounding."""
    from upflow_pytorch_amd.model.pwc_modules import FlowEstimatorDense_v2, fp32_conv_mode
    torch.manual_seed(3)
    est = FlowEstimatorDense_v2(115).cuda().eval()
    x = torch.randn(2, 115, 12, 40, device='cuda')
    with torch.no_grad():
        with fp32_conv_mode('miopen'):
            f_ref, o_ref = est(x)
        with fp32_conv_mode('hip_x3'):
            f_hip, o_hip = est(x)
    assert f_hip.shape == f_ref.shape
    assert float((f_hip - f_ref).abs().max()) <= 3e-6 * float(f_ref.abs().max())
    assert float((o_hip - o_ref).abs().max()) <= 3e-6 * float(o_ref.abs().max()) + 1e-6
