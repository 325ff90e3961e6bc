"""P1 — operator-level parity of the HIP kernels (through the C-ABI) against
  (a) the golden vectors generated from the imported reference (tests/golden/*.npz), and
  (b) the CPU oracle on seeded random inputs, incl. ragged / tiny / maximum-displacement shapes.
Tolerances: warp mask bits identical; fp32 values <= 1e-6 relative (a few ulp of fp32); bf16/fp16
I/O compared against the oracle evaluated on the SAME rounded inputs, within output rounding.
"""
import glob
import os

import pytest
import torch

import oracle
from oracle import ops as oops
from conftest import load_golden, unpack_mask, GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    from upflow_pytorch_amd import ops
    return ops


def dev(t):
    return t.cuda()


def relerr(a, b):
    return float((a - b).abs().max() / max(1.0, float(b.abs().max())))


# ---------------------------------------------------------------------------------- correlation
@pytest.mark.parametrize('i', [0, 1, 2, 3])
def test_corr_golden(hip, i):
    g = load_golden('corr_%d' % i)
    out = hip.corr81(dev(g['f1']), dev(g['f2'])).cpu()
    assert (out - g['out']).abs().max() <= 2e-6
    g1, g2 = hip.corr81_backward_raw(dev(g['f1']), dev(g['f2']), dev(g['grad_out']))
    assert (g1.cpu() - g['g1']).abs().max() <= 2e-6
    assert (g2.cpu() - g['g2']).abs().max() <= 2e-6


CORR_SHAPES = [(1, 1, 1, 1), (1, 3, 2, 5), (2, 32, 8, 32), (1, 32, 9, 33), (1, 33, 17, 44), (2, 64, 48, 160),
               (1, 196, 6, 20), (1, 128, 12, 40), (1, 96, 24, 80), (4, 32, 96, 320), (1, 16, 7, 13), (1, 5, 40, 36),
               # the 1/4-resolution and coarse levels of BASELINE configs 3, 4 and 5 (SURVEY.md §8 level table)
               (8, 32, 112, 256), (1, 32, 240, 720), (1, 196, 15, 45), (1, 128, 30, 90), (2, 196, 4, 13), (2, 128, 8, 26),
               (8, 196, 7, 16), (8, 128, 14, 32), (2, 96, 28, 64), (1, 64, 120, 360), (8, 196, 6, 20), (8, 64, 48, 160)]


@pytest.mark.parametrize('shape', CORR_SHAPES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_corr_vs_oracle(hip, shape, dtype):
    B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    f1 = torch.randn(shape, generator=g).to(dtype)
    f2 = torch.randn(shape, generator=g).to(dtype)
    want = oracle.corr81(f1.float(), f2.float())
    got = hip.corr81(dev(f1), dev(f2)).cpu()
    assert got.dtype == dtype and got.shape == (B, 81, H, W)
    if dtype == torch.float32:
        assert (got - want).abs().max() <= 2e-6
    else:
        # exact products, fp32 accumulation: only the output rounding differs
        eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        assert (got.float() - want).abs().max() <= eps * max(1.0, float(want.abs().max())) + 1e-6
    # fused LeakyReLU(0.1)
    got_l = hip.corr81(dev(f1), dev(f2), 0.1).cpu().float()
    want_l = torch.nn.functional.leaky_relu(want, 0.1)
    tol = 2e-6 if dtype == torch.float32 else (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * max(1.0, float(want.abs().max())) + 1e-6
    assert (got_l - want_l).abs().max() <= tol


ALLC_SHAPES = [(1, 1, 1, 1), (2, 5, 3, 3), (1, 3, 5, 4), (1, 9, 3, 5), (2, 32, 24, 40), (1, 32, 9, 33), (2, 64, 48, 160), (1, 96, 24, 80), (1, 128, 12, 40), (2, 196, 6, 20),
               (1, 196, 15, 45), (2, 128, 8, 26), (1, 7, 5, 9), (1, 208, 9, 17), (1, 40, 19, 70), (2, 120, 10, 64)]


@pytest.mark.parametrize('shape', ALLC_SHAPES)
@pytest.mark.parametrize('variant', [0, 1, 2, 3])
def test_corr_every_tile_geometry(hip, shape, variant):
    """Every tile geometry of the all-channels-in-LDS kernel (8x32 / 4x32 / 2x32 / 4x16), forced where C fits, aligned and
    ragged widths, against the oracle; and against the channel-chunked round-1 kernels bit for bit (same products, same
    fp32 accumulation order per pixel: the channel quads are walked in the same order)."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(100 + sum(shape))
    f1 = torch.randn(shape, generator=g).bfloat16()
    f2 = torch.randn(shape, generator=g).bfloat16()
    want = oracle.corr81(f1.float(), f2.float())
    prev = hip.corr_set_option('variant', variant)
    try:
        got = hip.corr81(dev(f1), dev(f2), 0.1).cpu().float()
        got16 = hip.corr81(dev(f1.half()), dev(f2.half())).cpu().float()
    finally:
        hip.corr_set_option('variant', prev)
    want_l = torch.nn.functional.leaky_relu(want, 0.1)
    assert (got - want_l).abs().max() <= 2.0 ** -8 * max(1.0, float(want.abs().max())) + 1e-6
    want16 = oracle.corr81(f1.half().float(), f2.half().float())
    assert (got16 - want16).abs().max() <= 2.0 ** -11 * max(1.0, float(want16.abs().max())) + 1e-6
    hip.corr_set_option('old_path', 1)
    try:
        old = hip.corr81(dev(f1), dev(f2), 0.1).cpu().float()
    finally:
        hip.corr_set_option('old_path', 0)
    assert (got - old).abs().max() <= 2.0 ** -8 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize('shape', ALLC_SHAPES + [(8, 32, 96, 320), (2, 32, 240, 720), (16, 32, 112, 256)])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_corr_with_fused_normalisation_is_bit_identical(hip, shape, dtype):
    """upf_corr81_norm_forward (statistics launch + cost volume whose loader normalises) ==
    upf_normalize_forward on the [f1; f2] pair + upf_corr81_forward, BIT FOR BIT — plain and into a wider buffer with the
    fused LeakyReLU, incl. ragged widths, C not a multiple of 4 and rows split into several statistics segments."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(200 + sum(shape))
    pair = (torch.randn((2,) + shape, generator=g) * 1.7 + 0.3).to(dtype).cuda()
    if W < 4:                                            # rows shorter than a staging quad: not fused, and said so loudly
        assert not hip.corr81_norm_supported(pair[0])
        with pytest.raises(RuntimeError):
            hip.corr81_norm_forward_raw(pair[0], pair[1])
        return
    assert hip.corr81_norm_supported(pair[0])
    normed = hip.normalize(pair.view(2 * B, C, H, W))
    want = hip.corr81_forward_raw(normed[:B], normed[B:])
    got = hip.corr81_norm_forward_raw(pair[0], pair[1])
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    buf = torch.full((B, 115, H, W), 3.0, dtype=dtype, device='cuda')
    hip.corr81_norm_forward_raw(pair[0], pair[1], out=buf[:, :81], leaky_slope=0.1)
    assert torch.equal(buf[:, :81], hip.corr81_forward_raw(normed[:B], normed[B:], leaky_slope=0.1))
    assert bool((buf[:, 81:] == 3.0).all())
    # and against the oracle's normalisation + correlation (fp32 reference arithmetic): within 16-bit rounding
    ref = oracle.corr81(*[t.float() for t in oracle.normalize_pair(pair[0].cpu().float(), pair[1].cpu().float())])
    eps = 2.0 ** -6 if dtype == torch.bfloat16 else 2.0 ** -9
    assert (got.cpu().float() - ref).abs().max() <= eps * max(1.0, float(ref.abs().max()))


def test_corr_norm_rejects_unsupported(hip):
    a = torch.zeros(1, 8, 8, 8, device='cuda')
    assert not hip.corr81_norm_supported(a)
    with pytest.raises(RuntimeError):
        hip.corr81_norm_forward_raw(a, a)                                    # fp32: not fused (parity mode)
    b = torch.zeros(1, 212, 8, 8, device='cuda', dtype=torch.bfloat16)
    assert not hip.corr81_norm_supported(b)
    with pytest.raises(RuntimeError):
        hip.corr81_norm_forward_raw(b, b)
    with pytest.raises(RuntimeError):
        hip.corr_set_option('nope', 1)


def test_corr_into_wider_buffer(hip):
    """out_batch_stride: write straight into the first 81 of 115 channels (model/upflow.py:565)."""
    g = torch.Generator().manual_seed(3)
    f1 = torch.randn(2, 32, 24, 40, generator=g)
    f2 = torch.randn(2, 32, 24, 40, generator=g)
    buf = torch.full((2, 115, 24, 40), 7.0).cuda()
    hip.corr81_forward_raw(dev(f1), dev(f2), out=buf[:, :81])
    assert (buf[:, :81].cpu() - oracle.corr81(f1, f2)).abs().max() <= 2e-6
    assert bool((buf[:, 81:] == 7.0).all())


@pytest.mark.parametrize('shape', [(1, 3, 2, 5), (2, 32, 24, 40), (1, 33, 17, 44), (1, 196, 6, 20), (2, 32, 64, 208),
                                   (2, 32, 112, 256), (1, 32, 240, 720), (1, 196, 15, 45), (1, 128, 30, 90), (2, 196, 4, 13)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_corr_backward_vs_oracle(hip, shape, dtype):
    g = torch.Generator().manual_seed(11 + sum(shape))
    f1 = torch.randn(shape, generator=g).to(dtype)
    f2 = torch.randn(shape, generator=g).to(dtype)
    go = torch.randn(shape[0], 81, shape[2], shape[3], generator=g).to(dtype)
    w1, w2 = oracle.corr81_backward(f1.float(), f2.float(), go.float())
    a = dev(f1).requires_grad_(True)
    b = dev(f2).requires_grad_(True)
    out = hip.corr81(a, b)
    g1, g2 = torch.autograd.grad(out, (a, b), dev(go))
    tol = 5e-6 if dtype == torch.float32 else (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * max(1.0, float(w1.abs().max()))
    assert (g1.cpu().float() - w1).abs().max() <= tol
    assert (g2.cpu().float() - w2).abs().max() <= tol


@pytest.mark.parametrize('shape', [(1, 8, 12, 16), (1, 8, 14, 68), (2, 4, 11, 7)])
def test_corr_backward_border_is_safe_against_nonfinite_neighbours(hip, shape):
    """The tiled backward fetches 4-wide vectors of grad_out at displaced columns; where such a vector leaves the row it
    covers elements of the neighbouring row.  Those terms are dropped by the reference (correlation_cuda_kernel.cu:228-262),
    so an Inf / NaN elsewhere in grad_out must not leak into border pixels: poison one whole grad_out row and
    compare every OTHER affected pixel with the oracle evaluated on the same input."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    f1, f2 = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    go = torch.randn(B, 81, H, W, generator=g)
    go[:, :, H // 2, :] = float('inf')
    w1, w2 = oracle.corr81_backward(f1, f2, go)
    g1, g2 = hip.corr81_backward_raw(dev(f1), dev(f2), dev(go))
    for got, want in ((g1.cpu(), w1), (g2.cpu(), w2)):
        fin = torch.isfinite(want)
        assert torch.equal(torch.isfinite(got), fin), 'non-finite values leaked into %d pixels' % int((torch.isfinite(got) != fin).sum())
        assert fin.any() and (got[fin] - want[fin]).abs().max() <= 1e-5


GENERAL_SETS = [(4, 1, 4, 1, 1), (2, 1, 2, 1, 1), (3, 1, 4, 1, 2), (0, 1, 1, 1, 1), (5, 1, 3, 1, 3), (20, 1, 20, 1, 2),   # backward defined
                (3, 3, 3, 1, 2), (4, 1, 4, 2, 2), (5, 3, 5, 2, 3), (6, 5, 5, 1, 3)]                                     # forward only


def test_corr_general_parameters(hip):
    """upf_correlation_forward / upf_correlation_backward with the reference's full parameter list against the oracle's
    restatement, which round 4 PINS on a literal thread-by-thread emulation of correlation_cuda_kernel.cu:41-114 / :116-300
    and on hand-computed vectors (tests/test_oracle_golden.py::test_correlation_general_*).  Forward: every parameter set for
    which the reference's kernel stays inside its padded buffers; backward: kernel_size 1, stride1 1."""
    g = torch.Generator().manual_seed(21)
    f1 = torch.randn(2, 6, 20, 28, generator=g)
    f2 = torch.randn(2, 6, 20, 28, generator=g)
    for (pad, k, md, s1, s2) in GENERAL_SETS:
        want = oops.correlation_general(f1, f2, pad, k, md, s1, s2)
        got = hip.correlation_forward_general(dev(f1), dev(f2), pad, k, md, s1, s2).cpu()
        assert got.shape == want.shape, (pad, k, md, s1, s2)
        assert (got - want).abs().max() <= 5e-6, (pad, k, md, s1, s2)
        go = torch.randn(want.shape, generator=g)
        if oops.correlation_backward_supported(pad, k, md, s1, s2):
            w1, w2 = oops.correlation_general_backward(f1, f2, go, pad, k, md, s1, s2)
            g1, g2 = hip.correlation_backward_general(dev(f1), dev(f2), dev(go), pad, k, md, s1, s2)
            assert (g1.cpu() - w1).abs().max() <= 5e-6 and (g2.cpu() - w2).abs().max() <= 5e-6, (pad, k, md, s1, s2)
        else:
            with pytest.raises(hip.UpflowHipError, match='kernel_size'):
                hip.correlation_backward_general(dev(f1), dev(f2), dev(go), pad, k, md, s1, s2)
    # where the reference reads outside its padded buffers (undefined) the entry point refuses instead of inventing values
    for (pad, k, md, s1, s2) in [(3, 3, 2, 1, 1), (2, 3, 2, 2, 1), (4, 5, 4, 1, 3)]:
        assert not oops.correlation_well_defined(pad, k, md, s1, s2)
        with pytest.raises(hip.UpflowHipError, match='outside its padded buffer'):
            hip.correlation_forward_general(dev(f1), dev(f2), pad, k, md, s1, s2)


def test_corr_general_against_the_literal_emulation_and_the_module(hip):
    """The HIP general kernels straight against the scalar emulation of the reference's CUDA kernels (small case), in 16-bit too,
    and the `Correlation` module (the reference's constructor, correlation.py:47-61) under autograd for a non-81 parameter set."""
    from upflow_pytorch_amd.model.correlation_package.correlation import Correlation
    g = torch.Generator().manual_seed(22)
    f1 = torch.randn(1, 5, 6, 7, generator=g)
    f2 = torch.randn(1, 5, 6, 7, generator=g)
    for (pad, k, md, s1, s2) in [(3, 1, 4, 1, 2), (2, 3, 3, 1, 2), (4, 1, 4, 2, 1)]:
        lit = oops.correlation_forward_literal(f1, f2, pad, k, md, s1, s2)
        got = hip.correlation_forward_general(dev(f1), dev(f2), pad, k, md, s1, s2).cpu()
        assert (got - lit).abs().max() <= 2e-6, (pad, k, md, s1, s2)
        for dt, tol in ((torch.bfloat16, 2e-2), (torch.float16, 3e-3)):
            got16 = hip.correlation_forward_general(dev(f1).to(dt), dev(f2).to(dt), pad, k, md, s1, s2).float().cpu()
            assert (got16 - oops.correlation_forward_literal(f1.to(dt).float(), f2.to(dt).float(), pad, k, md, s1, s2)).abs().max() <= tol
    pad, k, md, s1, s2 = 3, 1, 4, 1, 2
    go = torch.randn(1, 25, 4, 5, generator=g)
    l1, l2 = oops.correlation_backward_literal(f1, f2, go, pad, k, md, s1, s2)
    a, b = dev(f1).requires_grad_(True), dev(f2).requires_grad_(True)
    out = Correlation(pad_size=pad, kernel_size=k, max_displacement=md, stride1=s1, stride2=s2, corr_multiply=1)(a, b)
    g1, g2 = torch.autograd.grad(out, (a, b), dev(go))
    assert (g1.cpu() - l1).abs().max() <= 2e-6 and (g2.cpu() - l2).abs().max() <= 2e-6


# ------------------------------------------------------------------------------------------ warp
WARP_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'warp_*.npz')))


@pytest.mark.parametrize('name', WARP_CASES)
def test_warp_golden(hip, name):
    g = load_golden(name)
    x, flow = g['x'], g['flow']
    B, C, H, W = x.shape
    xd = dev(x).requires_grad_(True)
    fd = dev(flow).requires_grad_(True)
    y = hip.warp(xd, fd, 'literal')
    # mask bits: warp a ones tensor, exactly how the fixture extracted them from the reference
    mask = hip.warp(torch.ones(B, 1, H, W).cuda(), dev(flow), 'literal').cpu() > 0
    ref_mask = unpack_mask(g['mask'], (B, 1, H, W))
    assert torch.equal(mask, ref_mask), 'mask bits differ at %d pixels' % int((mask != ref_mask).sum())
    assert relerr(y.detach().cpu(), g['y']) <= 1e-6
    gx, gf = torch.autograd.grad(y, (xd, fd), dev(g['grad_out']))
    assert (gx.cpu() - g['gx']).abs().max() <= 1e-5
    assert relerr(gf.cpu(), g['gflow']) <= 1e-4
    x2 = dev(x).requires_grad_(True)
    f2 = dev(flow).requires_grad_(True)
    y2 = hip.warp(x2, f2, None)
    assert relerr(y2.detach().cpu(), g['y_nomask']) <= 1e-6
    gx2, gf2 = torch.autograd.grad(y2, (x2, f2), dev(g['grad_out']))
    assert (gx2.cpu() - g['gx_nomask']).abs().max() <= 1e-5
    assert relerr(gf2.cpu(), g['gflow_nomask']) <= 1e-4


@pytest.mark.parametrize('shape', [(1, 1, 1, 1), (2, 3, 5, 7), (1, 32, 96, 320), (2, 128, 12, 40), (1, 7, 33, 65),
                                   (2, 32, 112, 256), (1, 32, 240, 720), (1, 128, 30, 90), (1, 96, 60, 180), (2, 128, 8, 26)])
@pytest.mark.parametrize('mode', ['literal', 'robust', None])
def test_warp_vs_oracle(hip, shape, mode):
    B, C, H, W = shape
    g = torch.Generator().manual_seed(31 + sum(shape))
    x = torch.randn(shape, generator=g)
    flow = torch.randn(B, 2, H, W, generator=g) * 4
    want = oracle.warp(x, flow, mode)
    got = hip.warp(dev(x), dev(flow), mode).cpu()
    assert torch.equal(got == 0, want == 0) or mode is None, 'mask pattern differs'
    assert relerr(got, want) <= 1e-6
    for dt in (torch.bfloat16, torch.float16):
        xq = x.to(dt)
        want_q = oracle.warp(xq.float(), flow, mode)
        got_q = hip.warp(dev(xq), dev(flow), mode).cpu().float()
        eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        assert (got_q - want_q).abs().max() <= eps * max(1.0, float(want_q.abs().max()))


def test_warp_nan_inf_flow_is_safe(hip):
    x = torch.ones(1, 2, 8, 8).cuda()
    flow = torch.zeros(1, 2, 8, 8)
    flow[0, 0, 0, 0] = float('nan')
    flow[0, 1, 1, 1] = float('inf')
    flow[0, 0, 2, 2] = -1e30
    y = hip.warp(x, flow.cuda(), 'literal').cpu()
    assert y[0, :, 0, 0].abs().sum() == 0 and y[0, :, 1, 1].abs().sum() == 0 and y[0, :, 2, 2].abs().sum() == 0
    assert bool((y[0, :, 4:, 4:] == 1).all())


def test_scatter_gradients_surface_overflow_and_nan(hip):
    """the fixed-point scatter accumulators (warp / SGU-blend backward) used to clamp a contribution at
    +-2.3e5 and map NaN to 0 — a diverging step produced finite, wrong gradients that no overflow check could see.  Now a
    non-finite or oversized contribution poisons exactly the elements it reaches (NaN out), everything else is untouched."""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(1, 3, 12, 16, generator=g).cuda().requires_grad_(True)
    flow = (torch.randn(1, 2, 12, 16, generator=g) * 0.4).cuda().requires_grad_(True)
    go = torch.randn(1, 3, 12, 16, generator=g).cuda()
    y = hip.warp(x, flow, 'robust')
    gx0, gf0 = torch.autograd.grad(y, (x, flow), go, retain_graph=True)
    assert torch.isfinite(gx0).all() and torch.isfinite(gf0).all()
    bad = go.clone()
    bad[0, 0, 5, 7] = float('nan')
    bad[0, 1, 2, 3] = 1e30
    bad[0, 2, 9, 9] = float('inf')
    gx1, gf1 = torch.autograd.grad(y, (x, flow), bad)
    nf = ~torch.isfinite(gx1)
    assert 3 <= int(nf.sum()) <= 12                               # each bad output pixel reaches <= 4 source pixels of its channel
    assert not nf[0, 0, :3].any() and nf[0, 0, 4:7, 6:9].any() and nf[0, 1, 1:4, 2:5].any() and nf[0, 2, 8:11, 8:11].any()
    assert torch.equal(gx1[~nf], gx0[~nf])                        # bit for bit elsewhere (integer accumulation)
    # (the flow gradient of a pixel is a plain fp32 sum in this launch form: NaN / inf propagate, 1e30 stays a huge finite number)
    assert not torch.isfinite(gf1[0, :, 5, 7]).any() and not torch.isfinite(gf1[0, :, 9, 9]).any() and float(gf1[0, :, 2, 3].abs().max()) > 1e20
    ok = torch.ones_like(gf1, dtype=torch.bool)
    ok[0, :, 5, 7] = ok[0, :, 9, 9] = ok[0, :, 2, 3] = False
    assert torch.equal(gf1[ok], gf0[ok])
    # an honest sum beyond +-2^18 is flagged too instead of wrapping around silently
    big = torch.full_like(go, 3.0e4)
    gx2, = torch.autograd.grad(hip.warp(x, torch.zeros_like(flow), 'robust'), x, big)
    assert torch.allclose(gx2, big)                               # identity warp: one contribution of 3e4 per element, fine
    x1 = torch.ones(1, 1, 4, 16, device='cuda', requires_grad=True)
    f1 = torch.zeros(1, 2, 4, 16, device='cuda')
    f1[0, 0] = -torch.arange(16, device='cuda', dtype=torch.float32).view(1, 16)        # every pixel of a row samples column 0
    g1, = torch.autograd.grad(hip.warp(x1, f1, 'none'), x1, torch.full((1, 1, 4, 16), 3.0e4, device='cuda'))
    assert not torch.isfinite(g1[0, 0, :, 0]).any()               # 16 x 3e4 = 4.8e5 > 2^18: NaN, not a silently wrong finite number
    assert torch.isfinite(g1[0, 0, :, 1:]).all()


# ---------------------------------------------------------------------------------- flow upsample
@pytest.mark.parametrize('i', range(6))
def test_flow_upsample_golden(hip, i):
    g = load_golden('upsample_%d' % i)
    h, w = [int(v) for v in g['size']]
    xd = dev(g['x']).requires_grad_(True)
    y = hip.flow_upsample(xd, h, w, True)
    assert relerr(y.detach().cpu(), g['y']) <= 2e-6
    assert relerr(hip.flow_upsample(dev(g['x']), h, w, False).cpu(), g['y_norate']) <= 2e-6
    gx, = torch.autograd.grad(y, xd, dev(g['grad_out']))
    assert relerr(gx.cpu(), g['gx']) <= 1e-5


# -------------------------------------------------------------------------------------- SGU blend
@pytest.mark.parametrize('i', range(4))
def test_sgu_blend_golden(hip, i):
    g = load_golden('sgu_blend_%d' % i)
    olf = g.get('output_level_flow')
    xo = dev(g['x_out']).requires_grad_(True)
    fi = dev(g['flow_init']).requires_grad_(True)
    od = dev(olf).requires_grad_(True) if olf is not None else None
    _, up, inter_flow, inter_mask = hip.sgu_blend(fi, xo, od)
    tol = 2e-6 * max(1.0, float(g['flow_up'].abs().max()))
    assert (up.detach().cpu() - g['flow_up']).abs().max() <= (tol if olf is None else 5e-5)
    assert (inter_flow.cpu() - g['inter_flow']).abs().max() <= tol
    assert (inter_mask.cpu() - g['inter_mask']).abs().max() <= 1e-6
    if olf is None:
        gxo, gfi = torch.autograd.grad(up, (xo, fi), dev(g['grad_out']))
        assert relerr(gfi.cpu(), g['g_flow_init']) <= 1e-4
    else:
        gxo, go = torch.autograd.grad(up, (xo, od), dev(g['grad_out']))
        assert relerr(go.cpu(), g['g_output_level_flow']) <= 1e-4
    assert relerr(gxo.cpu(), g['g_x_out']) <= 1e-4


# -------------------------------------------------------------------------------------- normalize
@pytest.mark.parametrize('i', range(3))
def test_normalize_golden(hip, i):
    g = load_golden('normalize_%d' % i)
    a = dev(g['a']).requires_grad_(True)
    b = dev(g['b']).requires_grad_(True)
    na, nb = hip.normalize_pair(a, b)
    assert relerr(na.detach().cpu(), g['na']) <= 2e-6
    assert relerr(nb.detach().cpu(), g['nb']) <= 2e-6
    ga, gb = torch.autograd.grad([na, nb], [a, b], [dev(g['goa']), dev(g['gob'])])
    assert relerr(ga.cpu(), g['ga']) <= 1e-4
    assert relerr(gb.cpu(), g['gb']) <= 1e-4


@pytest.mark.parametrize('shape', [(8, 32, 112, 256), (1, 32, 240, 720), (1, 196, 15, 45), (1, 128, 30, 90), (2, 196, 4, 13), (1, 3, 1, 2)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_normalize_vs_oracle_config_shapes(hip, shape, dtype):
    """The level shapes of BASELINE configs 3-5 (ragged 15x45 / 30x90 / 4x13 rows included), values and gradients."""
    g = torch.Generator().manual_seed(41 + sum(shape))
    x = (torch.randn(shape, generator=g) * 2 + 0.5).to(dtype)
    go = torch.randn(shape, generator=g).to(dtype)
    xr = x.float().requires_grad_(True)
    want = oracle.normalize_pair(xr, xr)[0]
    (gwant,) = torch.autograd.grad(want, xr, go.float())
    xd = dev(x).requires_grad_(True)
    got = hip.normalize(xd)
    (ggot,) = torch.autograd.grad(got, xd, dev(go))
    eps = {torch.float32: 3e-6, torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dtype]
    assert (got.detach().cpu().float() - want.detach()).abs().max() <= eps * max(1.0, float(want.abs().max()))
    geps = {torch.float32: 1e-4, torch.bfloat16: 2.0 ** -6, torch.float16: 2.0 ** -9}[dtype]
    assert (ggot.cpu().float() - gwant).abs().max() <= geps * max(1.0, float(gwant.abs().max()))


def test_normalize_bf16(hip):
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(2, 32, 96, 320, generator=g) * 3 + 1).to(torch.bfloat16)
    want = oracle.normalize_pair(x.float(), x.float())[0]
    got = hip.normalize(dev(x)).cpu().float()
    assert (got - want).abs().max() <= 2.0 ** -7 * float(want.abs().max())


# -------------------------------------------------------------------------------------- occlusion
@pytest.mark.parametrize('i', range(2))
def test_occ_golden(hip, i):
    g = load_golden('occ_%d' % i)
    o1, o2 = hip.occ_check(dev(g['flow_f']), dev(g['flow_b']))
    assert (o1.cpu() != g['occ_fw']).float().mean() <= 1e-4
    assert (o2.cpu() != g['occ_bw']).float().mean() <= 1e-4


# ---------------------------------------------------------------------------------- error behaviour
def test_errors_are_loud(hip):
    a = torch.zeros(1, 4, 8, 8).cuda()
    with pytest.raises(RuntimeError):
        hip.corr81(a, torch.zeros(1, 4, 8, 9).cuda())
    with pytest.raises(RuntimeError):
        hip.corr81(a, a.double())
    with pytest.raises(RuntimeError):
        hip.warp(a, torch.zeros(1, 2, 8, 9).cuda())
    with pytest.raises(RuntimeError):
        hip.correlation_forward_general(a, a, 0, 1, 4, 1, 1)      # empty output (H+0-8 <= 0)


def test_forward_kernels_are_bit_deterministic(hip):
    """No atomics on any forward path: two launches on the same bits give the same bits."""
    g = torch.Generator().manual_seed(9)
    f1 = torch.randn(2, 64, 48, 160, generator=g).cuda().bfloat16()
    f2 = torch.randn(2, 64, 48, 160, generator=g).cuda().bfloat16()
    flow = (torch.randn(2, 2, 48, 160, generator=g) * 3).cuda()
    xo = torch.randn(2, 3, 48, 160, generator=g).cuda()
    for fn in (lambda: hip.corr81(f1, f2, 0.1), lambda: hip.warp(f1, flow, 'literal'), lambda: hip.normalize(f1),
               lambda: hip.sgu_blend(flow, xo)[1], lambda: hip.flow_upsample(flow, 96, 320, True),
               lambda: hip.occ_check(flow, -flow)[0]):
        assert torch.equal(fn(), fn())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(4, 8, 12, 40), (2, 5, 7, 13), (2, 3, 9, 1)])
def test_warp_into_channel_slices_equals_warp(dtype, shape):
    """upf_warp_forward_strided: x / y as channel slices of wider buffers == the plain warp, bit for bit."""
    from upflow_pytorch_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    xbuf = torch.randn(B, C + 3, H, W, generator=g).to(dtype).cuda()
    flow = (torch.randn(B, 2, H, W, generator=g) * 3).cuda()
    ybuf = torch.full((B, 2 * C + 1, H, W), 5.0, dtype=dtype, device='cuda')
    for mode in ('literal', 'robust', 'none'):
        for shift in (0, B // 2):
            want = ops.warp(xbuf[:, 2:2 + C].contiguous(), flow, mode, shift)
            got = ops.warp_into(xbuf[:, 2:2 + C], flow, ybuf[:, C:2 * C], mode, shift)
            assert torch.equal(got, want)
            assert bool((ybuf[:, :C] == 5).all()) and bool((ybuf[:, 2 * C:] == 5).all()), 'wrote outside its slice'


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_flow_update_matches_torch_chain(dtype):
    """upf_flow_update == the convert / add / convert chain of torch ops it replaces (model/upflow.py:566-572)."""
    from upflow_pytorch_amd import ops
    g = torch.Generator().manual_seed(7)
    a = torch.randn(6, 2, 12, 40, generator=g).cuda() * 4
    b = torch.randn(6, 2, 12, 40, generator=g).to(dtype).cuda()
    c = torch.randn(6, 2, 12, 40, generator=g).to(dtype).cuda()
    assert torch.equal(ops.flow_update(a, b, c), a + (b.float() + c.float()))
    assert torch.equal(ops.flow_update(a, b), a + b.float())
    buf = torch.full((6, 7, 12, 40), 9.0, dtype=dtype, device='cuda')
    ops.flow_update(a, b, out=buf[:, 3:5])
    assert torch.equal(buf[:, 3:5], (a + b.float()).to(dtype))
    ops.flow_update(a, out=buf[:, 5:7])
    assert torch.equal(buf[:, 5:7], a.to(dtype)) and bool((buf[:, :3] == 9).all())
    with pytest.raises(RuntimeError):
        ops.flow_update(a.cpu(), b.cpu())


@pytest.mark.parametrize('shape', [((6, 20), (12, 40)), ((4, 13), (256, 832)), ((64, 208), (256, 832)), ((1, 3), (7, 9)),
                                   ((5, 1), (11, 6)), ((24, 40), (9, 15)), ((3, 5), (3, 5)),
                                   # 8x (16 lanes per input pixel), 16x (workgroup variant with weight tables), footprints wider than the
                                   # group kernel's column table / taller than the workgroup kernel's tables (their fallback loops)
                                   ((32, 104), (256, 832)), ((16, 52), (256, 832)), ((6, 5), (6, 200)), ((2, 3), (400, 12))])
@pytest.mark.parametrize('if_rate', [True, False])
def test_flow_upsample_backward_matches_autograd(shape, if_rate):
    """upf_flow_upsample_backward (deterministic gather, both kernel variants) vs autograd through
    F.interpolate(bilinear, align_corners=True) * size ratio (model/pwc_modules.py:77-90)."""
    import torch.nn.functional as F
    from upflow_pytorch_amd import ops
    (h, w), (H, W) = shape
    g = torch.Generator().manual_seed(h * 1000 + W)
    x = torch.randn(3, 2, h, w, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(3, 2, H, W, generator=g).cuda()
    y = ops.flow_upsample(x, H, W, if_rate)
    (gx,) = torch.autograd.grad(y, x, gy)
    xr = x.detach().clone().requires_grad_(True)
    yr = F.interpolate(xr, size=(H, W), mode='bilinear', align_corners=True)
    if if_rate:
        yr = yr * torch.tensor([W / w, H / h], device='cuda').view(1, 2, 1, 1)
    (gr,) = torch.autograd.grad(yr, xr, gy)
    assert relerr(y.detach().cpu(), yr.detach().cpu()) <= 2e-6
    assert relerr(gx.cpu(), gr.cpu()) <= 1e-5, float((gx - gr).abs().max())
    (gx2,) = torch.autograd.grad(ops.flow_upsample(x, H, W, if_rate), x, gy)
    assert torch.equal(gx, gx2), 'backward must be deterministic'


@pytest.mark.parametrize('shape', [((256, 832), [(4, 13), (8, 26), (16, 52), (32, 104), (64, 208)]), ((37, 53), [(5, 7), (19, 27), (37, 53)]), ((64, 96), [(16, 24)])])
@pytest.mark.parametrize('use_occ', [True, False])
def test_fused_distillation_term_matches_the_composition(shape, use_occ):
    """upf_msd_upup_forward / _backward (one direction of the 'upup' pyramid-distillation term, model/upflow.py:461-487) == the
    composition it replaces: flow_upsample -> abs_robust sums -> s / (s_occ + 1e-6), summed over the levels, and its autograd
    gradients with respect to every level flow; deterministic."""
    from upflow_pytorch_amd import ops
    (H, W), levels = shape
    B = 2
    g = torch.Generator().manual_seed(H * 7 + W)
    y = (torch.randn(B, 2, H, W, generator=g) * 3).cuda()
    occ = (torch.rand(B, 1, H, W, generator=g) > 0.3).float().cuda() if use_occ else None
    xs = [(torch.randn(B, 2, h, w, generator=g) * 0.5).cuda().requires_grad_(True) for h, w in levels]
    weight = 0.7
    assert ops.msd_upup_supported(y, xs)
    got = ops.msd_upup_loss(xs, y, occ, weight)
    ggot = torch.autograd.grad(got, xs)
    ref = 0
    for x in xs:
        s, s_occ = ops.robust_loss_sums(ops.flow_upsample(x, H, W, True), y, occ, q=0.4, eps=0.01)
        ref = ref + (s / (s_occ + 1e-6) if use_occ else s / float(y.numel()))
    ref = weight * ref
    gref = torch.autograd.grad(ref, xs)
    assert abs(float(got) - float(ref)) <= 2e-5 * max(1.0, abs(float(ref))), (float(got), float(ref))
    for a, b in zip(ggot, gref):
        assert relerr(a.cpu(), b.cpu()) <= 2e-4, float((a - b).abs().max())
    assert float(got) == float(ops.msd_upup_loss(xs, y, occ, weight))
    again = torch.autograd.grad(ops.msd_upup_loss(xs, y, occ, weight), xs)
    assert all(torch.equal(a, b) for a, b in zip(ggot, again))


def test_grey_is_the_reference_expression_bit_for_bit():
    from upflow_pytorch_amd import ops
    img = torch.randn(2, 3, 33, 47, generator=torch.Generator().manual_seed(5)).cuda()
    r, g, b = img[:, 0:1], img[:, 1:2], img[:, 2:3]
    assert torch.equal(ops.grey(img), 0.2989 * r + 0.5870 * g + 0.1140 * b)


def test_new_entry_points_reject_bad_arguments():
    """upf_warp_forward_strided / upf_flow_update / upf_conv_set_option fail loudly (RuntimeError) on misuse."""
    from upflow_pytorch_amd import ops, _lib
    x = torch.zeros(2, 4, 8, 16, dtype=torch.bfloat16, device='cuda')
    flow = torch.zeros(2, 2, 8, 16, device='cuda')
    with pytest.raises(RuntimeError):
        ops.warp_into(x, flow, torch.zeros(2, 4, 8, 8, dtype=torch.bfloat16, device='cuda'))          # shape mismatch
    with pytest.raises(RuntimeError):
        ops.warp_into(x.permute(0, 1, 3, 2), flow, x.clone())                                          # not a channel slice
    with pytest.raises(RuntimeError):
        ops.warp_into(x, flow, x.float())                                                             # dtype mismatch
    with pytest.raises(RuntimeError):                                                                 # y batch stride < C*H*W
        _lib.call('upf_warp_forward_strided', _lib.ptr(x), 0, _lib.ptr(flow), _lib.ptr(x), 7, 2, 4, 8, 16, _lib.UPF_BF16, 1, 0,
                  _lib.stream_ptr(x.device))
    a = torch.zeros(2, 2, 8, 16, device='cuda')
    with pytest.raises(RuntimeError):
        ops.flow_update(a, None, a.bfloat16())                                                        # c without b
    with pytest.raises(RuntimeError):
        ops.flow_update(a, a.bfloat16(), out=torch.zeros(2, 2, 8, 16, dtype=torch.float16, device='cuda'))   # mixed 16-bit types
    with pytest.raises(RuntimeError):
        ops.conv_set_option('no_such_option', 1)


@pytest.mark.parametrize('shape', [(2, 3, 24, 40), (1, 3, 7, 9), (1, 3, 64, 208)])
def test_census_distance_matches_reference_spelling(shape):
    """upf_census_forward / _backward vs the reference's 49-channel conv2d formulation (oracle.census_distance,
    utils/loss.py:52-67) and its autograd gradient."""
    from upflow_pytorch_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    im1 = torch.rand(shape, generator=g) - 0.45
    im2 = (im1 + 0.1 * torch.randn(shape, generator=g)).requires_grad_(True)
    want = oracle.census_distance(im1, im2)
    gout = torch.randn(want.shape, generator=g)
    (gwant,) = torch.autograd.grad(want, im2, gout)

    def grey(t):
        r, gg, b = torch.split(t, 1, 1)
        return 0.2989 * r + 0.5870 * gg + 0.1140 * b
    a = im1.cuda()
    b = im2.detach().cuda().requires_grad_(True)
    got = ops.census_distance(grey(a), grey(b))
    (ggot,) = torch.autograd.grad(got, b, gout.cuda())
    assert relerr(got.detach().cpu(), want.detach()) <= 2e-6
    assert relerr(ggot.cpu(), gwant) <= 2e-5, float((ggot.cpu() - gwant).abs().max())
    (ggot2,) = torch.autograd.grad(ops.census_distance(grey(a), grey(b)), b, gout.cuda())
    assert torch.equal(ggot, ggot2)
    with pytest.raises(RuntimeError):
        ops.census_distance(grey(im1), grey(im1))          # CPU tensors are rejected


@pytest.mark.parametrize('i', [0, 1])
def test_census_loss_golden(i):
    """The package's census loss (fused HIP distance + the reference's reduction) against the reference's own scalars and
    gradients wrt the warped image (tests/golden/census_*.npz, generated by importing utils/loss.py)."""
    from upflow_pytorch_amd.utils.loss import loss_functions
    g = load_golden('census_%d' % i)
    im1 = dev(g['img1'])
    for k in range(3):
        w = dev(g['img1_warp']).requires_grad_(True)
        v = loss_functions.census_loss_torch(img1=im1, img1_warp=w, mask=dev(g['masks'][k]), q=0.4, charbonnier_or_abs_robust=False,
                                             if_use_occ=True, averge=True)
        (gw,) = torch.autograd.grad(v, w)
        assert abs(float(v) - float(g['loss_occ_%d' % k])) <= 2e-6 * max(1.0, abs(float(v)))
        assert (gw.cpu() - g['grad_occ_%d' % k]).abs().max() <= 2e-6
    v = loss_functions.census_loss_torch(img1=im1, img1_warp=dev(g['img1_warp']), mask=dev(g['masks'][0]), q=0.4,
                                         charbonnier_or_abs_robust=False, if_use_occ=False, averge=True)
    assert abs(float(v) - float(g['loss_mean'])) <= 2e-6


# ------------------------------------------------------------------- the reference's literal FFI (legacy shim)
@pytest.mark.parametrize('i', [0, 1, 2, 3])
def test_legacy_correlation_cuda_ffi_golden(i):
    """`correlation_cuda.forward / backward` with the pybind signatures of correlation_cuda.cc:10-17, :89-97 and the
    reference's ownership rules (the caller passes EMPTY tensors, `input1.new()`, correlation.py:22-24,35-39; the callee
    resize_s them and writes in place, returns 1) against the reference's vectors."""
    from upflow_pytorch_amd import correlation_cuda
    g = load_golden('corr_%d' % i)
    a, b = dev(g['f1']), dev(g['f2'])
    r1, r2, out = a.new(), a.new(), a.new()
    assert out.numel() == 0
    assert correlation_cuda.forward(a, b, r1, r2, out, 4, 1, 4, 1, 1, 1) == 1
    assert out.shape == g['out'].shape and out.device == a.device
    assert (out.cpu() - g['out']).abs().max() <= 2e-6
    g1, g2 = a.new(), a.new()
    assert correlation_cuda.backward(a, b, r1, r2, dev(g['grad_out']), g1, g2, 4, 1, 4, 1, 1, 1) == 1
    assert g1.shape == a.shape and g2.shape == b.shape
    assert (g1.cpu() - g['g1']).abs().max() <= 2e-6 and (g2.cpu() - g['g2']).abs().max() <= 2e-6
    # a pre-sized, stale output tensor is resized and fully overwritten too (correlation_cuda.cc:36-42 resize_ + fill_)
    stale = torch.full((3, 5), 9.0, device=a.device)
    assert correlation_cuda.forward(a, b, r1, r2, stale, 4, 1, 4, 1, 1, 1) == 1
    assert torch.equal(stale, out)
    # the outputs are written IN PLACE into the caller's storage (correlation_cuda.cc:36-42), not into a temporary + copy
    pre = torch.full(tuple(g['out'].shape), 7.0, device=a.device)
    ptr = pre.data_ptr()
    assert correlation_cuda.forward(a, b, r1, r2, pre, 4, 1, 4, 1, 1, 1) == 1 and pre.data_ptr() == ptr and torch.equal(pre, out)
    # other parameter sets: the general kernels, forward and (kernel_size 1, stride1 1) backward; elsewhere backward raises
    other = a.new()
    assert correlation_cuda.forward(a, b, r1, r2, other, 2, 1, 2, 1, 1, 1) == 1
    assert (other.cpu() - oops.correlation_general(g['f1'], g['f2'], 2, 1, 2, 1, 1)).abs().max() <= 5e-6
    go2 = torch.randn(other.shape, generator=torch.Generator().manual_seed(5))
    w1, w2 = oops.correlation_general_backward(g['f1'], g['f2'], go2, 2, 1, 2, 1, 1)
    assert correlation_cuda.backward(a, b, r1, r2, dev(go2), g1, g2, 2, 1, 2, 1, 1, 1) == 1
    assert (g1.cpu() - w1).abs().max() <= 5e-6 and (g2.cpu() - w2).abs().max() <= 5e-6
    with pytest.raises(RuntimeError):
        correlation_cuda.backward(a, b, r1, r2, dev(torch.zeros(1, 25, 1, 1)), g1, g2, 2, 1, 2, 2, 1, 1)      # stride1 2
    with pytest.raises(RuntimeError):
        correlation_cuda.forward(g['f1'], g['f2'], r1, r2, out, 4, 1, 4, 1, 1, 1)          # CPU tensors: no fallback


def test_reference_correlation_function_body_runs_on_the_top_level_module():
    """The body of the reference's CorrelationFunction.forward / backward (model/correlation_package/correlation.py:17-44: empty
    `input1.new()` tensors, `with torch.cuda.device_of(input1)`, the 11 / 13 positional arguments) executed against the module
    that `import correlation_cuda` — the reference's own line 4 — resolves to."""
    import importlib
    import sys
    sys.modules.pop('correlation_cuda', None)
    correlation_cuda = importlib.import_module('correlation_cuda')
    g = load_golden('corr_1')
    input1, input2 = dev(g['f1']), dev(g['f2'])
    with torch.cuda.device_of(input1):
        rbot1, rbot2, output = input1.new(), input2.new(), input1.new()
        correlation_cuda.forward(input1, input2, rbot1, rbot2, output, 4, 1, 4, 1, 1, 1)
    assert (output.cpu() - g['out']).abs().max() <= 2e-6
    grad_output = dev(g['grad_out'])
    with torch.cuda.device_of(input1):
        rbot1, rbot2, grad_input1, grad_input2 = input1.new(), input2.new(), input1.new(), input2.new()
        correlation_cuda.backward(input1, input2, rbot1, rbot2, grad_output, grad_input1, grad_input2, 4, 1, 4, 1, 1, 1)
    assert (grad_input1.cpu() - g['g1']).abs().max() <= 2e-6 and (grad_input2.cpu() - g['g2']).abs().max() <= 2e-6


def test_operator_on_second_device_after_first():
    """LDS opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) is per device: an operator used on cuda:1 after cuda:0
    must work.  Skipped on 1-GPU boxes."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    from upflow_pytorch_amd import ops
    g = torch.Generator().manual_seed(1)
    f1 = torch.randn(1, 32, 16, 32, generator=g).bfloat16()
    f2 = torch.randn(1, 32, 16, 32, generator=g).bfloat16()
    a = ops.corr81(f1.cuda(0), f2.cuda(0)).cpu()
    b = ops.corr81(f1.cuda(1), f2.cuda(1)).cpu()
    assert torch.equal(a, b)


# ------------------------------------------------------------------------------- loss-side operators (csrc/loss.hip)
@pytest.mark.parametrize('i', range(3))
def test_boundary_warp_hip_golden(hip, i):
    """upf_boundary_warp_forward / _backward against the reference's tools.boundary_dilated_warp.warp_im and its autograd
    gradient wrt the flow (tests/golden/bwarp_*.npz)."""
    g = load_golden('bwarp_%d' % i)
    flow = dev(g['flow']).requires_grad_(True)
    out = hip.boundary_warp(dev(g['image']), flow, dev(g['start']))
    assert (out.detach().cpu() - g['out']).abs().max() <= 1e-6
    (gf,) = torch.autograd.grad(out, flow, dev(g['grad_out']))
    assert (gf.cpu() - g['gflow']).abs().max() <= 1e-5
    (gf2,) = torch.autograd.grad(hip.boundary_warp(dev(g['image']), flow, dev(g['start'])), flow, dev(g['grad_out']))
    assert torch.equal(gf, gf2)
    with pytest.raises(RuntimeError):
        hip.boundary_warp(g['image'], g['flow'], g['start'])                  # CPU tensors: no fallback


def test_boundary_warp_out_of_frame_and_nan(hip):
    """Clamp-to-edge: samples far outside the frame read edge pixels with the reference's clamped-corner weights
    (utils/tools.py:409-412, :458-466); NaN / huge flows must not fault."""
    g = torch.Generator().manual_seed(1)
    I = torch.rand(1, 3, 12, 16, generator=g)
    flow = torch.zeros(1, 2, 6, 8)
    flow[0, 0, 0, 0], flow[0, 1, 1, 1], flow[0, 0, 2, 2] = 1e30, -1e30, float('nan')
    flow[0, :, 3, 3] = torch.tensor([100.5, -40.25])
    start = torch.tensor([3.0, 2.0]).view(1, 2, 1, 1)
    got = hip.boundary_warp(I.cuda(), flow.cuda(), start.cuda()).cpu()
    want = oops.boundary_warp(I, flow, start)
    ok = torch.ones(6, 8, dtype=torch.bool)
    ok[0, 0] = ok[1, 1] = ok[2, 2] = False                                    # (int conversion of 1e30 / NaN is undefined in torch)
    assert (got - want)[0, :, ok].abs().max() <= 1e-6 and torch.isfinite(got[0, :, ok]).all()


@pytest.mark.parametrize('i', range(2))
def test_robust_loss_hip_golden(hip, i):
    g = load_golden('robust_%d' % i)
    for tag, occ in (('mean', None), ('occ', g['occ'])):
        x, y = dev(g['x']).requires_grad_(True), dev(g['y']).requires_grad_(True)
        s, so = hip.robust_loss_sums(x, y, None if occ is None else dev(occ))
        v = s / (so + 1e-6) if occ is not None else s / x.numel()
        assert abs(float(v) - float(g['loss_' + tag])) <= 2e-6 * max(1.0, abs(float(v)))
        gx, gy = torch.autograd.grad(v, (x, y))
        assert (gx.cpu() - g['gx_' + tag]).abs().max() <= 1e-7 and (gy.cpu() - g['gy_' + tag]).abs().max() <= 1e-7
        s2, _ = hip.robust_loss_sums(x, y, None if occ is None else dev(occ))
        assert torch.equal(s, s2)                                             # deterministic reduction


@pytest.mark.parametrize('shape', [(4, 3, 256, 832), (2, 2, 64, 208), (1, 3, 5, 7)])
def test_robust_loss_hip_vs_oracle(hip, shape):
    """Full training sizes (config 3: 256x832 crops): value and gradients against the oracle's torch spelling."""
    gg = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(shape, generator=gg) - 0.45
    y = x + 0.1 * torch.randn(shape, generator=gg)
    occ = (torch.rand(shape[0], 1, shape[2], shape[3], generator=gg) > 0.2).float()
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    sw, sow = oops.robust_loss_sums(xr, yr, occ)
    vw = sw / (sow + 1e-6)
    gxw, gyw = torch.autograd.grad(vw, (xr, yr))
    xd, yd = dev(x).requires_grad_(True), dev(y).requires_grad_(True)
    s, so = hip.robust_loss_sums(xd, yd, dev(occ))
    v = s / (so + 1e-6)
    gx, gy = torch.autograd.grad(v, (xd, yd))
    assert abs(float(v) - float(vw)) <= 5e-6 * abs(float(vw)) and float(so) == float(sow)
    assert relerr(gx.cpu(), gxw) <= 1e-5 and relerr(gy.cpu(), gyw) <= 1e-5


@pytest.mark.parametrize('i', range(2))
def test_smooth_edge1_hip_golden(hip, i):
    g = load_golden('smooth1_%d' % i)
    pred = dev(g['pred']).requires_grad_(True)
    v = hip.smooth_edge1(dev(g['img']), pred)
    assert abs(float(v) - float(g['loss'])) <= 2e-6
    (gp,) = torch.autograd.grad(v, pred)
    assert (gp.cpu() - g['gpred']).abs().max() <= 1e-8
    (gp2,) = torch.autograd.grad(hip.smooth_edge1(dev(g['img']), pred), pred)
    assert torch.equal(gp, gp2)


def test_smooth_edge1_hip_vs_oracle_full_size(hip):
    gg = torch.Generator().manual_seed(3)
    img = torch.rand(4, 3, 256, 832, generator=gg) - 0.45
    pred = torch.randn(4, 2, 256, 832, generator=gg) * 2
    pr = pred.clone().requires_grad_(True)
    vw = oops.smooth_edge1(img, pr)
    (gw,) = torch.autograd.grad(vw, pr)
    pd = dev(pred).requires_grad_(True)
    v = hip.smooth_edge1(dev(img), pd)
    (gp,) = torch.autograd.grad(v, pd)
    assert abs(float(v) - float(vw)) <= 5e-6 * abs(float(vw))
    assert (gp.cpu() - gw).abs().max() <= 1e-9 + 1e-5 * float(gw.abs().max())


def test_backward_kernels_are_bit_deterministic(hip):
    """warp / SGU-blend backward scatter through 64-bit fixed-point integer atomics (csrc/common.hpp: fix_add): two runs on
    the same bits give the same bits, also where many output pixels hit the same source pixel (constant converging flow)
    and where the channel range is split over workgroups."""
    g = torch.Generator().manual_seed(17)
    for (B, C, H, W) in [(2, 128, 12, 40), (4, 32, 64, 208), (1, 3, 33, 65)]:
        x = torch.randn(B, C, H, W, generator=g).cuda().requires_grad_(True)
        flow = (torch.randn(B, 2, H, W, generator=g) * 3).cuda()
        flow[:, :, : H // 2] = -torch.stack(torch.meshgrid(torch.arange(H // 2), torch.arange(W), indexing='ij')[::-1]).float().cuda() + 2.25   # all -> one pixel
        flow = flow.requires_grad_(True)
        gy = torch.randn(B, C, H, W, generator=g).cuda()
        outs = []
        for _ in range(2):
            y = hip.warp(x, flow, 'literal', B // 2)
            outs.append(torch.autograd.grad(y, (x, flow), gy))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for (B, h, w, Hf, Wf) in [(2, 24, 80, 24, 80), (2, 16, 52, 64, 208)]:
        xo = torch.randn(B, 3, h, w, generator=g).cuda().requires_grad_(True)
        fi = (torch.randn(B, 2, Hf, Wf, generator=g) * 3).cuda().requires_grad_(True)
        gu = torch.randn(B, 2, Hf, Wf, generator=g).cuda()
        outs = []
        for _ in range(2):
            up = hip.sgu_blend(fi, xo, None if (Hf, Wf) == (h, w) else fi)[1]
            outs.append(torch.autograd.grad(up, (xo, fi), gu))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('shape', [(2, 8, 12, 20), (1, 32, 24, 40)])
def test_warp_backward_vs_oracle_with_batch_shift_and_collisions(hip, shape):
    """Gradient of the warp wrt x and flow against autograd through the oracle, incl. the stacked-batch form
    (batch_shift: item n samples item (n + shift) % B, so gx lands on the shifted item) and many-to-one sampling."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(5 + sum(shape))
    x = torch.randn(shape, generator=g)
    flow = torch.randn(B, 2, H, W, generator=g) * 2
    flow[:, :, :3] = 0.5 - torch.stack(torch.meshgrid(torch.arange(3), torch.arange(W), indexing='ij')[::-1]).float()   # converge on (0.5, 0.5)
    gy = torch.randn(shape, generator=g)
    for shift in (0, B // 2):
        xr = x.clone().requires_grad_(True)
        fr = flow.clone().requires_grad_(True)
        yw = oracle.warp(torch.roll(xr, -shift, 0), fr, 'robust')
        gxw, gfw = torch.autograd.grad(yw, (xr, fr), gy)
        xd, fd = dev(x).requires_grad_(True), dev(flow).requires_grad_(True)
        gx, gf = torch.autograd.grad(hip.warp(xd, fd, 'robust', shift), (xd, fd), dev(gy))
        assert relerr(gx.cpu(), gxw) <= 1e-5 and relerr(gf.cpu(), gfw) <= 1e-4


def test_flow_sum3_autograd(hip):
    """ops.flow_sum3 = a + (b + c) in fp32 as one launch each way (model/upflow.py:566-572): value and gradients vs the tensor
    expression it replaces."""
    g = torch.Generator().manual_seed(5)
    a = torch.randn(2, 2, 9, 14, generator=g).cuda().requires_grad_(True)
    b = torch.randn(2, 2, 9, 14, generator=g).bfloat16().cuda().requires_grad_(True)
    c = torch.randn(2, 2, 9, 14, generator=g).bfloat16().cuda().requires_grad_(True)
    go = torch.randn(2, 2, 9, 14, generator=g).cuda()
    y = hip.flow_sum3(a, b, c)
    ga, gb, gc = torch.autograd.grad(y, (a, b, c), go)
    ar, br, cr = (t.detach().clone().requires_grad_(True) for t in (a, b, c))
    yr = ar + (br.float() + cr.float())
    gar, gbr, gcr = torch.autograd.grad(yr, (ar, br, cr), go)
    assert y.dtype == torch.float32 and torch.equal(y, yr)
    assert torch.equal(ga, gar) and torch.equal(gb, gbr) and torch.equal(gc, gcr)


def _level_sizes():
    """Every H and W a pyramid level (or the full frame) of BASELINE configs 1-5, KITTI's native 375x1242 and the golden
    sizes has: the divisors max(size - 1, 1) of the sampling positions."""
    sizes = set()
    for (H, W) in [(256, 256), (384, 1280), (256, 832), (448, 1024), (960, 2880), (375, 1242), (64, 128), (128, 192), (48, 160), (96, 320)]:
        h, w = H, W
        sizes.update((h, w))
        for _ in range(6):
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            sizes.update((h, w))
    return sorted(sizes)


def test_division_free_quotient_is_exact():
    """sampling.hpp: 2(j+fx)/(W-1) without the division sequence (Markstein's correction with a host-side reciprocal) is the
    IEEE quotient bit for bit — ALL 2^32 numerators, every divisor of every pyramid level of every configuration (VERDICT r2:
    'prove it ... on an exhaustive sweep').  The validity-mask goldens (1.4 M reference samples) pin the same thing end to end."""
    from upflow_pytorch_amd import _lib
    sizes = _level_sizes()
    assert len(sizes) >= 40 and 2880 in sizes and 20 in sizes and 1 in sizes
    bad = torch.zeros(len(sizes), dtype=torch.int64, device='cuda')
    for k, sz in enumerate(sizes):
        _lib.call('upf_div_selftest', int(sz), _lib._vp(bad[k:].data_ptr()), _lib.stream_ptr(bad.device))
    torch.cuda.synchronize()
    assert int(bad.sum()) == 0, {sz: int(b) for sz, b in zip(sizes, bad.tolist()) if b}
