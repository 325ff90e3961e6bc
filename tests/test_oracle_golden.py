"""Pin the oracle (oracle/ops.py) against the vectors generated from the imported reference
(tests/golden/make_golden.py).  CPU only."""
import glob
import os

import pytest
import torch

import oracle
from oracle import ops
from conftest import load_golden, unpack_mask, GOLDEN


@pytest.mark.parametrize('i', [0, 1, 2, 3])
def test_corr_forward_backward(i):
    g = load_golden('corr_%d' % i)
    out = oracle.corr81(g['f1'], g['f2'])
    # bit-exact at even sizes was the survey's probe; summation order differs from unfold+mean at
    # some shapes, so allow 1 ulp-class error
    assert (out - g['out']).abs().max() <= 2e-6
    out_u = oracle.corr81_unfold(g['f1'], g['f2'])
    assert torch.equal(out_u, g['out']), 'unfold restatement must be bit-identical to Corr_pyTorch'
    g1, g2 = oracle.corr81_backward(g['f1'], g['f2'], g['grad_out'])
    assert (g1 - g['g1']).abs().max() <= 2e-6
    assert (g2 - g['g2']).abs().max() <= 2e-6


WARP_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'warp_*.npz')))


@pytest.mark.parametrize('name', WARP_CASES)
def test_warp(name):
    g = load_golden(name)
    x, flow = g['x'], g['flow']
    B, C, H, W = x.shape
    mask = ops.warp_mask(flow, H, W, 'literal')
    ref_mask = unpack_mask(g['mask'], (B, 1, H, W))
    assert torch.equal(mask, ref_mask), 'mask bits differ: %d' % int((mask != ref_mask).sum())
    y = oracle.warp(x, flow, 'literal')
    assert (y - g['y']).abs().max() <= 1e-6 * max(1.0, float(g['y'].abs().max()))
    y2 = oracle.warp(x, flow, None)
    assert (y2 - g['y_nomask']).abs().max() <= 1e-6 * max(1.0, float(g['y_nomask'].abs().max()))
    gx, gf = oracle.warp_backward(x, flow, g['grad_out'], 'literal')
    assert (gx - g['gx']).abs().max() <= 1e-5
    assert (gf - g['gflow']).abs().max() <= 1e-4 * max(1.0, float(g['gflow'].abs().max()))
    gx2, gf2 = oracle.warp_backward(x, flow, g['grad_out'], None)
    assert (gx2 - g['gx_nomask']).abs().max() <= 1e-5
    assert (gf2 - g['gflow_nomask']).abs().max() <= 1e-4 * max(1.0, float(g['gflow_nomask'].abs().max()))


@pytest.mark.parametrize('i', range(6))
def test_flow_upsample(i):
    g = load_golden('upsample_%d' % i)
    h, w = [int(v) for v in g['size']]
    y = oracle.flow_upsample(g['x'], h, w, True)
    tol = 2e-6 * max(1.0, float(g['y'].abs().max()))
    assert (y - g['y']).abs().max() <= tol
    assert (oracle.flow_upsample(g['x'], h, w, False) - g['y_norate']).abs().max() <= tol
    assert (y - g['y_upsample_flow']).abs().max() <= tol
    xr = g['x'].clone().requires_grad_(True)
    gx, = torch.autograd.grad(oracle.flow_upsample(xr, h, w, True), xr, g['grad_out'])
    assert (gx - g['gx']).abs().max() <= 1e-5 * max(1.0, float(g['gx'].abs().max()))


@pytest.mark.parametrize('i', range(3))
def test_normalize(i):
    g = load_golden('normalize_%d' % i)
    na, nb = oracle.normalize_pair(g['a'], g['b'])
    assert (na - g['na']).abs().max() <= 1e-6 * max(1.0, float(g['na'].abs().max()))
    assert (nb - g['nb']).abs().max() <= 1e-6 * max(1.0, float(g['nb'].abs().max()))


@pytest.mark.parametrize('i', range(4))
def test_sgu_blend(i):
    g = load_golden('sgu_blend_%d' % i)
    olf = g.get('output_level_flow')
    fi, flow_up, inter_flow, inter_mask = oracle.sgu_blend(g['flow_init'], g['x_out'], olf)
    tol = 2e-6 * max(1.0, float(g['flow_up'].abs().max()))
    # flow_up samples flow_init at x + inter_flow: a 1-ulp difference of the up-sampled inter_flow
    # (3e-6 at |v|~28) times the field's gradient (<~10/px) bounds what can be pinned here
    assert (flow_up - g['flow_up']).abs().max() <= (tol if olf is None else 5e-5)
    assert (inter_flow - g['inter_flow']).abs().max() <= tol
    assert (inter_mask - g['inter_mask']).abs().max() <= 1e-6
    # gradients through the blend
    xo = g['x_out'].clone().requires_grad_(True)
    if olf is None:
        f0 = g['flow_init'].clone().requires_grad_(True)
        up = oracle.sgu_blend(f0, xo, None)[1]
        gxo, gf0 = torch.autograd.grad(up, (xo, f0), g['grad_out'])
        assert (gf0 - g['g_flow_init']).abs().max() <= 1e-4 * max(1.0, float(g['g_flow_init'].abs().max()))
    else:
        o = olf.clone().requires_grad_(True)
        up = oracle.sgu_blend(g['flow_init'], xo, o)[1]
        gxo, go = torch.autograd.grad(up, (xo, o), g['grad_out'])
        assert (go - g['g_output_level_flow']).abs().max() <= 1e-4 * max(1.0, float(g['g_output_level_flow'].abs().max()))
    assert (gxo - g['g_x_out']).abs().max() <= 1e-4 * max(1.0, float(g['g_x_out'].abs().max()))


@pytest.mark.parametrize('i', range(2))
def test_occ_check(i):
    g = load_golden('occ_%d' % i)
    o1, o2 = oracle.occ_check(g['flow_f'], g['flow_b'])
    # thresholded floats: allow a handful of borderline pixels
    assert (o1 != g['occ_fw']).float().mean() <= 1e-4
    assert (o2 != g['occ_bw']).float().mean() <= 1e-4


def _census_loss(dist, mask, q=0.4, use_occ=True, max_distance=3):
    """The reduction the reference applies to the census distance (utils/loss.py:36-48, :82-90), restated."""
    B, _, H, W = mask.shape
    valid = torch.zeros_like(mask)
    valid[:, :, max_distance:H - max_distance, max_distance:W - max_distance] = 1.0
    d = (dist.abs() + 0.01).pow(q)
    if use_occ:
        m = mask * valid
        return (d * m).sum() / (m.sum() * 2 + 1e-6)
    return d.mean()


@pytest.mark.parametrize('i', [0, 1])
def test_census_distance(i):
    """oracle.census_distance against the reference's census_loss_torch scalars and gradients (utils/loss.py:50-91)."""
    g = load_golden('census_%d' % i)
    im1 = g['img1']
    for k in range(3):
        w = g['img1_warp'].clone().requires_grad_(True)
        v = _census_loss(ops.census_distance(im1, w), g['masks'][k])
        (gw,) = torch.autograd.grad(v, w)
        assert abs(float(v) - float(g['loss_occ_%d' % k])) <= 1e-6 * max(1.0, abs(float(v)))
        assert (gw - g['grad_occ_%d' % k]).abs().max() <= 1e-6
    v = _census_loss(ops.census_distance(im1, g['img1_warp']), g['masks'][0], use_occ=False)
    assert abs(float(v) - float(g['loss_mean'])) <= 1e-6


# ------------------------------------------------------------------------------- loss-side operators
@pytest.mark.parametrize('i', range(3))
def test_boundary_warp_golden(i):
    g = load_golden('bwarp_%d' % i)
    flow = g['flow'].clone().requires_grad_(True)
    out = ops.boundary_warp(g['image'], flow, g['start'])
    assert (out - g['out']).abs().max() <= 1e-6
    (gf,) = torch.autograd.grad(out, flow, g['grad_out'])
    assert (gf - g['gflow']).abs().max() <= 1e-5


@pytest.mark.parametrize('i', range(2))
def test_robust_loss_golden(i):
    g = load_golden('robust_%d' % i)
    for tag, occ in (('mean', None), ('occ', g['occ'])):
        x, y = g['x'].clone().requires_grad_(True), g['y'].clone().requires_grad_(True)
        s, so = ops.robust_loss_sums(x, y, occ)
        v = s / (so + 1e-6) if occ is not None else s / x.numel()
        assert abs(float(v) - float(g['loss_' + tag])) <= 1e-6 * max(1.0, abs(float(v)))
        gx, gy = torch.autograd.grad(v, (x, y))
        assert (gx - g['gx_' + tag]).abs().max() <= 1e-7 and (gy - g['gy_' + tag]).abs().max() <= 1e-7


@pytest.mark.parametrize('i', range(2))
def test_smooth_edge1_golden(i):
    g = load_golden('smooth1_%d' % i)
    pred = g['pred'].clone().requires_grad_(True)
    v = ops.smooth_edge1(g['img'], pred)
    assert abs(float(v) - float(g['loss'])) <= 1e-6
    (gp,) = torch.autograd.grad(v, pred)
    assert (gp - g['gpred']).abs().max() <= 1e-8


# ------------------------------------------------------------------------------- general-parameter correlation (round 4: pinned)
def test_correlation_general_hand_computed_vectors():
    """Known answers worked out BY HAND from correlation_forward<T> (correlation_cuda_kernel.cu:41-114): output pixel (by, bx) is
    centred at padded (by*s1 + md, bx*s1 + md), channel tc = (tj + dr)*(2dr + 1) + (ti + dr) multiplies by the pixel displaced by
    (tj*s2, ti*s2), zeros outside the image, divided by k*k*C."""
    from oracle import ops as oops
    t = lambda v: torch.tensor(v, dtype=torch.float32)
    # (pad, k, md, s1, s2) = (1,1,1,1,1), one channel, 2x2
    f1, f2 = t([[[[1, 2], [3, 4]]]]), t([[[[5, 6], [7, 8]]]])
    want = t([[[0, 0], [0, 20]], [[0, 0], [15, 24]], [[0, 0], [18, 0]],
              [[0, 10], [0, 28]], [[5, 12], [21, 32]], [[6, 0], [24, 0]],
              [[0, 14], [0, 0]], [[7, 16], [0, 0]], [[8, 0], [0, 0]]]).unsqueeze(0)
    for fn in (oops.correlation_general, oops.correlation_forward_literal):
        assert torch.equal(fn(f1, f2, 1, 1, 1, 1, 1), want), fn.__name__
    # stride2 = 2 (displacements of +-2 px), two channels (division by C), 1x3
    f1, f2 = t([[[[1, 2, 3]], [[1, 1, 1]]]]), t([[[[4, 5, 6]], [[2, 0, 1]]]])
    want = torch.zeros(1, 9, 1, 3)
    want[0, 4, 0] = t([3.0, 5.0, 9.5])
    want[0, 5, 0] = t([3.5, 0, 0])
    want[0, 3, 0] = t([0, 0, 7.0])
    for fn in (oops.correlation_general, oops.correlation_forward_literal):
        assert torch.equal(fn(f1, f2, 2, 1, 2, 1, 2), want), fn.__name__
    # stride1 = 2 (every second pixel is an output), 1x5
    f1, f2 = t([[[[1, 2, 3, 4, 5]]]]), t([[[[6, 7, 8, 9, 10]]]])
    want = torch.zeros(1, 9, 1, 3)
    want[0, 4, 0] = t([6, 24, 50])
    want[0, 5, 0] = t([7, 27, 0])
    want[0, 3, 0] = t([0, 21, 45])
    for fn in (oops.correlation_general, oops.correlation_forward_literal):
        assert torch.equal(fn(f1, f2, 1, 1, 1, 2, 1), want), fn.__name__


@pytest.mark.parametrize('params', [(4, 1, 4, 1, 1), (2, 1, 2, 1, 1), (3, 1, 4, 1, 2), (2, 3, 3, 1, 2), (4, 1, 4, 2, 1), (3, 3, 5, 2, 3),
                                    (0, 1, 1, 1, 1), (5, 1, 3, 1, 3), (6, 5, 5, 1, 3)])
def test_correlation_general_equals_the_literal_kernel_emulation(params):
    """The vectorised restatement (what the GPU tests compare with, at sizes a scalar loop cannot reach) against the
    thread-by-thread emulation of the reference's CUDA kernels with their own flat index arithmetic — forward for every parameter
    set inside the reference's defined domain, and the autograd gradient against the emulated backward kernels where those are the
    gradient of the forward (kernel_size 1, stride1 1)."""
    from oracle import ops as oops
    pad, k, md, s1, s2 = params
    g = torch.Generator().manual_seed(sum(params))
    f1, f2 = torch.randn(2, 35, 9, 10, generator=g), torch.randn(2, 35, 9, 10, generator=g)   # 35 channels: two rounds of the 32 threads
    assert oops.correlation_well_defined(*params)
    a = oops.correlation_general(f1, f2, *params)
    assert (a - oops.correlation_forward_literal(f1, f2, *params)).abs().max() <= 1e-6
    if params == (4, 1, 4, 1, 1):
        assert torch.equal(a, oops.corr81(f1, f2))                    # ... which is pinned on the reference's python fallback
    if oops.correlation_backward_supported(*params):
        go = torch.randn(a.shape, generator=g)
        ga = oops.correlation_general_backward(f1, f2, go, *params)
        gl = oops.correlation_backward_literal(f1, f2, go, *params)
        assert (ga[0] - gl[0]).abs().max() <= 2e-6 and (ga[1] - gl[1]).abs().max() <= 2e-6
        if params == (4, 1, 4, 1, 1):
            g81 = oops.corr81_backward(f1, f2, go)
            assert (ga[0] - g81[0]).abs().max() <= 2e-6 and (ga[1] - g81[1]).abs().max() <= 2e-6
    else:
        assert k > 1 or s1 > 1


def test_correlation_general_refuses_the_undefined_domain():
    from oracle import ops as oops
    f = torch.zeros(1, 1, 8, 8)
    for params in [(3, 3, 2, 1, 1), (2, 3, 2, 2, 1), (4, 5, 4, 1, 3)]:
        assert not oops.correlation_well_defined(*params)
        with pytest.raises(ValueError):
            oops.correlation_general(f, f, *params)
        with pytest.raises(AssertionError):
            oops.correlation_forward_literal(f, f, *params)             # the emulation's bounds assertion fires: a foreign read
