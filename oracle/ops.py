"""CPU restatement (torch CPU tensors, explicit formulas) of the reference's hot-path operators.

TEST INFRASTRUCTURE — see oracle/__init__.py.  Every function cites the reference file:line it
follows (paths relative to /root/reference).  All arithmetic is IEEE fp32, one rounding per
elementwise op (torch CPU never contracts a*b+c into an FMA across ops), which is what makes the
`mask >= 1.0` bits of the warp reproducible.
"""
import torch
import torch.nn.functional as F

R = 4          # max_displacement (model/upflow.py:335, :561)
D = 2 * R + 1  # 9
ND = D * D     # 81


# ------------------------------------------------------------------------------------------------
# cost volume
# ------------------------------------------------------------------------------------------------
def corr81(f1, f2):
    """out[n, 9*(dy+4)+(dx+4), y, x] = (1/C) * sum_c f1[n,c,y,x] * f2[n,c,y+dy,x+dx], f2 zero outside.

    Direct statement of what `correlation_forward<T>` computes with the only parameters the model
    uses, (pad,k,md,s1,s2) = (4,1,4,1,1): model/correlation_package/correlation_cuda_kernel.cu:41-114
    (channel order tc = (tj+r)*(2r+1)+(ti+r), :106; division by nelems = k*k*C, :73,:108), called at
    model/upflow.py:561-562.  Same result as utils/pytorch_correlation.py:27-50 (mean over C, :47).
    Accumulates in the input dtype, channel by channel (fp32 -> fp32 like the CUDA kernel's scalar_t).
    """
    B, C, H, W = f1.shape
    f2p = F.pad(f2, (R, R, R, R))
    out = f1.new_zeros(B, ND, H, W)
    for dy in range(D):
        for dx in range(D):
            out[:, dy * D + dx] = (f1 * f2p[:, :, dy:dy + H, dx:dx + W]).sum(1) / C
    return out


def corr81_unfold(in1, in2, pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1):
    """The reference's pure-PyTorch fallback algorithm, restated: utils/pytorch_correlation.py:27-50.

    unfold(k=1) both inputs (:30-31) -> view the second as B*C single-channel images (:35-36) ->
    unfold again with a kernel as large as the image and padding=pad, which enumerates the 81
    shifted copies (:38) -> [B, C, H*W, 81] materialised -> multiply by f1 (:45) -> mean over C (:46).
    This is the algorithm `bench.py`'s cpu_baseline times.
    """
    assert pad_size == max_displacement and stride1 == stride2 == 1
    B, C, H, W = in1.shape
    k = kernel_size
    a = F.unfold(in1, kernel_size=k, padding=k // 2, stride=stride1)          # [B, C*k*k, H*W]
    b = F.unfold(in2, kernel_size=k, padding=k // 2, stride=stride2)
    ck = b.shape[1]
    b = b.reshape(B * ck, 1, H, W)
    b = F.unfold(b, kernel_size=(H, W), padding=pad_size, stride=stride2)      # [B*ck, H*W, 81]
    nwin = b.shape[2]
    b = b.reshape(B, ck, H * W, nwin).permute(0, 3, 1, 2)                      # [B, 81, ck, H*W]
    res = (b * a.unsqueeze(1)).mean(dim=2)
    return res.reshape(B, nwin, H, W)


def corr81_backward(f1, f2, grad_out):
    """gI1[n,c,y,x] = (1/C) sum_d gO[n,d,y,x] * f2[n,c,y+dy,x+dx];
    gI2[n,c,y,x] = (1/C) sum_d gO[n,d,y-dy,x-dx] * f1[n,c,y-dy,x-dx]  (out-of-range terms dropped).

    model/correlation_package/correlation_cuda_kernel.cu:116-207 (input1), :209-300 (input2) with
    (pad,k,md,s1,s2) = (4,1,4,1,1).
    """
    B, C, H, W = f1.shape
    f2p = F.pad(f2, (R, R, R, R))
    g1 = torch.zeros_like(f1)
    g2p = f1.new_zeros(B, C, H + 2 * R, W + 2 * R)
    for dy in range(D):
        for dx in range(D):
            go = grad_out[:, dy * D + dx].unsqueeze(1)
            g1 += go * f2p[:, :, dy:dy + H, dx:dx + W]
            g2p[:, :, dy:dy + H, dx:dx + W] += go * f1
    return g1 / C, g2p[:, :, R:R + H, R:R + W] / C


def correlation_well_defined(pad_size, kernel_size, max_displacement, stride1, stride2):
    """Parameter sets for which `correlation_forward<T>` (correlation_cuda_kernel.cu:41-114) only reads INSIDE its padded
    buffers.  The kernel centres output pixel (by, bx) at padded row y1 = by*stride1 + max_displacement (:62) and reads rows
    y1 + j and y1 + tj*stride2 + j for |j| <= kernel_rad, |tj| <= max_displacement/stride2 (:87-91): the smallest row index is
    max_displacement - (max_displacement/stride2)*stride2 - kernel_rad, negative whenever kernel_size > 1 and stride2 divides
    max_displacement closely enough — there the reference reads whatever precedes the buffer (undefined behaviour), so no
    output of it can be restated.  The largest index stays inside by construction of the output size (correlation_cuda.cc:24-34)."""
    kr = (kernel_size - 1) // 2
    return max_displacement - (max_displacement // stride2) * stride2 - kr >= 0


def correlation_general(f1, f2, pad_size, kernel_size, max_displacement, stride1, stride2):
    """General-parameter cost volume, restating correlation_cuda_kernel.cu:41-114 and the shape math
    of correlation_cuda.cc:19-34: padded inputs, kernel radius kr=(k-1)/2, displacement radius
    dr=md/stride2, output (2dr+1)^2 channels of size ceil((H+2p-2(kr+md))/s1).
    PINNED (round 4) on `correlation_forward_literal` below — a scalar, thread-by-thread emulation of the CUDA kernel's own
    index arithmetic — and on hand-computed vectors (tests/test_oracle_golden.py); at (4,1,4,1,1) it is corr81, which is pinned
    on the reference's Python fallback.  Raises outside `correlation_well_defined`.
    """
    import math
    if not correlation_well_defined(pad_size, kernel_size, max_displacement, stride1, stride2):
        raise ValueError('correlation: the reference reads outside its padded buffer for these parameters (undefined)')
    B, C, H, W = f1.shape
    kr = (kernel_size - 1) // 2
    br = kr + max_displacement
    pH, pW = H + 2 * pad_size, W + 2 * pad_size
    oH = int(math.ceil((pH - 2 * br) / stride1))
    oW = int(math.ceil((pW - 2 * br) / stride1))
    dr = max_displacement // stride2
    ds = 2 * dr + 1
    p1 = F.pad(f1, (pad_size,) * 4)
    p2 = F.pad(f2, (pad_size,) * 4)
    out = f1.new_zeros(B, ds * ds, oH, oW)
    nelems = kernel_size * kernel_size * C
    ys = torch.arange(oH) * stride1 + max_displacement
    xs = torch.arange(oW) * stride1 + max_displacement
    for tj in range(-dr, dr + 1):
        for ti in range(-dr, dr + 1):
            acc = f1.new_zeros(B, oH, oW)
            for j in range(-kr, kr + 1):
                for i in range(-kr, kr + 1):
                    a = p1[:, :, (ys + j)][:, :, :, (xs + i)]
                    b = p2[:, :, (ys + tj * stride2 + j)][:, :, :, (xs + ti * stride2 + i)]
                    acc = acc + (a * b).sum(1)
            out[:, (tj + dr) * ds + (ti + dr)] = acc / nelems
    return out


def _cdiv_trunc(a, b):
    """C integer division (truncation toward zero), as the CUDA kernels compute their index bounds."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def _channels_first(x, pad_size):
    """channels_first<T> (correlation_cuda_kernel.cu:15-39): NCHW -> zero-padded NHWC, as a flat list-like numpy buffer."""
    import numpy as np
    B, C, H, W = x.shape
    r = np.zeros((B, H + 2 * pad_size, W + 2 * pad_size, C), dtype=np.float64)
    r[:, pad_size:pad_size + H, pad_size:pad_size + W, :] = x.permute(0, 2, 3, 1).double().numpy()
    return r.reshape(-1)


def correlation_forward_literal(f1, f2, pad_size, kernel_size, max_displacement, stride1, stride2):
    """`correlation_forward<T>` emulated block by block, thread by thread, with the kernel's OWN flat index arithmetic
    (correlation_cuda_kernel.cu:41-114: pdimyxc / pdimxc / pdimc strides :66-68, indx1 / indx2 :88-89, the 32 per-thread partial
    sums and their serial reduction :81-104, tc and tindx :105-107, division by nelems :108) on the padded NHWC buffers that
    channels_first<T> builds (:15-39), launched over (batch, outputHeight, outputWidth) blocks of 32 threads as
    correlation_forward_cuda_kernel does (:302-393), output geometry from correlation_cuda.cc:24-34.  Pure python scalar loops in
    float64 — small cases only; asserts that no index leaves the buffer (the reference would read foreign memory there)."""
    import math
    import numpy as np
    THREADS = 32
    B, C, H, W = f1.shape
    kernel_rad = (kernel_size - 1) // 2
    border = kernel_rad + max_displacement
    pH, pW = H + 2 * pad_size, W + 2 * pad_size
    oH = int(math.ceil(float(pH - 2 * border) / float(stride1)))
    oW = int(math.ceil(float(pW - 2 * border) / float(stride1)))
    displacement_rad = max_displacement // stride2
    displacement_size = 2 * displacement_rad + 1
    nOut = displacement_size * displacement_size
    r1, r2 = _channels_first(f1, pad_size), _channels_first(f2, pad_size)
    pdimyxc, pdimxc, pdimc = pH * pW * C, pW * C, C
    tdimcyx, tdimyx, tdimx = nOut * oH * oW, oH * oW, oW
    nelems = float(kernel_size * kernel_size * pdimc)
    out = np.zeros(B * nOut * oH * oW, dtype=np.float64)
    for n in range(B):
        for by in range(oH):
            for bx in range(oW):
                y1, x1 = by * stride1 + max_displacement, bx * stride1 + max_displacement
                for tj in range(-displacement_rad, displacement_rad + 1):
                    for ti in range(-displacement_rad, displacement_rad + 1):
                        prod_sum = [0.0] * THREADS
                        x2, y2 = x1 + ti * stride2, y1 + tj * stride2
                        for c in range(THREADS):
                            for j in range(-kernel_rad, kernel_rad + 1):
                                for i in range(-kernel_rad, kernel_rad + 1):
                                    for ch in range(c, pdimc, THREADS):
                                        assert 0 <= y1 + j < pH and 0 <= x1 + i < pW and 0 <= y2 + j < pH and 0 <= x2 + i < pW, 'read outside the padded buffer'
                                        indx1 = n * pdimyxc + (y1 + j) * pdimxc + (x1 + i) * pdimc + ch
                                        indx2 = n * pdimyxc + (y2 + j) * pdimxc + (x2 + i) * pdimc + ch
                                        prod_sum[c] += r1[indx1] * r2[indx2]
                        reduce_sum = 0.0
                        for index in range(THREADS):
                            reduce_sum += prod_sum[index]
                        tc = (tj + displacement_rad) * displacement_size + (ti + displacement_rad)
                        out[n * tdimcyx + tc * tdimyx + by * tdimx + bx] = reduce_sum / nelems
    return torch.from_numpy(out.reshape(B, nOut, oH, oW)).to(f1.dtype)


def correlation_backward_supported(pad_size, kernel_size, max_displacement, stride1, stride2):
    """Where the reference's backward kernels ARE the gradient of its forward: kernel_size 1 and stride1 1.  Their blocks sit at
    padded pixel y = blockIdx.x * stride1 + pad_size over an inputHeight x inputWidth grid (correlation_cuda_kernel.cu:129-130,
    :222-223, launch :475-520): with stride1 > 1 most input pixels are never visited and rows past the image are, and with
    kernel_size > 1 the (ymin, ymax) window of :137-141 attributes the whole patch sum to the centre tap — neither is the
    derivative of :86-94.  Nothing in the model uses such parameters (model/upflow.py:335)."""
    return kernel_size == 1 and stride1 == 1 and correlation_well_defined(pad_size, kernel_size, max_displacement, stride1, stride2)


def correlation_general_backward(f1, f2, grad_out, pad_size, kernel_size, max_displacement, stride1, stride2):
    """(gradInput1, gradInput2) of correlation_general by autograd — the definition the HIP kernel is held to where
    correlation_backward_supported; tests check that it equals the literal emulation of the reference's kernels below."""
    a = f1.detach().clone().requires_grad_(True)
    b = f2.detach().clone().requires_grad_(True)
    out = correlation_general(a, b, pad_size, kernel_size, max_displacement, stride1, stride2)
    return torch.autograd.grad(out, (a, b), grad_out)


def correlation_backward_literal(f1, f2, grad_out, pad_size, kernel_size, max_displacement, stride1, stride2):
    """`correlation_backward_input1<T>` / `_input2<T>` (correlation_cuda_kernel.cu:116-207, :209-300) emulated block by block
    with their own index arithmetic: block (y, x, c) per item over the inputHeight x inputWidth x channels grid (:475-520),
    y = blockIdx.x * stride1 + pad_size (:129), the truncating integer divisions of the (xmin .. ymax) windows (:137-141, :252-256),
    the early-outs (:143-151, :258-266), the clamps (:153-157), the 32 per-thread partial sums over output channels (:181-196) and
    their serial reduction (:199-205).  Reads outside the padded buffer are asserted to carry no weight (an in-bounds gradOutput
    window never meets them) and taken as zero.  float64 scalar loops, small cases only."""
    import numpy as np
    THREADS = 32
    B, C, H, W = f1.shape
    nOut, oH, oW = grad_out.shape[1:]
    kernel_rad = (kernel_size - 1) // 2
    displacement_rad = max_displacement // stride2
    displacement_size = 2 * displacement_rad + 1
    assert nOut == displacement_size * displacement_size
    pH, pW = H + 2 * pad_size, W + 2 * pad_size
    r1, r2 = _channels_first(f1, pad_size), _channels_first(f2, pad_size)
    go = grad_out.double().numpy().reshape(-1)
    pdimyxc, pdimxc, pdimc = pH * pW * C, pW * C, C
    tdimcyx, tdimyx, tdimx = nOut * oH * oW, oH * oW, oW
    nelems = float(kernel_size * kernel_size * C)
    g1 = np.zeros((B, C, H, W), dtype=np.float64)
    g2 = np.zeros((B, C, H, W), dtype=np.float64)

    def rd(buf, n, yy, xx, c):
        if 0 <= yy < pH and 0 <= xx < pW:
            return buf[n * pdimyxc + yy * pdimxc + xx * pdimc + c], True
        return 0.0, False

    for n in range(B):
        for by in range(H):
            for bx in range(W):
                y, x = by * stride1 + pad_size, bx * stride1 + pad_size
                if y - pad_size >= H or x - pad_size >= W:
                    continue                                   # (stride1 > 1: the reference writes past the image here)
                for c in range(C):
                    # ---- input1 (:116-207)
                    xmin = _cdiv_trunc(x - kernel_rad - max_displacement, stride1)
                    ymin = _cdiv_trunc(y - kernel_rad - max_displacement, stride1)
                    xmax = _cdiv_trunc(x + kernel_rad - max_displacement, stride1)
                    ymax = _cdiv_trunc(y + kernel_rad - max_displacement, stride1)
                    if not (xmax < 0 or ymax < 0 or xmin >= oW or ymin >= oH or xmin > xmax or ymin > ymax):
                        xmin, xmax, ymin, ymax = max(0, xmin), min(oW - 1, xmax), max(0, ymin), min(oH - 1, ymax)
                        prod_sum = [0.0] * THREADS
                        for t in range(THREADS):
                            for tc in range(t, nOut, THREADS):
                                i2 = (tc % displacement_size - displacement_rad) * stride2
                                j2 = (tc // displacement_size - displacement_rad) * stride2
                                val2, inside = rd(r2, n, y + j2, x + i2, c)
                                for j in range(ymin, ymax + 1):
                                    for i in range(xmin, xmax + 1):
                                        w = go[n * tdimcyx + tc * tdimyx + j * tdimx + i]
                                        prod_sum[t] += w * val2
                        g1[n, c, y - pad_size, x - pad_size] = sum(prod_sum) / nelems
                    # ---- input2 (:209-300)
                    prod_sum = [0.0] * THREADS
                    for t in range(THREADS):
                        for tc in range(t, nOut, THREADS):
                            i2 = (tc % displacement_size - displacement_rad) * stride2
                            j2 = (tc // displacement_size - displacement_rad) * stride2
                            xmin = _cdiv_trunc(x - kernel_rad - max_displacement - i2, stride1)
                            ymin = _cdiv_trunc(y - kernel_rad - max_displacement - j2, stride1)
                            xmax = _cdiv_trunc(x + kernel_rad - max_displacement - i2, stride1)
                            ymax = _cdiv_trunc(y + kernel_rad - max_displacement - j2, stride1)
                            if xmax < 0 or ymax < 0 or xmin >= oW or ymin >= oH or xmin > xmax or ymin > ymax:
                                continue
                            xmin, xmax, ymin, ymax = max(0, xmin), min(oW - 1, xmax), max(0, ymin), min(oH - 1, ymax)
                            val1, inside = rd(r1, n, y - j2, x - i2, c)
                            for j in range(ymin, ymax + 1):
                                for i in range(xmin, xmax + 1):
                                    prod_sum[t] += go[n * tdimcyx + tc * tdimyx + j * tdimx + i] * val1
                    g2[n, c, y - pad_size, x - pad_size] = sum(prod_sum) / nelems
    return torch.from_numpy(g1).to(f1.dtype), torch.from_numpy(g2).to(f1.dtype)


# ------------------------------------------------------------------------------------------------
# backward warp  (grid_sample, bilinear, zeros padding, torch-1.1 align_corners=True semantics)
# ------------------------------------------------------------------------------------------------
def _sample_coords(flow, H, W):
    """Sampling position in pixels, with the reference's normalise -> un-normalise round trip.

    model/pwc_modules.py:187-199: vgrid = meshgrid + flow; vgrid_x = 2*vgrid_x/max(W-1,1) - 1.
    ATen grid_sampler_unnormalize(align_corners=True): ((g + 1) / 2) * (size - 1).
    """
    B = flow.shape[0]
    xx = torch.arange(W, dtype=torch.float32).view(1, 1, W).expand(B, H, W)
    yy = torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(B, H, W)
    gx = 2.0 * (xx + flow[:, 0]) / max(W - 1, 1) - 1.0
    gy = 2.0 * (yy + flow[:, 1]) / max(H - 1, 1) - 1.0
    ix = ((gx + 1.0) / 2.0) * (W - 1)
    iy = ((gy + 1.0) / 2.0) * (H - 1)
    return ix, iy


def _taps(ix, iy, H, W):
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    # ATen bilinear weights: nw=(x1-ix)(y1-iy), ne=(ix-x0)(y1-iy), sw=(x1-ix)(iy-y0), se=(ix-x0)(iy-y0)
    wts = [(x1 - ix) * (y1 - iy), (ix - x0) * (y1 - iy), (x1 - ix) * (iy - y0), (ix - x0) * (iy - y0)]
    pos = [(x0, y0), (x1, y0), (x0, y1), (x1, y1)]
    inb = [((px >= 0) & (px <= W - 1) & (py >= 0) & (py <= H - 1)) for px, py in pos]
    return wts, pos, inb


def warp_mask(flow, H, W, mode='literal'):
    """Validity mask of WarpingLayer_no_div (model/pwc_modules.py:201-206).

    'literal': grid_sample(ones) >= 1.0 — the four weight*tap products summed sequentially
    ((nw+ne)+sw)+se in fp32 without FMA (bit-exact vs ATen).
    'robust' : exact in-bounds predicate 0<=x<=W-1 and 0<=y<=H-1 on meshgrid+flow (H2/P3b; non-default).
    """
    if mode == 'robust':
        B = flow.shape[0]
        xx = torch.arange(W, dtype=torch.float32).view(1, 1, W).expand(B, H, W)
        yy = torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(B, H, W)
        px, py = xx + flow[:, 0], yy + flow[:, 1]
        return ((px >= 0) & (px <= W - 1) & (py >= 0) & (py <= H - 1)).unsqueeze(1)
    ix, iy = _sample_coords(flow, H, W)
    wts, _, inb = _taps(ix, iy, H, W)
    s = torch.zeros_like(ix)
    for w, m in zip(wts, inb):
        s = s + torch.where(m, w, torch.zeros_like(w))
    return (s >= 1.0).unsqueeze(1)


def warp(x, flow, mask_mode='literal'):
    """Backward warp.  mask_mode None -> tools.torch_warp (utils/tools.py:1274-1319);
    'literal' / 'robust' -> WarpingLayer_no_div.forward (model/pwc_modules.py:184-207).
    Differentiable w.r.t. x and flow (the mask is a constant factor), so autograd through this
    function is the oracle for the backward kernels too.
    """
    B, C, H, W = x.shape
    ix, iy = _sample_coords(flow, H, W)
    wts, pos, inb = _taps(ix, iy, H, W)
    xf = x.reshape(B, C, H * W)
    out = None
    for w, (px, py), m in zip(wts, pos, inb):
        idx = (py.clamp(0, H - 1) * W + px.clamp(0, W - 1)).long().view(B, 1, H * W).expand(B, C, H * W)
        v = torch.gather(xf, 2, idx).view(B, C, H, W)
        term = v * torch.where(m, w, torch.zeros_like(w)).unsqueeze(1)
        out = term if out is None else out + term
    if mask_mode is not None:
        out = out * warp_mask(flow.detach(), H, W, mask_mode).to(out.dtype)
    return out


def warp_backward(x, flow, grad_out, mask_mode='literal'):
    xr = x.detach().clone().requires_grad_(True)
    fr = flow.detach().clone().requires_grad_(True)
    y = warp(xr, fr, mask_mode)
    return torch.autograd.grad(y, (xr, fr), grad_out)


# ------------------------------------------------------------------------------------------------
# flow up-sampling
# ------------------------------------------------------------------------------------------------
def _bilinear_ac(x, h, w):
    """F.interpolate(bilinear, align_corners=True) restated: src = dst*(in-1)/(out-1) (0 if out==1),
    i0 = floor(src), l1 = src - i0, value = l0h*(l0w*a + l1w*b) + l1h*(l0w*c + l1w*d).
    Call sites model/pwc_modules.py:74,79,101."""
    B, C, h_, w_ = x.shape
    sy = (h_ - 1) / (h - 1) if h > 1 else 0.0
    sx = (w_ - 1) / (w - 1) if w > 1 else 0.0
    ys = torch.arange(h, dtype=torch.float32) * torch.tensor(sy, dtype=torch.float32)
    xs = torch.arange(w, dtype=torch.float32) * torch.tensor(sx, dtype=torch.float32)
    y0 = ys.floor().long().clamp(max=h_ - 1)
    x0 = xs.floor().long().clamp(max=w_ - 1)
    y1 = (y0 + 1).clamp(max=h_ - 1)
    x1 = (x0 + 1).clamp(max=w_ - 1)
    ly1 = (ys - y0.float()).view(1, 1, h, 1)
    lx1 = (xs - x0.float()).view(1, 1, 1, w)
    ly0, lx0 = 1.0 - ly1, 1.0 - lx1
    a = x[:, :, y0][:, :, :, x0]
    b = x[:, :, y0][:, :, :, x1]
    c = x[:, :, y1][:, :, :, x0]
    d = x[:, :, y1][:, :, :, x1]
    return ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * c + lx1 * d)


def flow_upsample(x, h, w, if_rate=True):
    """upsample2d_flow_as (model/pwc_modules.py:77-90) / upsample_flow (:93-104): bilinear
    align_corners=True resize, then u *= w/w_, v *= h/h_ (size ratios, python floats)."""
    _, _, h_, w_ = x.shape
    res = _bilinear_ac(x, h, w)
    if if_rate:
        scale = torch.tensor([w / w_, h / h_], dtype=torch.float32).view(1, 2, 1, 1)
        res = res * scale
    return res


# ------------------------------------------------------------------------------------------------
# SGU interpolation-blend
# ------------------------------------------------------------------------------------------------
def sgu_blend(flow_init, x_out, output_level_flow=None):
    """model/upflow.py:79-88.  inter_flow = x_out[:, :2]; inter_mask = sigmoid(x_out[:, 2:3]); with
    output_level_flow both are bilinearly up-sampled to its size (flow channels rescaled, :85-86) and
    flow_init := output_level_flow (:87); flow_up = torch_warp(flow_init, inter_flow)*(1-mask) +
    flow_init*mask (:88).  Returns (flow_init, flow_up, inter_flow, inter_mask) like :89."""
    inter_flow = x_out[:, :2]
    inter_mask = torch.sigmoid(x_out[:, 2:3])
    if output_level_flow is not None:
        h, w = output_level_flow.shape[2:]
        inter_flow = flow_upsample(inter_flow, h, w, if_rate=True)
        inter_mask = flow_upsample(inter_mask, h, w, if_rate=False)
        flow_init = output_level_flow
    flow_up = warp(flow_init, inter_flow, None) * (1 - inter_mask) + flow_init * inter_mask
    return flow_init, flow_up, inter_flow, inter_mask


# ------------------------------------------------------------------------------------------------
# feature normalisation, occlusion check, metric
# ------------------------------------------------------------------------------------------------
def normalize_pair(a, b):
    """network_tools.normalize_features with the inference flags of test.py:22-30
    (moments_across_channels=False, moments_across_images=False): per tensor, per sample, per channel
    mean and UNBIASED variance over H*W, (f - mean) / sqrt(var + 1e-16).  model/upflow.py:110-137."""
    out = []
    for f in (a, b):
        mean = f.mean(dim=(2, 3), keepdim=True)
        var = f.var(dim=(2, 3), keepdim=True)
        out.append((f - mean) / torch.sqrt(var + 1e-16))
    return out


def occ_check(flow_f, flow_b, alpha1=0.1, alpha2=0.5):
    """tools.occ_check_model(obj_out_all='obj') as configured at model/upflow.py:364-365:
    forward-backward check (utils/tools.py:550-588, |.|_1 magnitude :559) OR-ed with the
    outgoing-flow mask (:641-677): result = 1 where (consistent) or (flow leaves the image)."""
    def mag(v):
        return torch.sum(torch.pow(v ** 2, 0.5), dim=1, keepdim=True)
    m = mag(flow_f) + mag(flow_b)
    bw_w = warp(flow_b, flow_f, None)
    fw_w = warp(flow_f, flow_b, None)
    thr = alpha1 * m + alpha2
    occ_fw = (mag(flow_f + bw_w) < thr).float()
    occ_bw = (mag(flow_b + fw_w) < thr).float()

    def outgoing(flow):
        B, _, H, W = flow.shape
        xx = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W)
        yy = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1)
        px, py = xx + flow[:, 0:1], yy + flow[:, 1:2]
        return ((px <= W - 1) & (px >= 0) & (py <= H - 1) & (py >= 0)).float()

    def merge(occ, out):
        return ((occ == 1) | (out == 0)).float()
    return merge(occ_fw, outgoing(flow_f)), merge(occ_bw, outgoing(flow_b))


def epe(a, b):
    """Mean end-point error, dataset/kitti_dataset.py:464-475 with mask == 1."""
    return float((a.double() - b.double()).pow(2).sum(1).sqrt().mean())


def census_distance(img1, img2, max_distance=3):
    """Soft census (ternary) distance of two RGB images, the reference's spelling
    (/root/reference/utils/loss.py:52-67): 49-channel identity conv2d, t = d/sqrt(0.81+d^2), sum d2/(0.1+d2).
    -> [B,1,H,W].  (Differentiable torch ops: tests also take its autograd gradient as the oracle.)"""
    import torch.nn.functional as F
    patch = 2 * max_distance + 1
    n = patch * patch

    def ternary(image):
        r, g, b = torch.split(image, 1, 1)
        gray = 0.2989 * r + 0.5870 * g + 0.1140 * b
        weight = torch.eye(n, dtype=gray.dtype, device=gray.device).view(n, 1, patch, patch)
        t = F.conv2d(gray, weight, bias=None, stride=[1, 1], padding=[max_distance, max_distance]) - gray
        return t / torch.sqrt(0.81 + t ** 2)
    d = (ternary(img1) - ternary(img2)) ** 2
    return torch.sum(d / (0.1 + d), 1, keepdim=True)


# ------------------------------------------------------------------------------------------------
# loss-side operators (training): boundary-dilated warp, abs_robust term, edge-aware smoothness
# ------------------------------------------------------------------------------------------------
def boundary_warp(I_nchw, flow_nchw, start_n211):
    """tools.boundary_dilated_warp.warp_im restated (/root/reference/utils/tools.py:351-499): grid + crop offset
    (:353-367), + flow (:497), floor / clamp of the corner indices (:404-412), four gathers, weights from the CLAMPED
    corner coordinates (:458-466), output = wa*Ia + wb*Ib + wc*Ic + wd*Id (:467).  Differentiable torch ops (autograd
    through this function is the oracle of the backward kernel)."""
    B, C, Hi, Wi = I_nchw.shape
    _, _, h, w = flow_nchw.shape
    xx = torch.arange(w, dtype=torch.float32).view(1, 1, w)
    yy = torch.arange(h, dtype=torch.float32).view(1, h, 1)
    start = start_n211.float().reshape(-1, 2)
    if start.shape[0] == 1:
        start = start.expand(B, 2)
    x = (xx + start[:, 0].view(B, 1, 1)) + flow_nchw[:, 0].float()
    y = (yy + start[:, 1].view(B, 1, 1)) + flow_nchw[:, 1].float()
    x0 = torch.floor(x).int()
    y0 = torch.floor(y).int()
    x1 = torch.clamp(x0 + 1, 0, Wi - 1)
    y1 = torch.clamp(y0 + 1, 0, Hi - 1)
    x0 = torch.clamp(x0, 0, Wi - 1)
    y0 = torch.clamp(y0, 0, Hi - 1)
    flat = I_nchw.float().reshape(B, C, Hi * Wi)

    def tap(yi, xi):
        idx = (yi.long() * Wi + xi.long()).view(B, 1, h * w).expand(B, C, h * w)
        return torch.gather(flat, 2, idx).view(B, C, h, w)
    x0f, x1f, y0f, y1f = x0.float(), x1.float(), y0.float(), y1.float()
    wa = ((x1f - x) * (y1f - y)).unsqueeze(1)
    wb = ((x1f - x) * (y - y0f)).unsqueeze(1)
    wc = ((x - x0f) * (y1f - y)).unsqueeze(1)
    wd = ((x - x0f) * (y - y0f)).unsqueeze(1)
    return wa * tap(y0, x0) + wb * tap(y1, x0) + wc * tap(y0, x1) + wd * tap(y1, x1)


def robust_loss_sums(x, y, occ=None, q=0.4, eps=0.01):
    """The two sums of network_tools.photo_loss_multi_type('abs_robust') (model/upflow.py:270-272, :284-287):
    sum((|x - y| + eps)^q * occ) and sum(occ) (occ = 1 without a mask)."""
    d = (torch.abs(x - y) + eps).pow(q)
    if occ is None:
        return d.sum(), torch.tensor(float(x.shape[0] * x.shape[2] * x.shape[3]))
    return (d * occ).sum(), occ.sum()


def smooth_edge1(img, pred):
    """network_tools.edge_aware_smoothness_order1 (model/upflow.py:197-216)."""
    def gx(t):
        return t[:, :, :-1, :] - t[:, :, 1:, :]

    def gy(t):
        return t[:, :, :, :-1] - t[:, :, :, 1:]
    wx = torch.exp(-torch.mean(torch.abs(gx(img)), 1, keepdim=True))
    wy = torch.exp(-torch.mean(torch.abs(gy(img)), 1, keepdim=True))
    return torch.mean(torch.abs(gx(pred)) * wx) + torch.mean(torch.abs(gy(pred)) * wy)
