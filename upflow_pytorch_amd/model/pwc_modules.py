"""Host-side mirror of `/root/reference/model/pwc_modules.py`: the same module/function names,
constructor arguments and state_dict keys, so `UPFlow_net` checkpoints load unchanged.

What runs where: convolutions stay PyTorch-ROCm (MIOpen) as BASELINE.json's north star asks; the
backward warp and the flow up-sampling are single HIP launches from libupflow_hip.so.
"""
import logging

import torch
import torch.nn as nn
import torch.nn.functional as tf

from .. import ops
from ..utils.tools import tools


def conv(in_planes, out_planes, kernel_size=3, stride=1, dilation=1, isReLU=True, if_IN=False, IN_affine=False, if_BN=False):
    """Conv2d('same'-style padding) [+ LeakyReLU(0.1)] [+ InstanceNorm | BatchNorm], as a Sequential
    whose conv is element 0 (keys `<name>.0.weight/bias`).  model/pwc_modules.py:10-49."""
    layers = [nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, dilation=dilation,
                        padding=((kernel_size - 1) * dilation) // 2, bias=True)]
    if isReLU:
        layers.append(nn.LeakyReLU(0.1, inplace=True))
    if if_IN:
        layers.append(nn.InstanceNorm2d(out_planes, affine=IN_affine))
    elif if_BN:
        layers.append(nn.BatchNorm2d(out_planes, affine=IN_affine))
    return nn.Sequential(*layers)


def initialize_msra(modules):
    """Kaiming-normal weights, zero biases for every (transposed) convolution. pwc_modules.py:52-69."""
    logging.info("Initializing MSRA")
    for layer in modules:
        if isinstance(layer, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.kaiming_normal_(layer.weight)
            if layer.bias is not None:
                nn.init.constant_(layer.bias, 0)


def upsample2d_as(inputs, target_as, mode="bilinear"):
    _, _, h, w = target_as.size()
    return tf.interpolate(inputs, [h, w], mode=mode, align_corners=True)     # pwc_modules.py:72-74


def _resize_flow(inputs, h, w, mode, if_rate):
    if mode != "bilinear":
        res = tf.interpolate(inputs, [h, w], mode=mode)
        if if_rate:
            _, _, h_, w_ = inputs.size()
            scale = torch.tensor([w / w_, h / h_], dtype=res.dtype, device=res.device).view(1, 2, 1, 1)
            res = res * scale
        return res
    return ops.flow_upsample(inputs, h, w, if_rate)


def upsample2d_flow_as(inputs, target_as, mode="bilinear", if_rate=False):
    """Bilinear align_corners=True resize to `target_as`'s size; with `if_rate` the u/v channels are
    multiplied by w/w_ and h/h_ (pwc_modules.py:77-90).  Out of place (the reference mutates chunk
    views, which torch-2 autograd rejects: SURVEY.md §7-H6) and fused into one HIP launch."""
    _, _, h, w = target_as.size()
    return _resize_flow(inputs, h, w, mode, if_rate)


def upsample_flow(inputs, target_size=None, target_flow=None, mode="bilinear"):
    """pwc_modules.py:93-104: always rescales the flow values."""
    if target_size is not None:
        h, w = target_size
    elif target_flow is not None:
        _, _, h, w = target_flow.size()
    else:
        raise ValueError('wrong input')
    return _resize_flow(inputs, h, w, mode, True)


def rescale_flow(flow, div_flow, width_im, height_im, to_local=True):
    """pwc_modules.py:107-119."""
    if to_local:
        u_scale = float(flow.size(3) / width_im / div_flow)
        v_scale = float(flow.size(2) / height_im / div_flow)
    else:
        u_scale = float(width_im * div_flow / flow.size(3))
        v_scale = float(height_im * div_flow / flow.size(2))
    scale = torch.tensor([u_scale, v_scale], dtype=flow.dtype, device=flow.device).view(1, 2, 1, 1)
    return flow * scale


class FeatureExtractor(nn.Module):
    """Six [stride-2 conv, conv] stages; returns the pyramid coarsest first. pwc_modules.py:122-142."""

    def __init__(self, num_chs, if_end_relu=True, if_end_norm=False):
        super(FeatureExtractor, self).__init__()
        self.num_chs = num_chs
        self.convs = nn.ModuleList(
            nn.Sequential(conv(ci, co, stride=2), conv(co, co, isReLU=if_end_relu, if_IN=if_end_norm))
            for ci, co in zip(num_chs[:-1], num_chs[1:]))

    def forward(self, x, outs=None, pitched=False):
        """outs (inference fast path): per stage, None or the [B, C, H, W] view the stage's output is written to.
        pitched: the stages' own intermediates are allocated with 16-byte aligned rows (ops.empty_nchw) where their width is ragged."""
        pyramid = []
        cache = self.__dict__.setdefault('_fast_cache', {})
        hip = getattr(self, 'hip_convs', True)         # False: PyTorch-ROCm (MIOpen) even where the MFMA kernel applies
        for i, stage in enumerate(self.convs):
            if (hip and FUSE_PAIRS[0] and not getattr(self, '_no_fuse_pairs', False) and len(stage[0]) == 2 and len(stage[1]) == 2
                    and ops.conv_pair_supported(x, stage[0][0], stage[1][0])):
                # (round 6) the first stages — 3 -> 16 -> 16 at 1/2, 16 -> 32 -> 32 at 1/4 resolution — as ONE launch each, the stride-2 layer's
                # output staying in LDS (csrc/conv_pair.hip)
                pc = cache.get(('pair', i)) or cache.setdefault(('pair', i), _PackedConvPair(stage[0], stage[1]))
                ho, wo = ops.conv3x3_out_hw(x.shape[2], x.shape[3], 2)
                y = outs[i] if (outs is not None and outs[i] is not None) else ops.empty_nchw((x.shape[0], stage[1][0].out_channels, ho, wo), x.dtype, x.device, pitched=pitched)
                x = pc(x, y)
                pyramid.append(x)
                continue
            x = fast_conv_seq(stage[0], x, cache, allow_hip=hip, pitched=pitched)      # stride-2 conv
            x = fast_conv_seq(stage[1], x, cache, out=None if outs is None else outs[i], allow_hip=hip, pitched=pitched)   # stride-1 conv
            pyramid.append(x)
        return pyramid[::-1]

    def out_shapes(self, H, W):
        """[(C, H, W)] of the stage outputs, finest first."""
        shapes = []
        for c in self.num_chs[1:]:
            H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            shapes.append((c, H, W))
        return shapes


class WarpingLayer_no_div(nn.Module):
    """Backward warp by a pixel-unit flow with the `grid_sample(ones) >= 1.0` validity mask
    (pwc_modules.py:179-207) — one fused HIP launch (csrc/warp.hip).

    `mask_mode`: 'literal' (default) reproduces the reference's mask bits exactly; 'robust' is the
    exact in-bounds predicate, an explicit non-default switch used for well-posed whole-network
    parity (protocol P3b)."""

    def __init__(self, mask_mode='literal'):
        super(WarpingLayer_no_div, self).__init__()
        self.mask_mode = mask_mode

    def forward(self, x, flow, batch_shift=0):
        return ops.warp(x, flow, self.mask_mode, batch_shift)


class WarpingLayer(nn.Module):
    """The div_flow variant of pwc_modules.py:156-176: flow is divided by `div_flow` and normalised
    by the IMAGE size; expressed through the same kernel by rescaling the flow to feature pixels."""

    def __init__(self, mask_mode='literal'):
        super(WarpingLayer, self).__init__()
        self.mask_mode = mask_mode

    def forward(self, x, flow, height_im, width_im, div_flow):
        H, W = x.shape[2:]
        sx = (W - 1) / max(width_im - 1, 1) / div_flow
        sy = (H - 1) / max(height_im - 1, 1) / div_flow
        scale = torch.tensor([sx, sy], dtype=torch.float32, device=flow.device).view(1, 2, 1, 1)
        return ops.warp(x, flow.float() * scale, self.mask_mode)


class _PackedConv3x3(object):
    """Lazily packed weights of one `conv(...)` Sequential for the matrix-core kernel (csrc/conv3x3.hip):
    re-packed when weight OR bias changes (autograd version counters), dtype, device or storage.  In-place edits through
    `.data` (EMA, manual surgery) do not bump the version counter: call `net.invalidate_packed()` after those
    (`load_state_dict` / `load_model` do it themselves)."""

    def __init__(self, seq):
        self.conv = seq[0]
        self.slope = 0.1 if any(isinstance(m, nn.LeakyReLU) for m in seq) else 0.0
        self.key = None
        self.packed = None
        self.bias = None
        self.key32 = None
        self.packed32 = None

    def _key(self):
        w, b = self.conv.weight, self.conv.bias
        return (w._version, w.dtype, w.device, w.data_ptr(), b._version, b.data_ptr())

    def get(self):
        key = self._key()
        if key != self.key:
            self.packed = ops.conv3x3_pack(self.conv.weight)
            self.bias = self.conv.bias.detach().float().contiguous()
            self.key = key
        return self.packed, self.bias

    def get32(self):
        """The split-precision operand (csrc/conv_x3.hip) of the fp32 UPCAST of 16-bit weights, for rows shorter than 8 pixels."""
        key = self._key()
        if key != self.key32:
            self.packed32 = ops.conv3x3_pack(self.conv.weight.detach().float())
            self.bias = self.conv.bias.detach().float().contiguous()
            self.key32 = key
        return self.packed32, self.bias

    def invalidate(self):
        self.key = None
        self.key32 = None

    def __call__(self, x_view, y_view):
        c = self.conv
        if x_view.dtype != torch.float32 and x_view.shape[3] < 8:
            # rows shorter than 8 pixels (the coarsest levels of small inputs) are below the 16-bit kernel's tile: the same
            # contraction — exact products of the 16-bit operands, fp32 accumulation, one rounding to 16 bits — through the
            # split-precision kernel on fp32 copies.  (Until round 4 these levels went through MIOpen, whose fp16 kernels are not
            # reproducible from run to run: tools/pipe_stress_small.py, 466 of 720 outputs differing at 2 x 128 x 256.)
            packed, bias = self.get32()
            ho, wo = ops.conv3x3_out_hw(x_view.shape[2], x_view.shape[3], c.stride[0])
            yf = torch.empty((x_view.shape[0], c.out_channels, ho, wo), dtype=torch.float32, device=x_view.device)
            ops.conv3x3_forward_raw(x_view.float(), packed, bias, yf, c.dilation[0], self.slope, c.stride[0], c.kernel_size[0])
            y_view.copy_(yf)
            return y_view
        packed, bias = self.get()
        return ops.conv3x3_forward_raw(x_view, packed, bias, y_view, c.dilation[0], self.slope, c.stride[0], c.kernel_size[0])


def packed_convs(net):
    """Every packed-weight holder (_PackedConv3x3 / _PackedConvC8) the modules of `net` have created so far: the per-module
    caches `_fast_cache`, `_packed` (dense stacks) and `_packed8` (their channel-octet forms)."""
    out = []
    for m in net.modules():
        d = m.__dict__
        out += list(d.get('_fast_cache', {}).values())
        out += list(d.get('_packed', None) or [])
        for v in d.get('_packed8', {}).values():
            out += list(v[0]) if isinstance(v, tuple) else list(v)
    return [pc for pc in out if hasattr(pc, 'invalidate')]


def packed_operands(net):
    """The device tensors behind packed_convs(net) — what a captured graph of net's forward reads besides the parameters."""
    ts = []
    for pc in packed_convs(net):
        ts += [t for t in (getattr(pc, 'packed', None), getattr(pc, 'packed32', None), getattr(pc, 'bias', None)) if torch.is_tensor(t)]
        ts += [t for t in getattr(pc, 'extra_operands', lambda: [])() if torch.is_tensor(t)]
    return ts


_NO_NARROW = [False]        # experiment switch (tools/ab_bench.py "no_narrow=1"): Cout <= 16 layers on the 32-channel kernel


class _PackedConvC8(object):
    """_PackedConv3x3 for the channel-octet entry (ops.conv_c8_forward_raw): weights packed through a channel map — for every
    channel position of the layer's C8 input slice and for every plane of its NCHW tail the input channel of the Conv2d it
    carries (-1: padding)."""

    def __init__(self, seq, c8_channels=(), tail_channels=()):
        self.conv = seq[0]
        self.slope = 0.1 if any(isinstance(m, nn.LeakyReLU) for m in seq) else 0.0
        self.maps = (tuple(c8_channels), tuple(tail_channels))
        # layers with <= 16 output channels on octets only: the 16-channel matrix instruction (ops.conv_c8_forward_narrow_raw)
        self.narrow = (not _NO_NARROW[0]) and ops.conv_c8_narrow_ok(self.conv.out_channels, self.conv.kernel_size[0], self.conv.dilation[0],
                                                                    self.conv.stride[0], len(self.maps[1]) > 0) and len(self.maps[0]) > 0
        self.key = None
        self.packed = None
        self.bias = None

    def get(self):
        w, b = self.conv.weight, self.conv.bias
        key = (w._version, w.dtype, w.device, w.data_ptr(), b._version, b.data_ptr())
        if key != self.key:
            self.packed = ops.conv_c8_pack16(w, self.maps[0]) if self.narrow else ops.conv_c8_pack(w, self.maps[0], self.maps[1])
            self.bias = self.conv.bias.detach().float().contiguous()
            self.key = key
        return self.packed, self.bias

    def invalidate(self):
        self.key = None

    def dual(self, x2, y_a, y_b):
        """A 1x1 projection NCHW -> octets stored into two buffers by one launch (ops.conv1x1_c8_dual_raw); falls back to two launches
        where the rows are not 16-byte aligned."""
        packed, bias = self.get()
        c = self.conv
        p = ops.nchw_pitch(x2)
        if (DUAL_1X1[0] and c.kernel_size == (1, 1) and not self.narrow and c.out_channels <= 32 and p is not None and p % 8 == 0
                and x2.data_ptr() % 16 == 0 and x2.stride(0) % 8 == 0):
            return ops.conv1x1_c8_dual_raw(x2, packed, bias, y_a, y_b, self.slope)
        self(None, x2, y_a)
        return self(None, x2, y_b)

    def __call__(self, x8, x2, y):
        packed, bias = self.get()
        if self.narrow:
            return ops.conv_c8_forward_narrow_raw(x8, packed, bias, y, self.slope)
        return ops.conv_c8_forward_raw(x8, x2, packed, bias, y, self.conv.dilation[0], self.slope, self.conv.kernel_size[0], self.conv.stride[0])


class _PackedTailC8(object):
    """The MERGED NARROW TAIL of a dense stack in the channel-octet layout (round 6; include/upflow_hip.h: upf_conv_forward_c8_split /
    upf_conv_forward_c8_narrow_init).  The stack's layers read nested channel suffixes of one buffer (pwc_modules.py:279-286), so a layer
    j > m reads [conv_{j-1} | ... | conv_m | what conv_m read]:  y_j = W_j[:, front part] * (outputs of conv_m .. conv_{j-1}) + W_j[:, rest] *
    (conv_m's input), and the second term shares its input with conv_m.  A launch with 2 ... 16 output channels costs what the staging of
    its input costs, and a layer whose width is not a whole number of the kernel's 32-channel blocks computes padding anyway: the pass of
    conv_m computes, in those padding channels, the second term of the later narrow layers J as fp32 partials (bias included), and each
    layer of J then is a 16-channel-instruction launch over the few channels IN FRONT of conv_m's input, starting from its partial.
      SGU estimator (model/upflow.py:24-60), m = conv4, J = {conv5, conv_last}: 16 + [8 | 3 -> 4] = 28 channels in one block; 160->16,
          176->8, 184->3 (three passes over >= 160 channels) become 160->28, 16->8, 24->3;
      flow estimator (pwc_modules.py:250-286), m = conv3 (96 channels = three blocks of a four-block workgroup), J = {conv_last}: 371->96
          and 563->2 become 371->100 (the same launch shape) and 192->2.
    seqs: the stack's six Sequentials; m, J: indices into them; kmap: the octet-position map of conv_m's input (as _PackedConvC8's)."""

    def __init__(self, seqs, m, J, kmap):
        self.m, self.J = int(m), sorted(int(j) for j in J)
        self.kmap = list(kmap)
        self.main = seqs[m][0]
        self.convs = [seqs[m][0]] + [seqs[j][0] for j in self.J]
        self.widths = [q[0].out_channels for q in seqs]
        self.slopes = {j: (0.1 if any(isinstance(m_, nn.LeakyReLU) for m_ in seqs[j]) else 0.0) for j in [m] + self.J}
        self.cmain = self.main.out_channels
        self.cin = self.main.in_channels
        # partial record of a pixel: every layer of J padded to a multiple of 4 floats
        self.offsets, off = {}, 0
        for j in self.J:
            self.offsets[j] = off
            off += (self.widths[j] + 3) // 4 * 4
        self.pp = off
        self.cout = self.cmain + off
        self.key = None
        self.packed = self.bias = None
        self.finish = {}                                 # layer index -> packed operand of its finishing launch

    @staticmethod
    def plan(widths):
        """widths: output channels of conv1..conv5, conv_last -> (m, J) or None.  J: later layers of at most 16 channels, taken from the
        head backwards while their partial rows fit the padding of conv_m's blocks (a workgroup computes 1, 2 or 4 blocks of 32
        channels: csrc/conv_c8.hip launch_c8); m: the choice that removes the most staged input, sum over J of conv_m's input."""
        n = len(widths)
        cin = [None] * n                                  # input channels relative to the stack's input: only differences matter
        acc = 0
        for k in range(n):
            cin[k] = acc
            acc += widths[k]
        best, best_score = None, 0
        for m in range(n - 2, -1, -1):
            if widths[m] % 8:
                continue
            mt = (widths[m] + 31) // 32
            free = 32 * (4 if mt >= 3 else mt) - widths[m]
            if mt > 4:
                continue
            J, used = [], 0
            for j in range(n - 1, m, -1):
                need = (widths[j] + 3) // 4 * 4
                if widths[j] > 16 or used + need > free or (j < n - 1 and widths[j] % 8):
                    break
                J.append(j); used += need
            # every layer between m and a member of J must have whole-octet outputs (the finishing launch reads them as octets)
            if not J or any(widths[k] % 8 for k in range(m, n - 1)):
                continue
            score = len(J) * (cin[m] + 1000)              # (+ the stack's input, a constant: more members first, then the later m)
            if score > best_score:
                best, best_score = (m, sorted(J)), score
        return best

    def _key(self):
        k = []
        for c in self.convs:
            k += [c.weight._version, c.weight.dtype, c.weight.device, c.weight.data_ptr(), c.bias._version, c.bias.data_ptr()]
        return tuple(k)

    def get(self):
        key = self._key()
        if key != self.key:
            c0 = self.main
            rows, bias = [c0.weight.detach()], [c0.bias.detach().float()]
            self.finish = {}
            for j, c in zip(self.J, self.convs[1:]):
                front = sum(self.widths[self.m:j])               # layer input = [outputs of conv_m .. conv_{j-1} (front) | conv_m's input (cin)]
                w = c.weight.detach()
                assert w.shape[1] == front + self.cin
                pad = (c.out_channels + 3) // 4 * 4 - c.out_channels
                rows.append(w[:, front:])
                bias.append(c.bias.detach().float())
                if pad:
                    rows.append(w.new_zeros((pad, self.cin) + tuple(w.shape[2:])))
                    bias.append(bias[0].new_zeros(pad))
                self.finish[j] = ops.conv_c8_pack16(w[:, :front].contiguous(), list(range(front)))
            self.packed = ops.conv_c8_pack(torch.cat(rows, 0).contiguous(), self.kmap)
            self.bias = torch.cat(bias).contiguous()
            self.key = key
        return self.packed, self.bias

    def extra_operands(self):
        return list(self.finish.values())

    def invalidate(self):
        self.key = None

    def main_pass(self, x8, y8):
        """conv_m: x8 = its input octets, y8 = its output octets -> the fp32 partial records of the layers J."""
        packed, bias = self.get()
        B, _, H, W, _ = x8.shape
        part = torch.empty((B, self.pp // 4, H, W, 4), dtype=torch.float32, device=x8.device)      # planes of channel quads
        ops.conv_c8_forward_split_raw(x8, packed, bias, y8, part, self.slopes[self.m])
        return part

    def finish_pass(self, j, x8_front, part, y):
        """layer j of J: x8_front = the octets in front of conv_m's input that layer j reads, y = its destination."""
        self.get()
        return ops.conv_c8_forward_narrow_init_raw(x8_front, self.finish[j], part, self.offsets[j], self.widths[j], y, self.slopes[j])


class _PackedConvPair(object):
    """Packed operands of TWO conv(...) Sequentials run as one launch (ops.conv_pair_forward_raw / csrc/conv_pair.hip): 3x3 layers with
    strides (1, 2) — the halves of the SGU guidance stem (model/upflow.py:30-33) — or (2, 1) — the first stages of the feature pyramid
    (pwc_modules.py:122-142)."""

    def __init__(self, seq_a, seq_b):
        self.convs = (seq_a[0], seq_b[0])
        self.strides = (seq_a[0].stride[0], seq_b[0].stride[0])
        self.slopes = tuple(0.1 if any(isinstance(m_, nn.LeakyReLU) for m_ in q) else 0.0 for q in (seq_a, seq_b))
        self.key = None
        self.packed = self.packed_b = self.bias = self.bias_b = None

    def _key(self):
        k = []
        for c in self.convs:
            k += [c.weight._version, c.weight.dtype, c.weight.device, c.weight.data_ptr(), c.bias._version, c.bias.data_ptr()]
        return tuple(k)

    def get(self):
        key = self._key()
        if key != self.key:
            a, b = self.convs
            self.packed, self.packed_b = ops.conv_pair_pack(a.weight, b.weight)
            self.bias, self.bias_b = a.bias.detach().float().contiguous(), b.bias.detach().float().contiguous()
            self.key = key
        return self.packed, self.bias, self.packed_b, self.bias_b

    def extra_operands(self):
        return [self.packed_b, self.bias_b]

    def invalidate(self):
        self.key = None

    def __call__(self, x, y):
        pa, ba, pb, bb = self.get()
        return ops.conv_pair_forward_raw(x, pa, ba, self.slopes[0], pb, bb, self.slopes[1], y, self.strides)


DUAL_1X1 = [True]            # experiment / parity switch: False = the 1x1 projection computed once per destination buffer (rounds 3-5)
FUSE_PAIRS = [True]          # experiment / parity switch: False = every layer of the SGU guidance stem its own launch (rounds 1-5)
MERGE_TAIL = [True]          # experiment / parity switch: False = every layer of a dense stack its own pass (rounds 3-5)


def c8_level_ok(nb, H, W, dtype):
    """Levels whose dense stacks run in the channel-octet layout (ops.conv_c8_*): 16-bit and a grid that fills the chip (the C8
    kernels have no split-K form for the coarse levels).  Any width (round 5): a pixel of an octet tensor is one 16-byte entry, so
    its rows are aligned whatever W is; the NCHW tensors that feed the octet kernels (pyramid features, the frames) are kept
    row-pitched at ragged levels (ops.empty_nchw).  Rounds 3-4 required W % 8 == 0, which no level of a native KITTI frame meets."""
    return dtype in (torch.bfloat16, torch.float16) and nb * ((W + 31) // 32) * ((H + 7) // 8) >= 200


def fast_conv_seq(seq, x, cache, out=None, allow_hip=True, pitched=False):
    """Run one `conv(...)` Sequential (Conv2d [+ LeakyReLU]) — through the matrix-core kernel when it is a
    3x3 (stride 1/2, dilation <= 16) or 1x1 convolution in an eligible inference setting, through MIOpen otherwise.  `cache` is a dict
    that keeps the packed weights per Sequential.  `out`: optional destination (a channel slice of a wider NCHW buffer).
    pitched: a result this function allocates gets 16-byte aligned rows at ragged widths (ops.empty_nchw) — for consumers that are
    pitch-aware (the convolutions, ops.warp_into, the fused cost volume)."""
    c = seq[0]
    k = c.kernel_size[0]
    if (allow_hip and _fast_conv_ok(x) and c.kernel_size in ((3, 3), (1, 1)) and c.stride[0] == c.stride[1] and c.groups == 1 and len(seq) <= 2
            and c.padding == (((k - 1) * c.dilation[0]) // 2,) * 2
            and (ops.conv3x3_supported(x, c.out_channels, c.dilation[0], c.stride[0], k)
                 or (x.shape[3] < 8 and 1 <= c.dilation[0] <= 16 and (c.stride[0] == 1 or (c.stride[0] == 2 and c.dilation[0] == 1 and k == 3))))):
        pc = cache.get(id(seq))           # (16-bit rows shorter than 8 pixels: _PackedConv3x3 takes the split-precision kernel)
        if pc is None:
            pc = cache[id(seq)] = _PackedConv3x3(seq)
        ho, wo = ops.conv3x3_out_hw(x.shape[2], x.shape[3], c.stride[0])
        y = out if out is not None else ops.empty_nchw((x.shape[0], c.out_channels, ho, wo), x.dtype, x.device, pitched=pitched)
        return pc(x, y)
    if torch.is_grad_enabled() and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and c.weight.dtype == torch.float32:
        y = train_conv_seq(seq, x)                       # training on the matrix cores: 16-bit activations, fp32 master weights
    else:
        y = seq(x if x.is_contiguous() else x.contiguous())
    if out is not None:
        out.copy_(y)
        return out
    return y


def train_conv_seq(seq, x):
    """One `conv(...)` Sequential under autograd with 16-bit activations and fp32 master weights (config
    `train_conv_dtype`): forward, data gradient and weight gradient on the matrix cores (ops.ConvTrainFunction) where the
    geometry allows (rows of >= 8 pixels; inside ConvTrainFunction the weight gradient of ragged-width levels and the
    gradients of the stride-2 layers take PyTorch-ROCm's kernels), and the same arithmetic through PyTorch-ROCm in fp32
    otherwise (rows shorter than 8 pixels)."""
    c = seq[0]
    k = c.kernel_size[0]
    slope = 0.0
    for m in list(seq)[1:]:
        if isinstance(m, nn.LeakyReLU):
            slope = float(m.negative_slope)
        else:
            return seq(x.float()).to(x.dtype)
    if (c.kernel_size in ((3, 3), (1, 1)) and c.stride[0] == c.stride[1] and c.groups == 1 and c.padding == (((k - 1) * c.dilation[0]) // 2,) * 2
            and ops.conv_train_supported(x, c.weight, c.stride[0], c.dilation[0])):
        return ops.conv_train(x, c.weight, c.bias, c.dilation[0], slope, c.stride[0])
    return seq(x.float()).to(x.dtype)


# fp32 inference (the parity mode): 'hip_x3' = the split-precision matrix-core kernel (csrc/conv_x3.hip, 3 fp16 products per
# operand pair; 'hip_x3s': with the low-order products in their own accumulators), 'miopen' = PyTorch-ROCm.  Set by UPFlow_net for the duration of a forward (config `fp32_conv`).
FP32_CONV = ['hip_x3']
FP32_CONV_NPROD = {'hip_x3': 3, 'hip_x3s': 11}


class fp32_conv_mode(object):
    def __init__(self, mode):
        if mode != 'miopen' and mode not in FP32_CONV_NPROD:
            raise ValueError("fp32_conv must be one of %s or 'miopen', got %r" % (sorted(FP32_CONV_NPROD), mode))
        self.mode = mode

    def __enter__(self):
        self.saved = (FP32_CONV[0], ops.CONV_X3_NPROD[0])
        FP32_CONV[0] = self.mode
        ops.CONV_X3_NPROD[0] = FP32_CONV_NPROD.get(self.mode, ops.CONV_X3_NPROD[0])

    def __exit__(self, *exc):
        FP32_CONV[0], ops.CONV_X3_NPROD[0] = self.saved


def _fast_conv_ok(t):
    """Inference on the GPU in bf16 / fp16 or — unless fp32_conv = 'miopen' — in fp32, any size: the hand-written MFMA convolutions
    and the copy-free concat buffer apply (16-bit rows shorter than 8 pixels through the split-precision kernel:
    _PackedConv3x3.__call__); otherwise the same arithmetic runs through MIOpen + torch.cat."""
    if torch.is_grad_enabled() or not t.is_cuda:
        return False
    if t.dtype == torch.float32:
        return FP32_CONV[0] != 'miopen'
    return t.dtype in (torch.bfloat16, torch.float16)


class _DenseStack(tools.abstract_model):
    """conv1..conv5 each see everything before them; new features are concatenated in front
    (pwc_modules.py:279-286 / model/upflow.py:53-60).

    Fast path: the growing concatenation lives in ONE buffer laid out
        [conv5 | conv4 | conv3 | conv2 | conv1 | x | tail]
    every conv reads a channel SUFFIX of it and writes its own slice (bias + LeakyReLU fused), so the five
    `torch.cat` copies, the separate activation passes and MIOpen's im2col / layout transposes vanish."""
    _NAMES = ('conv1', 'conv2', 'conv3', 'conv4', 'conv5')

    def _build(self, ch_in, f_channels, out_channel):
        n = ch_in
        for i, f in enumerate(f_channels):
            setattr(self, 'conv%d' % (i + 1), conv(n, f))
            n += f
        self.conv_last = conv(n, out_channel, isReLU=False)
        self._ch_in, self._f, self._n_total = ch_in, tuple(f_channels), n
        self._packed = None
        return n

    def alloc_buffer(self, B, H, W, dtype, device, tail=0):
        """-> (buf [B, n_total + tail, H, W], x_view = the slot of the stack's input)."""
        buf = torch.empty((B, self._n_total + tail, H, W), dtype=dtype, device=device)
        x0 = self._n_total - self._ch_in
        return buf, buf[:, x0:self._n_total]

    def forward_in_buffer(self, buf, out=None):
        """buf from alloc_buffer with the input slot filled -> (x5 view [B, n_total, H, W], x_out)."""
        if self._packed is None:
            self._packed = [_PackedConv3x3(getattr(self, n)) for n in self._NAMES] + [_PackedConv3x3(self.conv_last)]
        nt = self._n_total
        hi = nt - self._ch_in
        for pc, f in zip(self._packed[:5], self._f):
            pc(buf[:, hi:nt], buf[:, hi - f:hi])
            hi -= f
        x5 = buf[:, :nt]
        if out is None:
            out = torch.empty((buf.shape[0], self.conv_last[0].out_channels) + tuple(buf.shape[2:]), dtype=buf.dtype, device=buf.device)
        self._packed[5](x5, out)
        return x5, out

    def c8_ok(self):
        """The whole stack can live in a channel-octet buffer: every width a whole number of octets."""
        return self._ch_in % 8 == 0 and all(f % 8 == 0 for f in self._f)

    def c8_outputs_ok(self):
        """The layers' OUTPUT widths are whole octets (the input part may then be any octet-position map, forward_in_buffer_c8)."""
        return all(f % 8 == 0 for f in self._f)

    def forward_in_buffer_c8(self, buf8, out=None, in_map=None):
        """forward_in_buffer on a channel-octet buffer (ops.c8_empty) whose input octets are filled: the same layout in
        octets, every layer reads an octet range by LDS-DMA and writes its octets straight from the accumulators; conv_last
        writes NCHW planes -> x_out [B, out_channels, H, W].
        in_map None: the input part is the stack's ch_in channels in order (ch_in % 8 == 0), the buffer has n_total / 8 octets.
        in_map: for every POSITION of the input octets the input channel it carries, or -1 for a padding position (which must
        hold finite values: the weights there are zero) — e.g. the flow estimator's [cost volume in the octet order of
        ops.corr81_c8_channel_map | features | flow, flow, 0 x 6].  Octets after the input part (a tail) are not read."""
        key = None if in_map is None else tuple(in_map)
        cache = self.__dict__.setdefault('_packed8', {})
        if key not in cache:
            imap = list(range(self._ch_in)) if in_map is None else list(in_map)
            assert len(imap) % 8 == 0 and sorted(m for m in imap if m >= 0) == list(range(self._ch_in)), 'in_map must place every input channel once'
            packed, nconv = [], 0
            for name, f in zip(self._NAMES, self._f):
                # layer input = [conv_{k-1} | ... | conv1 | x]: identity on the conv outputs, then the input part's map
                packed.append(_PackedConvC8(getattr(self, name), list(range(nconv)) + [m + nconv if m >= 0 else -1 for m in imap]))
                nconv += f
            packed.append(_PackedConvC8(self.conv_last, list(range(nconv)) + [m + nconv if m >= 0 else -1 for m in imap]))
            # the merged narrow tail (_PackedTailC8): the narrow last layers' shared-input part computed by an earlier layer's pass
            widths = list(self._f) + [self.conv_last[0].out_channels]
            seqs = [getattr(self, n_) for n_ in self._NAMES] + [self.conv_last]
            plain = all(all(isinstance(x_, (nn.Conv2d, nn.LeakyReLU)) for x_ in q_) for q_ in seqs)
            pl = _PackedTailC8.plan(widths) if plain else None
            if pl is not None:
                nbefore = sum(self._f[:pl[0]])
                packed.append(_PackedTailC8(seqs, pl[0], pl[1], list(range(nbefore)) + [q + nbefore if q >= 0 else -1 for q in imap]))
            cache[key] = (packed, len(imap) // 8)
        packed, n_in = cache[key]
        nl = len(self._f)
        no = sum(self._f) // 8 + n_in
        starts, hi = [], sum(self._f) // 8                  # layer k reads octets [starts[k], no) and writes [starts[k + 1], starts[k])
        for f in self._f:
            starts.append(hi)
            hi -= f // 8
        starts.append(hi)
        if out is None:
            out = torch.empty((buf8.shape[0], self.conv_last[0].out_channels) + tuple(buf8.shape[2:4]), dtype=buf8.dtype, device=buf8.device)
        tail = packed[nl + 1] if (len(packed) > nl + 1 and MERGE_TAIL[0] and not getattr(self, '_no_merge_tail', False)) else None
        part = None
        for k in range(nl + 1):
            y = buf8[:, starts[k + 1]:starts[k]] if k < nl else out
            if tail is not None and k == tail.m:
                part = tail.main_pass(buf8[:, starts[k]:no], y)
            elif tail is not None and k in tail.J:
                tail.finish_pass(k, buf8[:, starts[k]:starts[tail.m]], part, y)
            else:
                packed[k](buf8[:, starts[k]:no], None, y)
        return out

    def _train_convs(self):
        """conv1..conv5, conv_last as plain nn.Conv2d if every Sequential is Conv2d [+ LeakyReLU] (no norm layers)."""
        seqs = [getattr(self, n) for n in self._NAMES] + [self.conv_last]
        slope = None
        for i, q in enumerate(seqs):
            act = list(q)[1:]
            if i < 5:
                if len(act) != 1 or not isinstance(act[0], nn.LeakyReLU):
                    return None, None
                slope = float(act[0].negative_slope)
            elif act:
                return None, None
        return [q[0] for q in seqs], slope

    def train_in_buffer_ok(self, inputs):
        convs, _ = self._train_convs()
        return convs is not None and not getattr(self, '_no_train_buffer', False) and ops.dense_stack_train_supported(inputs, convs)

    def forward_train(self, inputs, flow_tail=None):
        """Training (16-bit activations, fp32 master weights): the whole stack as ONE autograd node in the inference path's
        buffer layout (ops.DenseStackTrainFunction).  inputs: tensors whose concatenation is the stack's input.
        -> (buf [B, n_total (+ tail), H, W], x_out); buf[:, :n_total] is the reference's x5."""
        convs, slope = self._train_convs()
        return ops.dense_stack_train(inputs, convs, slope, flow_tail)

    def forward(self, x):
        if _fast_conv_ok(x):
            buf, slot = self.alloc_buffer(x.shape[0], x.shape[2], x.shape[3], x.dtype, x.device)
            slot.copy_(x)
            return self.forward_in_buffer(buf)
        if self.train_in_buffer_ok([x]):
            buf, out = self.forward_train([x])
            return buf, out
        cache = self.__dict__.setdefault('_fast_cache', {})
        for name in self._NAMES:
            x = torch.cat([fast_conv_seq(getattr(self, name), x, cache), x], dim=1)
        return x, fast_conv_seq(self.conv_last, x, cache)


class FlowEstimatorDense_v2(_DenseStack):
    def __init__(self, ch_in, f_channels=(128, 128, 96, 64, 32), out_channel=2):
        super(FlowEstimatorDense_v2, self).__init__()
        self.n_channels = self._build(ch_in, f_channels, out_channel)


class FlowEstimatorDense(FlowEstimatorDense_v2):
    """pwc_modules.py:229-247 = the v2 stack with its default widths."""

    def __init__(self, ch_in):
        super(FlowEstimatorDense, self).__init__(ch_in)


class OpticalFlowEstimator(nn.Module):
    """Plain (non-dense) estimator, pwc_modules.py:210-226."""

    def __init__(self, ch_in):
        super(OpticalFlowEstimator, self).__init__()
        self.convs = nn.Sequential(conv(ch_in, 128), conv(128, 128), conv(128, 96), conv(96, 64), conv(64, 32))
        self.conv_last = conv(32, 2, isReLU=False)

    def forward(self, x):
        x_intm = self.convs(x)
        return x_intm, self.conv_last(x_intm)


class ContextNetwork_v2_(nn.Module):
    """Seven 3x3 convs with dilations 1,2,4,8,16,1,1; last one linear. pwc_modules.py:396-412."""

    def __init__(self, ch_in, f_channels=(128, 128, 128, 96, 64, 32, 2)):
        super(ContextNetwork_v2_, self).__init__()
        dil = (1, 2, 4, 8, 16, 1, 1)
        chans = (ch_in,) + tuple(f_channels)
        self.convs = nn.Sequential(*[conv(chans[i], chans[i + 1], 3, 1, dil[i], isReLU=(i < 6)) for i in range(7)])

    def forward(self, x):
        cache = self.__dict__.setdefault('_fast_cache', {})
        for seq in self.convs:                      # the dilation-16 layer falls back to MIOpen inside
            x = fast_conv_seq(seq, x, cache)
        return x

    def forward_c8(self, x, in_map=None):
        """Inference on a large grid (c8_level_ok): the chain in the channel-octet layout — conv0 reads the estimator buffer
        (NCHW planes; or, with in_map, a channel-octet buffer [B, n_oct, H, W, 8] whose position p carries input channel
        in_map[p], -1 = padding) and writes octets, conv1..5 read and write octets (LDS-DMA staging), conv6 writes the two
        NCHW flow planes."""
        key = None if in_map is None else tuple(in_map)
        cache = self.__dict__.setdefault('_packed8', {})
        if key not in cache:
            chans = [s_[0].in_channels for s_ in self.convs]
            first = _PackedConvC8(self.convs[0], (), range(chans[0])) if in_map is None else _PackedConvC8(self.convs[0], list(in_map))
            cache[key] = [first] + [_PackedConvC8(self.convs[i], range(chans[i])) for i in range(1, 7)]
        B, H, W = x.shape[0], x.shape[2], x.shape[3]
        t = None
        for i, pc in enumerate(cache[key]):
            co = self.convs[i][0].out_channels
            y = ops.c8_empty(B, co, H, W, x.dtype, x.device) if i < 6 else torch.empty((B, co, H, W), dtype=x.dtype, device=x.device)
            if i == 0:
                pc(x if in_map is not None else None, None if in_map is not None else x, y)
            else:
                pc(t, None, y)
            t = y
        return t


class ContextNetwork(ContextNetwork_v2_):
    """pwc_modules.py:377-393."""

    def __init__(self, ch_in):
        super(ContextNetwork, self).__init__(ch_in)
