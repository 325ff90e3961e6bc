"""UPFlow_net — drop-in shell of `/root/reference/model/upflow.py` on the MI355X-native operators.

Same class names (`UPFlow_net`, `UPFlow_net.config`, `network_tools.sgu_model`), the same config
flags, the same `net(input_dict) -> output_dict` contract (model/upflow.py:370-392) and the same 80
state_dict keys, so `net.load_model('upflow_kitti2015.pth', if_relax=True)` works as in test.py:33.

What is native here (one hand-written HIP launch each, csrc/*.hip):
    cost volume (+LeakyReLU, written straight into the estimator's 115-channel input buffer),
    backward warp (+validity mask), flow up-sampling (+rescale), SGU interpolation-blend,
    feature normalisation, occlusion check.
Convolutions: fp32 (the parity mode) and everything under autograd stay PyTorch-ROCm (MIOpen); in bf16/fp16
inference the flow-estimator / context / SGU-estimator convolutions run on the hand-written MFMA kernel
(csrc/conv3x3.hip), and so do the feature-pyramid and 1x1 convolutions unless
`hip_pyramid_convs=False` restores the north star's literal split ("the feature-pyramid convolutions stay
PyTorch-ROCm").  Precision: with bf16/fp16 weights (net.bfloat16()) features and conv activations are 16-bit,
while flows, sampling positions, masks, normalisation statistics and all accumulators stay fp32.

Extra (non-reference) config flags, all defaulting to the reference's behaviour:
    warp_mask_mode = 'literal' | 'robust'   validity-mask semantics of the feature warps (§7-H2)
    hip_pyramid_convs = True | False        16-bit inference: feature pyramid on the MFMA kernel / on MIOpen
    train_conv_dtype = 'fp32' | 'bf16' | 'fp16'   training: 'fp32' = the parity mode (every convolution PyTorch-ROCm);
                                            16-bit = decoder activations in that type, fp32 master weights, forward /
                                            data-gradient / weight-gradient of the decoder convolutions on the MFMA kernels
"""
import os
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..utils.tools import tools
from ..utils.loss import loss_functions
from .pwc_modules import (conv, initialize_msra, upsample2d_flow_as, upsample_flow, FlowEstimatorDense_v2,
                          ContextNetwork_v2_, WarpingLayer_no_div, FeatureExtractor, _DenseStack, _fast_conv_ok,
                          fast_conv_seq, c8_level_ok, _PackedConvC8)
from .correlation_package.correlation import Correlation


FLOW16_IN_BLEND = [True]     # experiment / parity switch: False = the estimator's flow slot written by upf_flow_update(_c8) in its own launch (rounds 1-5)


class network_tools():
    class sgu_model(tools.abstract_model):
        """Self-guided upsample module (model/upflow.py:20-92)."""

        def __init__(self, mask_mode='literal'):
            super(network_tools.sgu_model, self).__init__()

            class FlowEstimatorDense_temp(_DenseStack):
                def __init__(self, ch_in, f_channels=(128, 128, 96, 64, 32), ch_out=2):
                    super(FlowEstimatorDense_temp, self).__init__()
                    self.num_feature_channel = self._build(ch_in, f_channels, ch_out)

            self.warping_layer = WarpingLayer_no_div(mask_mode)
            self.dense_estimator_mask = FlowEstimatorDense_temp(64, f_channels=(32, 32, 32, 16, 8), ch_out=3)
            self.upsample_output_conv = nn.Sequential(conv(3, 16, kernel_size=3, stride=1, dilation=1),
                                                      conv(16, 16, stride=2),
                                                      conv(16, 32, kernel_size=3, stride=1, dilation=1),
                                                      conv(32, 32, stride=2), )

        def forward(self, flow_init, feature_1, feature_2, output_level_flow=None, batch_shift=0):
            """-> (flow_init, flow_up, inter_flow, inter_mask), model/upflow.py:71-89.
            batch_shift (not in the reference): both flow directions stacked along the batch, feature_2 is
            then feature_1 itself and item n's "other frame" is item (n + batch_shift) % B."""
            if flow_init.shape[2:] != feature_1.shape[2:]:
                flow_init = upsample2d_flow_as(flow_init, feature_1, mode="bilinear", if_rate=True)
            feature_2_warp = self.warping_layer(feature_2, flow_init, batch_shift)
            est = self.dense_estimator_mask
            if _fast_conv_ok(feature_1):
                # concat-free: both halves of the estimator input are written into its buffer slot
                buf, slot = est.alloc_buffer(feature_1.shape[0], feature_1.shape[2], feature_1.shape[3], feature_1.dtype, feature_1.device)
                slot[:, :feature_1.shape[1]] = feature_1
                slot[:, feature_1.shape[1]:] = feature_2_warp
                _, x_out = est.forward_in_buffer(buf)
            else:
                _, x_out = est(torch.cat((feature_1, feature_2_warp), dim=1))
            # slice + sigmoid + (up-sampling) + torch_warp + blend: ONE launch (csrc/sgu_blend.hip)
            return ops.sgu_blend(flow_init, x_out, output_level_flow)

        def forward_in_buffer(self, flow_init, buf, slot, output_level_flow=None, batch_shift=0, flow16=None):
            """Inference fast path of forward(): `slot` (the estimator's input slot of `buf`, from
            dense_estimator_mask.alloc_buffer) already holds feature_1 in its first half — written there by the conv
            that produced it; the other frame's features are warped straight into the second half."""
            c1 = slot.shape[1] // 2
            ops.warp_into(slot[:, :c1], flow_init, slot[:, c1:], self.warping_layer.mask_mode, batch_shift)
            _, x_out = self.dense_estimator_mask.forward_in_buffer(buf)
            if flow16 is not None:
                # (round 6) flow16: the flow slot of the flow estimator's input buffer — the blend stores the rounded flow there itself
                return flow_init, ops.sgu_blend_flow16(flow_init, x_out, flow16), None, None
            return ops.sgu_blend(flow_init, x_out, output_level_flow, want_inter=False)   # (flow_init, flow_up, None, None)

        def forward_in_buffer_c8(self, flow_init, buf8, output_level_flow=None, batch_shift=0, tap=None, flow16=None):
            """forward_in_buffer with the estimator's buffer in the channel-octet layout (pwc_modules.c8_level_ok): feature_1 is
            already in its octets (written there by the 1x1 convolution); the other frame's features are warped from those
            octets into the second half, the stack runs on octets, x_out comes back as NCHW planes for the blend."""
            est = self.dense_estimator_mask
            o0 = (est._n_total - est._ch_in) // 8
            half = est._ch_in // 16
            ops.warp_c8_into(buf8[:, o0:o0 + half], flow_init, buf8[:, o0 + half:o0 + 2 * half], self.warping_layer.mask_mode, batch_shift)
            x_out = est.forward_in_buffer_c8(buf8)
            if tap is not None:
                tap('sgu_x_out', x_out)
            if flow16 is not None:
                return flow_init, ops.sgu_blend_flow16(flow_init, x_out, flow16), None, None
            return ops.sgu_blend(flow_init, x_out, output_level_flow, want_inter=False)

        def output_conv(self, x, out=None, out8=None, pitched=False):
            """out8: octet slice of a channel-octet buffer the LAST layer writes instead of `out` (forward_in_buffer_c8).
            pitched: intermediates with 16-byte aligned rows at ragged widths (ops.empty_nchw)."""
            cache = self.__dict__.setdefault('_fast_cache', {})
            n = len(self.upsample_output_conv)
            from . import pwc_modules as _pm
            if n == 4 and _pm.FUSE_PAIRS[0] and not getattr(self, '_no_fuse_pairs', False) and all(len(q) == 2 for q in self.upsample_output_conv):
                # (round 6) [3x3, 3x3 stride 2] x 2 as TWO launches whose intermediates stay in LDS (csrc/conv_pair.hip): the full-resolution
                # 16-channel and the half-resolution 32-channel activations (126 + 63 MB at 384x1280) are never written
                sq = self.upsample_output_conv
                if ops.conv_pair_supported(x, sq[0][0], sq[1][0]):
                    B_, _, H_, W_ = x.shape
                    h1, w1 = ops.conv3x3_out_hw(H_, W_, 2)
                    mid = ops.empty_nchw((B_, sq[1][0].out_channels, h1, w1), x.dtype, x.device, pitched=True)
                    if ops.conv_pair_supported(mid, sq[2][0], sq[3][0]):
                        pa = cache.get('pair01') or cache.setdefault('pair01', _pm._PackedConvPair(sq[0], sq[1]))
                        pb = cache.get('pair23') or cache.setdefault('pair23', _pm._PackedConvPair(sq[2], sq[3]))
                        pa(x, mid)
                        h2, w2 = ops.conv3x3_out_hw(h1, w1, 2)
                        if out8 is not None:
                            return pb(mid, out8)
                        y = out if out is not None else ops.empty_nchw((B_, sq[3][0].out_channels, h2, w2), x.dtype, x.device, pitched=pitched)
                        return pb(mid, y)
            for i, seq in enumerate(self.upsample_output_conv):       # matrix-core kernel when eligible
                if i == n - 1 and out8 is not None:
                    pc = cache.get('c8_last')
                    if pc is None:
                        pc = cache['c8_last'] = _PackedConvC8(seq, (), range(seq[0].in_channels))
                    return pc(None, x, out8)
                x = fast_conv_seq(seq, x, cache, out=out if i == n - 1 else None, pitched=pitched)
            return x

    @classmethod
    def normalize_features(cls, feature_list, normalize, center, moments_across_channels=True, moments_across_images=True):
        """model/upflow.py:94-137.  The flag combination the published model uses (per-channel,
        per-image statistics, center and normalize: test.py:24-26) is one fused HIP launch per
        tensor; other combinations use the same arithmetic spelled in torch ops."""
        if normalize and center and not moments_across_channels and not moments_across_images:
            return [ops.normalize(f) for f in feature_list]
        axes = [1, 2, 3] if moments_across_channels else [2, 3]
        means = [torch.mean(f.float(), dim=axes, keepdim=True) for f in feature_list]
        vars_ = [torch.var(f.float(), dim=axes, keepdim=True) for f in feature_list]
        if moments_across_images:
            means = [torch.mean(torch.stack(means, dim=0), dim=(0,))] * len(feature_list)
            vars_ = [torch.var(torch.stack(vars_, dim=0), dim=(0,))] * len(feature_list)      # sic, upflow.py:124
        stds = [torch.sqrt(v + 1e-16) for v in vars_]
        out = list(feature_list)
        if center:
            out = [f - m.to(f.dtype) for f, m in zip(out, means)]
        if normalize:
            out = [f / s.to(f.dtype) for f, s in zip(out, stds)]
        return out

    # ---- loss terms (training only; plain torch ops) --------------------------------------------
    @classmethod
    def edge_aware_smoothness_order1(cls, img, pred):
        """model/upflow.py:197-216 — one fused reduction launch (csrc/loss.hip: upf_smooth_edge1_*)."""
        return ops.smooth_edge1(img, pred)

    @classmethod
    def edge_aware_smoothness_order2(cls, img, pred):
        """model/upflow.py:218-243."""
        def dx(t, s=1):
            return t[:, :, :-s, :] - t[:, :, s:, :]

        def dy(t, s=1):
            return t[:, :, :, :-s] - t[:, :, :, s:]
        wx = torch.exp(-dx(img, 2).abs().mean(1, keepdim=True))
        wy = torch.exp(-dy(img, 2).abs().mean(1, keepdim=True))
        return (dx(dx(pred)).abs() * wx).mean() + (dy(dy(pred)).abs() * wy).mean()

    @classmethod
    def flow_smooth_delta(cls, flow, if_second_order=False):
        return loss_functions.flow_smooth_delta(flow, if_second_order)

    @classmethod
    def weighted_ssim(cls, x, y, weight, c1=float('inf'), c2=9e-6, weight_epsilon=0.01):
        """model/upflow.py:139-195."""
        if c1 == float('inf') and c2 == float('inf'):
            raise ValueError('Both c1 and c2 are infinite, SSIM loss is zero. This is likely unintended.')

        def pool(z):
            return F.avg_pool2d(z, (3, 3), (1, 1))
        w_avg = pool(weight)
        w_eps = weight + weight_epsilon
        inv = 1.0 / (w_avg + weight_epsilon)

        def wpool(z):
            return pool(z * w_eps) * inv
        mu_x, mu_y = wpool(x), wpool(y)
        sigma_x = wpool(x ** 2) - mu_x ** 2
        sigma_y = wpool(y ** 2) - mu_y ** 2
        sigma_xy = wpool(x * y) - mu_x * mu_y
        if c1 == float('inf'):
            n, d = (2 * sigma_xy + c2), (sigma_x + sigma_y + c2)
        elif c2 == float('inf'):
            n, d = 2 * mu_x * mu_y + c1, mu_x ** 2 + mu_y ** 2 + c1
        else:
            n = (2 * mu_x * mu_y + c1) * (2 * sigma_xy + c2)
            d = (mu_x ** 2 + mu_y ** 2 + c1) * (sigma_x + sigma_y + c2)
        return torch.clamp((1 - n / d) / 2, 0, 1), w_avg

    @classmethod
    def photo_loss_multi_type(cls, x, y, occ_mask, photo_loss_type='abs_robust', photo_loss_delta=0.4, photo_loss_use_occ=False):
        """model/upflow.py:265-288."""
        occ_weight = occ_mask
        if photo_loss_type == 'abs_robust':
            # sub / abs / add / pow / mul / sum of the reference as ONE deterministic reduction (csrc/loss.hip)
            s, s_occ = ops.robust_loss_sums(x, y, occ_mask if photo_loss_use_occ else None, q=photo_loss_delta, eps=0.01)
            return s / (s_occ + 1e-6) if photo_loss_use_occ else s / float(x.numel())
        elif photo_loss_type == 'charbonnier':
            loss_diff = ((x - y) ** 2 + 1e-6).pow(photo_loss_delta)
        elif photo_loss_type == 'L1':
            loss_diff = (x - y + 1e-6).abs()
        elif photo_loss_type == 'SSIM':
            loss_diff, occ_weight = cls.weighted_ssim(x, y, occ_mask)
        else:
            raise ValueError('wrong photo_loss type: %s' % photo_loss_type)
        if photo_loss_use_occ:
            return torch.sum(loss_diff * occ_weight) / (torch.sum(occ_weight) + 1e-6)
        return torch.mean(loss_diff)


class UPFlow_net(tools.abstract_model):
    class config(tools.abstract_config):
        def __init__(self):
            # defaults of model/upflow.py:293-323
            self.occ_type = 'for_back_check'
            self.alpha_1 = 0.1
            self.alpha_2 = 0.5
            self.occ_check_obj_out_all = 'obj'
            self.stop_occ_gradient = False
            self.smooth_level = 'final'
            self.smooth_type = 'edge'
            self.smooth_order_1_weight = 1
            self.smooth_order_2_weight = 0
            self.photo_loss_type = 'abs_robust'
            self.photo_loss_delta = 0.4
            self.photo_loss_use_occ = False
            self.photo_loss_census_weight = 0
            self.if_norm_before_cost_volume = False
            self.norm_moments_across_channels = True
            self.norm_moments_across_images = True
            self.multi_scale_distillation_weight = 0
            self.multi_scale_distillation_style = 'upup'
            self.multi_scale_distillation_occ = True
            self.if_froze_pwc = False
            self.input_or_sp_input = 1
            self.if_use_boundary_warp = True
            self.if_sgu_upsample = False
            self.if_use_cor_pytorch = False
            # --- not in the reference; defaults keep its behaviour
            self.warp_mask_mode = 'literal'
            self.hip_pyramid_convs = True
            self.train_conv_dtype = 'fp32'          # 'bf16' / 'fp16': decoder convolutions under autograd on the matrix cores
            self.fp32_conv = 'hip_x3'               # fp32 inference: 'hip_x3' / 'hip_x3s' split-precision MFMA kernel, 'miopen' PyTorch-ROCm
            self.fp16_overflow_check = False        # fp16 features / activations: raise if an output is not finite (one reduction + a host sync per forward)

        def __call__(self, ):
            return UPFlow_net(self)

    def __init__(self, conf: config):
        super(UPFlow_net, self).__init__()
        self.conf = conf
        if conf.if_use_cor_pytorch:
            raise ops.UpflowHipError(
                "if_use_cor_pytorch=True selects the reference's CPU fallback (utils/pytorch_correlation.py); "
                "this package has no CPU path — its restatement lives in oracle/ as test infrastructure")
        # same construction order as model/upflow.py:335-361 (=> same state_dict key order)
        self.search_range = 4
        self.num_chs = [3, 16, 32, 64, 96, 128, 196]
        self.estimator_f_channels = (128, 128, 96, 64, 32)
        self.context_f_channels = (128, 128, 128, 96, 64, 32, 2)
        self.output_level = 4
        self.num_levels = 7
        self.leakyRELU = nn.LeakyReLU(0.1, inplace=True)
        self.feature_pyramid_extractor = FeatureExtractor(self.num_chs)
        self.feature_pyramid_extractor.hip_convs = bool(getattr(conf, 'hip_pyramid_convs', True))
        self.warping_layer = WarpingLayer_no_div(conf.warp_mask_mode)
        self.dim_corr = (self.search_range * 2 + 1) ** 2
        self.num_ch_in = self.dim_corr + 32 + 2
        self.flow_estimators = FlowEstimatorDense_v2(self.num_ch_in, f_channels=self.estimator_f_channels)
        self.context_networks = ContextNetwork_v2_(self.flow_estimators.n_channels + 2, f_channels=self.context_f_channels)
        self.conv_1x1 = nn.ModuleList([conv(c, 32, kernel_size=1, stride=1, dilation=1) for c in (196, 128, 96, 64, 32)])
        self.correlation = Correlation(pad_size=self.search_range, kernel_size=1, max_displacement=self.search_range,
                                       stride1=1, stride2=1, corr_multiply=1)
        self.sgi_model = network_tools.sgu_model(conf.warp_mask_mode) if conf.if_sgu_upsample else None
        self.occ_check_model = tools.occ_check_model(occ_type=conf.occ_type, occ_alpha_1=conf.alpha_1,
                                                     occ_alpha_2=conf.alpha_2, obj_out_all=conf.occ_check_obj_out_all)
        initialize_msra(self.modules())
        if conf.if_froze_pwc:
            self.froze_PWC()

    # -------------------------------------------------------------------------------------------
    def forward(self, input_dict: dict):
        """input_dict: im1, im2, if_loss [, im1_raw, im2_raw, start, im1_sp, im2_sp]
        -> output_dict: flow_f_out, flow_b_out, occ_fw, occ_bw [, smooth_loss, photo_loss, im1_warp,
        im2_warp, census_loss, msd_loss]          (model/upflow.py:370-492)"""
        im1_ori, im2_ori = input_dict['im1'], input_dict['im2']
        if input_dict['if_loss'] and self.conf.input_or_sp_input != 1:
            im1, im2 = input_dict['im1_sp'], input_dict['im2_sp']
        else:
            im1, im2 = im1_ori, im2_ori
        flow_f, flow_b, flows = self.forward_2_frame_v3(im1, im2, if_loss=input_dict['if_loss'])
        occ_fw, occ_bw = self.occ_check_model(flow_f=flow_f, flow_b=flow_b)
        out = {'flow_f_out': flow_f, 'flow_b_out': flow_b, 'occ_fw': occ_fw, 'occ_bw': occ_bw}
        if getattr(self.conf, 'fp16_overflow_check', False) and not torch.cuda.is_current_stream_capturing():
            # fp16 stores saturate to infinity beyond 65504 (bf16 does not): an overflowing feature or activation reaches the flows as
            # inf / NaN — every operator here propagates non-finite values — so ONE check of the outputs guards the whole forward
            if not (bool(torch.isfinite(flow_f).all()) and bool(torch.isfinite(flow_b).all())):
                raise ops.UpflowHipError('non-finite flow: an fp16 feature / activation overflowed (|x| > 65504) — run this input in '
                                         'bfloat16 (net.to_inference(torch.bfloat16)) or fp32')
        if input_dict['if_loss']:
            self._losses(input_dict, out, flows, im1_ori.float(), im2_ori.float())
        return out

    def _losses(self, input_dict, out, flows, im1, im2):
        """Unsupervised losses, model/upflow.py:394-491 (plain torch ops on top of the HIP warps)."""
        c = self.conf
        nt = network_tools
        flow_f, flow_b, occ_fw, occ_bw = out['flow_f_out'], out['flow_b_out'], out['occ_fw'], out['occ_bw']
        # smoothness
        if c.smooth_level == 'final':
            s_f, s_b, s_im1, s_im2 = flow_f, flow_b, im1, im2
        elif c.smooth_level == '1/4':
            s_f, s_b = flows[0]
            s_im1 = F.interpolate(im1, s_f.shape[2:], mode='area')
            s_im2 = F.interpolate(im2, s_f.shape[2:], mode='area')
        else:
            raise ValueError('wrong smooth level choosed: %s' % c.smooth_level)
        smooth = 0
        for order, weight in ((1, c.smooth_order_1_weight), (2, c.smooth_order_2_weight)):
            if weight <= 0:
                continue
            if c.smooth_type == 'edge':
                fn = nt.edge_aware_smoothness_order1 if order == 1 else nt.edge_aware_smoothness_order2
                smooth = smooth + weight * fn(img=s_im1, pred=s_f) + weight * fn(img=s_im2, pred=s_b)
            elif c.smooth_type == 'delta':
                smooth = smooth + weight * nt.flow_smooth_delta(s_f, order == 2) + weight * nt.flow_smooth_delta(s_b, order == 2)
            else:
                raise ValueError('wrong smooth_type: %s' % c.smooth_type)
        out['smooth_loss'] = smooth
        # photometric
        if c.if_use_boundary_warp:
            im1_s, im2_s, start = input_dict['im1_raw'].float(), input_dict['im2_raw'].float(), input_dict['start']
            im1_warp = tools.boundary_dilated_warp.warp_im(im2_s, flow_f, start)
            im2_warp = tools.boundary_dilated_warp.warp_im(im1_s, flow_b, start)
        else:
            im1_warp = tools.torch_warp(im2, flow_f)
            im2_warp = tools.torch_warp(im1, flow_b)
        if c.stop_occ_gradient:
            occ_fw, occ_bw = occ_fw.clone().detach(), occ_bw.clone().detach()
        kw = dict(photo_loss_type=c.photo_loss_type, photo_loss_delta=c.photo_loss_delta, photo_loss_use_occ=c.photo_loss_use_occ)
        out['photo_loss'] = nt.photo_loss_multi_type(im1, im1_warp, occ_fw, **kw) + nt.photo_loss_multi_type(im2, im2_warp, occ_bw, **kw)
        out['im1_warp'], out['im2_warp'] = im1_warp, im2_warp
        # census
        if c.photo_loss_census_weight > 0:
            ckw = dict(q=c.photo_loss_delta, charbonnier_or_abs_robust=False, if_use_occ=c.photo_loss_use_occ, averge=True)
            census = loss_functions.census_loss_torch(img1=im1, img1_warp=im1_warp, mask=occ_fw, **ckw) + \
                loss_functions.census_loss_torch(img1=im2, img1_warp=im2_warp, mask=occ_bw, **ckw)
            out['census_loss'] = census * c.photo_loss_census_weight
        else:
            out['census_loss'] = None
        # pyramid distillation (model/upflow.py:461-487): detached final flow teaches every level
        if c.multi_scale_distillation_weight > 0:
            label_f, label_b = flow_f.detach(), flow_b.detach()
            lv_f, lv_b = [f for f, _ in flows], [b for _, b in flows]
            if (c.multi_scale_distillation_style == 'upup' and not getattr(self, '_no_fused_msd', False) and not os.environ.get('UPF_NO_FUSED_MSD') and hasattr(ops, 'msd_upup_supported')
                    and ops.msd_upup_supported(label_f, lv_f) and ops.msd_upup_supported(label_b, lv_b)):
                # one pass over the label per direction instead of up-sample + robust sum + scalar kernels per level (ops.MsdUpupFunction)
                use = c.multi_scale_distillation_occ
                out['msd_loss'] = (ops.msd_upup_loss(lv_f, label_f, occ_fw if use else None, c.multi_scale_distillation_weight)
                                   + ops.msd_upup_loss(lv_b, label_b, occ_bw if use else None, c.multi_scale_distillation_weight))
                return
            label_f, label_b = label_f.clone(), label_b.clone()
            terms = []
            for lvl_f, lvl_b in flows:
                if c.multi_scale_distillation_style == 'down':
                    tf_, tb_ = upsample_flow(label_f, target_flow=lvl_f), upsample_flow(label_b, target_flow=lvl_b)
                    of_ = F.interpolate(occ_fw, lvl_f.shape[2:], mode='nearest')
                    ob_ = F.interpolate(occ_bw, lvl_b.shape[2:], mode='nearest')
                elif c.multi_scale_distillation_style == 'upup':
                    tf_, tb_, of_, ob_ = label_f, label_b, occ_fw, occ_bw
                    lvl_f, lvl_b = upsample_flow(lvl_f, target_flow=label_f), upsample_flow(lvl_b, target_flow=label_b)
                else:
                    raise ValueError('wrong multi_scale_distillation_style: %s' % c.multi_scale_distillation_style)
                mkw = dict(photo_loss_type='abs_robust', photo_loss_use_occ=c.multi_scale_distillation_occ)
                terms.append(nt.photo_loss_multi_type(x=lvl_f, y=tf_, occ_mask=of_, **mkw))
                terms.append(nt.photo_loss_multi_type(x=lvl_b, y=tb_, occ_mask=ob_, **mkw))
            out['msd_loss'] = c.multi_scale_distillation_weight * sum(terms)
        else:
            out['msd_loss'] = None

    # -------------------------------------------------------------------------------------------
    def forward_2_frame_v3(self, x1_raw, x2_raw, if_loss=False):
        """Coarse-to-fine bidirectional decode, model/upflow.py:494-533."""
        cdt = self.feature_pyramid_extractor.convs[0][0][0].weight.dtype      # compute dtype of the convs (of the PYRAMID: to_inference)
        from .pwc_modules import fp32_conv_mode
        with fp32_conv_mode(getattr(self.conf, 'fp32_conv', 'hip_x3')):
            return self._forward_2_frame_v3(x1_raw, x2_raw, if_loss, cdt)

    def _forward_2_frame_v3(self, x1_raw, x2_raw, if_loss, cdt):
        if not torch.is_grad_enabled() or getattr(self, 'stacked_training', True):
            # (training too: every operator on this path is per-item and differentiable — both directions as one
            # batch halve the launch count and double every convolution's batch; `stacked_training = False` restores
            # the reference's per-direction schedule below)
            B = x1_raw.shape[0]
            # (16-bit inference: rows pitched to 16 bytes when the frame width is ragged — KITTI's native 1242 — so that the stem
            # and the first pyramid stage take the aligned kernels; ops.empty_nchw.  `_no_pitch = True`: contiguous everywhere)
            X = ops.empty_nchw((2 * B,) + tuple(x1_raw.shape[1:]), cdt, x1_raw.device,
                               pitched=self._pitched() and not torch.is_grad_enabled() and x1_raw.is_cuda)
            X[:B].copy_(x1_raw)                         # cast + stack in one pass per frame (was: two casts, then a cat)
            X[B:].copy_(x2_raw)
            ddt = self.flow_estimators.conv1[0].weight.dtype
            if ddt != cdt:
                # `pyramid_dtype` (to_inference): the decoder runs in another 16-bit type; its one reader of the frames, the SGU's
                # guidance stem, gets its own cast of them
                if torch.is_grad_enabled() or not x1_raw.is_cuda or ddt == torch.float32 or cdt == torch.float32:
                    raise ops.UpflowHipError('pyramid_dtype: 16-bit GPU inference only')
                Xd = ops.empty_nchw(tuple(X.shape), ddt, X.device, pitched=self._pitched())
                Xd[:B].copy_(x1_raw)
                Xd[B:].copy_(x2_raw)
                return self._forward_stacked(X, B, frames_dec=Xd)
            tdt = {'bf16': torch.bfloat16, 'fp16': torch.float16}.get(getattr(self.conf, 'train_conv_dtype', 'fp32'))
            if tdt is not None and torch.is_grad_enabled() and cdt == torch.float32:
                # training on the matrix cores: 16-bit activations from the first layer on, fp32 master weights
                # (flows, sampling positions, masks, statistics and the losses stay fp32)
                # (the decoder's convolutions are shared by the pyramid levels: their parameter gradients are deferred to
                # one multi-level contraction per layer, ops.shared_conv_grads)
                convs = [m for m in self.modules() if isinstance(m, nn.Conv2d)] if getattr(self, 'shared_grad_sinks', True) else []
                with ops.shared_conv_grads(convs):
                    return self._forward_stacked(X.to(tdt), B, tdt)
            return self._forward_stacked(X, B)
        x1_raw = x1_raw.to(cdt)
        x2_raw = x2_raw.to(cdt)
        x1_pyramid = self.feature_pyramid_extractor(x1_raw)
        x2_pyramid = self.feature_pyramid_extractor(x2_raw)
        B, _, h0, w0 = x1_pyramid[0].shape
        flow_f = torch.zeros(B, 2, h0, w0, dtype=torch.float32, device=x1_raw.device)
        flow_b = torch.zeros_like(flow_f)
        flows = []
        for level in range(self.output_level + 1):
            x1, x2 = x1_pyramid[level], x2_pyramid[level]
            flow_f, flow_b, res_f, res_b = self.decode_level_res(
                level=level, flow_1=flow_f, flow_2=flow_b, feature_1=x1, feature_1_1x1=self.conv_1x1[level](x1),
                feature_2=x2, feature_2_1x1=self.conv_1x1[level](x2), img_ori_1=x1_raw, img_ori_2=x2_raw)
            flow_f = flow_f + res_f
            flow_b = flow_b + res_b
            flows.append([flow_f, flow_b])
        flow_f_out = upsample2d_flow_as(flow_f, x1_raw, mode="bilinear", if_rate=True)
        flow_b_out = upsample2d_flow_as(flow_b, x1_raw, mode="bilinear", if_rate=True)
        if self.conf.if_sgu_upsample:
            g1 = self.sgi_model.output_conv(x1_raw)
            g2 = self.sgi_model.output_conv(x2_raw)
            flow_f_out = self.self_guided_upsample(flow_up_bilinear=flow_f, feature_1=g1, feature_2=g2, output_level_flow=flow_f_out)
            flow_b_out = self.self_guided_upsample(flow_up_bilinear=flow_b, feature_1=g2, feature_2=g1, output_level_flow=flow_b_out)
        return flow_f_out, flow_b_out, flows[::-1]

    def _forward_stacked(self, X, B, train_dtype=None, frames_dec=None):
        """Inference form of forward_2_frame_v3 (model/upflow.py:494-533), same arithmetic, different schedule:
        the two frames are stacked along the batch, X = [im1; im2], so item n < B carries the forward direction
        and item n >= B the backward one.  Every stage then runs ONCE on 2B items with shared weights — feature
        pyramid, 1x1 convs, SGU, warp (batch_shift = B samples "the other frame" without a gather copy),
        normalisation, cost volume, estimator, context network — halving the launch count and doubling every
        grid, which is what the coarse levels need on a 256-CU chip."""

        if (_fast_conv_ok(X) and X.dtype != torch.float32 and self.conf.if_norm_before_cost_volume and not self.conf.norm_moments_across_channels
                and not self.conf.norm_moments_across_images and not getattr(self, '_no_fast_stacked', False)
                and self.feature_pyramid_extractor.out_shapes(X.shape[2], X.shape[3])[-1][2] >= 8):   # every level takes the conv kernel
            return self._forward_stacked_fast(X, B, frames_dec)
        # (generic schedule with `pyramid_dtype`: the pyramid, the warp and the normalisation run in the pyramid's type; the 1x1 features
        # and the NORMALISED features are cast to the decoder's on their way into it — the in-buffer schedule above does it without casts)
        ddt = None if frames_dec is None else frames_dec.dtype
        pyramid = self.feature_pyramid_extractor(X)
        h0, w0 = pyramid[0].shape[2:]
        flow = torch.zeros(2 * B, 2, h0, w0, dtype=torch.float32, device=X.device)
        sgu = self.conf.if_sgu_upsample
        flows = []
        for level in range(self.output_level + 1):
            Fm = pyramid[level]
            A = fast_conv_seq(self.conv_1x1[level], Fm, self.__dict__.setdefault('_fast_cache', {}))
            if ddt is not None:
                A = A.to(ddt)
            flow_up = upsample2d_flow_as(flow, Fm, mode="bilinear", if_rate=True)
            if level == 0:
                Fw = torch.roll(Fm, shifts=B, dims=0)                        # no warp at the coarsest level (:539-541)
            else:
                if sgu:
                    flow_up = self.sgi_model(flow_up, A, A, batch_shift=B)[1]
                Fw = self.warping_layer(Fm, flow_up, batch_shift=B)
            if self.conf.if_norm_before_cost_volume:
                kw = dict(normalize=True, center=True, moments_across_channels=self.conf.norm_moments_across_channels,
                          moments_across_images=self.conf.norm_moments_across_images)
                if self.conf.norm_moments_across_images:                     # statistics shared inside each (f, f_warp) pair
                    n1, n2w = network_tools.normalize_features((Fm[:B], Fw[:B]), **kw)
                    n2, n1w = network_tools.normalize_features((Fm[B:], Fw[B:]), **kw)
                    Fn, Fwn = torch.cat([n1, n2], 0), torch.cat([n2w, n1w], 0)
                else:
                    Fn, Fwn = network_tools.normalize_features((Fm, Fw), **kw)
            else:
                Fn, Fwn = Fm, Fw
            if ddt is not None:
                Fn, Fwn = Fn.to(ddt), Fwn.to(ddt)
            flow = self._level_update(Fn, Fwn, A, flow_up, add_to_flow=True)
            flows.append(list(ops.split_batch(flow, B)))
        flow_out = upsample2d_flow_as(flow, X, mode="bilinear", if_rate=True)
        if sgu:
            G = self.sgi_model.output_conv(X if ddt is None else frames_dec)
            flow_out = self.sgi_model(flow, G, G, output_level_flow=flow_out, batch_shift=B)[1]
        f_out, b_out = ops.split_batch(flow_out, B)
        return f_out, b_out, flows[::-1]

    def _pitched(self):
        return not getattr(self, '_no_pitch', False)

    def _tap(self, name, t):
        """Debug tap (tools/pipe_debug3.py): with `net._taps = []` every named intermediate BUFFER of the fast schedule is kept
        (a reference, no copy, no launch) so that two runs can be compared tensor by tensor."""
        taps = getattr(self, '_taps', None)
        if taps is not None:
            taps.append((name, t))

    def _forward_stacked_fast(self, X, B, frames_dec=None):
        """_forward_stacked for bf16/fp16 with the published normalisation flags: same arithmetic, and every
        intermediate is produced IN the buffer its consumer reads — the pyramid's convs write the per-level
        [features; warped other frame] pair buffers that one normalisation launch pair covers, the 1x1 convs write
        the estimator / SGU input slots, the warps write their slots, and the per-level flow bookkeeping is one
        launch per sum (ops.flow_update).  No slot copies, no convert/add chains: ~40 fewer launches per step."""
        nb = 2 * B
        dev, pdt = X.device, X.dtype                     # pdt: the pyramid's type (features, warped features)
        dt = pdt if frames_dec is None else frames_dec.dtype      # dt: the decoder's (every estimator / context / SGU buffer)
        cache = self.__dict__.setdefault('_fast_cache', {})
        fpe, est, sgi = self.feature_pyramid_extractor, self.flow_estimators, self.sgi_model
        sgu = self.conf.if_sgu_upsample
        nlev = self.output_level + 1
        nc = self.dim_corr
        shapes = fpe.out_shapes(X.shape[2], X.shape[3])[::-1]                 # coarsest first, like the pyramid
        pit = self._pitched()
        # [features; warped other frame] per level; rows pitched to 16 bytes at ragged levels (ops.empty_nchw): every consumer —
        # the next pyramid stage, the 1x1 convolution, the warp, the statistics + cost volume — is pitch-aware
        pairs = [ops.empty_nchw((2, nb) + shp, pdt, dev, pitched=pit) for shp in shapes[:nlev]]
        outs = ([p[0] for p in pairs] + [None] * (len(shapes) - nlev))[::-1]   # stage order: finest first
        pyramid = fpe(X, outs=outs, pitched=pit)
        self._tap('X', X)
        for i_, p_ in enumerate(pyramid):
            self._tap('pyramid%d' % i_, p_)
        flow = torch.zeros((nb, 2) + shapes[0][1:], dtype=torch.float32, device=dev)
        flows = []
        for level in range(nlev):
            Fm, pair = pyramid[level], pairs[level]
            C, H, W = shapes[level]
            use_sgu = sgu and level > 0
            # large grids: the SGU stack and the context network run in the channel-octet layout (same arithmetic, same
            # summation order: bit-identical outputs; `_no_c8 = True` keeps NCHW everywhere)
            # (the octet kernels take any W; the NCHW tensors feeding them need 16-byte aligned rows: W % 8 == 0 or pitched buffers)
            c8 = c8_level_ok(nb, H, W, dt) and (W % 8 == 0 or (pit and W >= 8)) and not getattr(self, '_no_c8', False)
            # ... and so does the flow estimator when the cost volume can write octets (`_no_c8_est = True`: estimator in NCHW).
            # Its K order differs from the NCHW kernel's (81 cost-volume channels in 11 octets, flows padded to an octet), so
            # this part agrees with the NCHW path to fp32 summation order, not bit for bit.
            c8_est = (c8 and level > 0 and est.c8_outputs_ok() and not getattr(self, '_no_c8_est', False)
                      and ops.corr81_norm_supported(pair) and not getattr(self, '_no_fused_norm', False) and nc == 81
                      and est._ch_in == nc + 32 + 2 and self.conv_1x1[level][0].out_channels == 32
                      and self.context_networks.convs[0][0].in_channels == est._n_total + 2)
            if c8_est:
                flow, flows_entry = self._level_c8(level, Fm, pair, flow, nb, B, C, H, W, dt, dev, cache, use_sgu)
                flows.append(flows_entry)
                self._tap('L%d.flow' % level, flow)
                continue
            buf, slot = est.alloc_buffer(nb, H, W, dt, dev, tail=2)
            sbuf8 = None
            if use_sgu and c8 and sgi.dense_estimator_mask.c8_ok() and not getattr(self, '_no_c8_sgu', False):
                em = sgi.dense_estimator_mask
                sbuf8 = ops.c8_empty(nb, em._n_total, H, W, dt, dev)
                o0 = (em._n_total - em._ch_in) // 8
                pc = cache.get(('c8_1x1', level))
                if pc is None:
                    pc = cache[('c8_1x1', level)] = _PackedConvC8(self.conv_1x1[level], (), range(C))
                pc(None, Fm, sbuf8[:, o0:o0 + 4])                               # feature_1 as octets for the SGU stack
                fast_conv_seq(self.conv_1x1[level], Fm, cache, out=slot[:, nc:nc + 32])   # ... and as planes for the estimator
            elif use_sgu:
                sbuf, sslot = sgi.dense_estimator_mask.alloc_buffer(nb, H, W, dt, dev)
                A = fast_conv_seq(self.conv_1x1[level], Fm, cache, out=sslot[:, :32])
                slot[:, nc:nc + 32].copy_(A)
            else:
                fast_conv_seq(self.conv_1x1[level], Fm, cache, out=slot[:, nc:nc + 32])
            # (level 0: the initial zero flow already has the coarsest size — the resize is the identity: no launch)
            flow_up = flow if (level == 0 and tuple(flow.shape[2:]) == (H, W)) else upsample2d_flow_as(flow, Fm, mode="bilinear", if_rate=True)
            f16 = None
            if level == 0:                                                    # no warp at the coarsest level (:539-541)
                pair[1, :B].copy_(Fm[B:])
                pair[1, B:].copy_(Fm[:B])
            else:
                f16 = slot[:, nc + 32:] if (use_sgu and FLOW16_IN_BLEND[0]) else None
                if sbuf8 is not None:
                    flow_up = sgi.forward_in_buffer_c8(flow_up, sbuf8, batch_shift=B, flow16=f16)[1]
                elif use_sgu:
                    flow_up = sgi.forward_in_buffer(flow_up, sbuf, sslot, batch_shift=B, flow16=f16)[1]
                ops.warp_into(Fm, flow_up, pair[1], self.warping_layer.mask_mode, B)
            if ops.corr81_norm_supported(pair) and not getattr(self, '_no_fused_norm', False):
                # statistics pass + cost volume whose loader normalises: the normalised maps are never materialised
                ops.corr81_norm_forward_raw(pair[0], pair[1], out=slot[:, :nc], leaky_slope=0.1)
            else:
                normed = ops.normalize(pair.reshape(2 * nb, C, H, W))         # rows are (item, channel): one launch pair
                ops.corr81_forward_raw(normed[:nb], normed[nb:], out=slot[:, :nc], leaky_slope=0.1)
            if f16 is None:
                ops.flow_update(flow_up, out=slot[:, nc + 32:])
            _, res = est.forward_in_buffer(buf)
            ops.flow_update(flow_up, res, out=buf[:, est._n_total:])          # flow_up + res -> context network input
            # (forward_c8's first layer reads `buf` as NCHW planes: a contiguous buffer has aligned rows only for W % 8 == 0)
            fine = self.context_networks.forward_c8(buf) if (c8 and W % 8 == 0 and not getattr(self, '_no_c8_ctx', False)) else self.context_networks(buf)
            flow = ops.flow_update(flow_up, res, fine)                        # flow_up + (res + fine)
            flows.append([flow[:B], flow[B:]])
            self._tap('L%d.pair' % level, pair)
            self._tap('L%d.buf' % level, buf)
            self._tap('L%d.flow_up' % level, flow_up)
            self._tap('L%d.res' % level, res)
            self._tap('L%d.fine' % level, fine)
            self._tap('L%d.flow' % level, flow)
        flow_out = upsample2d_flow_as(flow, X, mode="bilinear", if_rate=True)
        self._tap('final.flow_bilinear', flow_out)
        if sgu:
            # (measured and not kept, round 3: this stem on a side stream = a parallel branch of the captured graph, forked before
            # or after the feature pyramid — 3.075 vs 3.076 ms; and with eager launches on two real streams 3.13 vs 3.14 ms: the step is
            # not idle-CU bound)
            guide = self._final_guidance(X if frames_dec is None else frames_dec, nb, tuple(flow.shape[2:]))
            if guide[0] == 'c8':
                flow_out = sgi.forward_in_buffer_c8(flow, guide[1], output_level_flow=flow_out, batch_shift=B)[1]
            else:
                flow_out = sgi.forward_in_buffer(flow, guide[1], guide[2], output_level_flow=flow_out, batch_shift=B)[1]
        return flow_out[:B], flow_out[B:], flows[::-1]

    def _final_guidance(self, X, nb, hw4):
        """The SGU's guidance features of the final up-sampling (model/upflow.py:525-531: sgi_model.output_conv on the frames)
        written into the input slot of the SGU stack's buffer -> ('c8', sbuf8) or ('nchw', sbuf, sslot)."""
        sgi = self.sgi_model
        em = sgi.dense_estimator_mask
        H4, W4 = hw4
        dt, dev = X.dtype, X.device
        last = sgi.upsample_output_conv[-1][0]
        hw = (X.shape[2], X.shape[3])
        for seq_ in sgi.upsample_output_conv:                                   # size of the stem's output (two stride-2 layers)
            hw = ops.conv3x3_out_hw(hw[0], hw[1], seq_[0].stride[0])
        pit = self._pitched()
        # (the stem's last layer reads NCHW rows: 16-byte aligned ones — a width that is a multiple of 8, or pitched intermediates)
        w_last = ops.conv3x3_out_hw(X.shape[2], X.shape[3], sgi.upsample_output_conv[1][0].stride[0])[1]
        if (c8_level_ok(nb, H4, W4, dt) and not getattr(self, '_no_c8', False) and not getattr(self, '_no_c8_sgu', False) and em.c8_ok()
                and (w_last % 8 == 0 or (pit and w_last >= 8))
                and last.stride[0] == 2 and last.in_channels > 16 and last.out_channels == em._ch_in // 2 and hw == (H4, W4)):
            sbuf8 = ops.c8_empty(nb, em._n_total, H4, W4, dt, dev)
            o0 = (em._n_total - em._ch_in) // 8
            sgi.output_conv(X, out8=sbuf8[:, o0:o0 + em._ch_in // 16], pitched=pit)   # the guidance stem's last layer writes octets
            return ('c8', sbuf8)
        sbuf, sslot = em.alloc_buffer(nb, H4, W4, dt, dev)
        G = sgi.output_conv(X, out=sslot[:, :32], pitched=pit)
        if tuple(G.shape[2:]) != (H4, W4):
            raise RuntimeError('sgu output_conv / flow size mismatch %s vs %s' % (tuple(G.shape), (H4, W4)))
        return ('nchw', sbuf, sslot)

    def _level_c8(self, level, Fm, pair, flow, nb, B, C, H, W, dt, dev, cache, use_sgu):
        """One level of _forward_stacked_fast with EVERY dense stack in the channel-octet layout: the estimator buffer is
        [conv5 | conv4 | conv3 | conv2 | conv1 | cost volume (11 octets, ops.corr81_c8_channel_map) | features (4) | flow (1) |
        refined flow (1)] octets; the cost volume, the 1x1 convolution and the flow bookkeeping write their octets directly."""
        est, sgi = self.flow_estimators, self.sgi_model
        nconv = sum(est._f) // 8
        ncorr = ops.CORR81_C8_OCTETS
        # input part of the estimator: [cost volume 81 | 1x1 features 32 | flow 2] (model/upflow.py:563-566) as 11 + 4 + 1 octets
        in_map = ops.corr81_c8_channel_map() + list(range(81, 113)) + [113, 114] + [-1] * 6
        n_est = nconv + len(in_map) // 8
        buf8 = ops.c8_empty(nb, (n_est + 1) * 8, H, W, dt, dev)
        o_feat, o_flow = nconv + ncorr, nconv + ncorr + 4
        pc = cache.get(('c8_1x1', level))
        if pc is None:
            pc = cache[('c8_1x1', level)] = _PackedConvC8(self.conv_1x1[level], (), range(C))
        if use_sgu:
            # (round 6: the projection is the input of BOTH stacks of the level: one launch stores it into both buffers)
            em = sgi.dense_estimator_mask
            sbuf8 = ops.c8_empty(nb, em._n_total, H, W, dt, dev)
            o0 = (em._n_total - em._ch_in) // 8
            pc.dual(Fm, buf8[:, o_feat:o_feat + 4], sbuf8[:, o0:o0 + 4])
        else:
            pc(None, Fm, buf8[:, o_feat:o_feat + 4])
        flow_up = upsample2d_flow_as(flow, Fm, mode="bilinear", if_rate=True)
        self._tap('L%d.flow_bilinear' % level, flow_up)
        if use_sgu:
            f16 = buf8[:, o_flow:o_flow + 1] if FLOW16_IN_BLEND[0] else None
            flow_up = sgi.forward_in_buffer_c8(flow_up, sbuf8, batch_shift=B, tap=lambda n_, t_: self._tap('L%d.%s' % (level, n_), t_), flow16=f16)[1]
            self._tap('L%d.sbuf8' % level, sbuf8)
            self._tap('L%d.flow_sgu' % level, flow_up)
        ops.warp_into(Fm, flow_up, pair[1], self.warping_layer.mask_mode, B)
        ops.corr81_norm_forward_c8(pair[0], pair[1], buf8[:, nconv:nconv + ncorr], leaky_slope=0.1)
        if not (use_sgu and FLOW16_IN_BLEND[0]):
            ops.flow_update_c8(flow_up, None, None, buf8[:, o_flow:o_flow + 1])
        res = est.forward_in_buffer_c8(buf8, in_map=in_map)
        ops.flow_update_c8(flow_up, res, None, buf8[:, o_flow + 1:o_flow + 2])        # flow_up + res -> context network input
        # the context network reads [x5 | refined flow] = every octet of the buffer
        nch = nconv * 8
        ctx_map = list(range(nch)) + [m + nch if m >= 0 else -1 for m in in_map] + [nch + 115, nch + 116] + [-1] * 6
        fine = self.context_networks.forward_c8(buf8, in_map=ctx_map)
        flow = ops.flow_update(flow_up, res, fine)                            # flow_up + (res + fine)
        self._tap('L%d.pair' % level, pair)
        self._tap('L%d.buf8' % level, buf8)
        self._tap('L%d.res' % level, res)
        self._tap('L%d.fine' % level, fine)
        return flow, [flow[:B], flow[B:]]

    def _level_update(self, Fn, Fwn, A, flow_up, add_to_flow=False):
        """res + fine for all 2B stacked items (model/upflow.py:557-572); add_to_flow: flow_up + (res + fine), the level's
        refined flow (:566-572)."""
        est = self.flow_estimators
        nb, _, H, W = Fn.shape
        nc = self.dim_corr
        fin = (lambda t: flow_up + t) if add_to_flow else (lambda t: t)
        if _fast_conv_ok(Fn):
            # one [2B, 565, H, W] buffer: corr81 (+LeakyReLU) -> 448..528, 1x1 features and flow -> 529..562, each
            # dense conv reads a suffix and writes its slice, refined flow appended at 563..564 for the context net
            buf, slot = est.alloc_buffer(nb, H, W, Fn.dtype, Fn.device, tail=2)
            ops.corr81_forward_raw(Fn.contiguous(), Fwn.contiguous(), out=slot[:, :nc], leaky_slope=0.1)
            slot[:, nc:nc + 32] = A
            slot[:, nc + 32:] = flow_up
            x5, res = est.forward_in_buffer(buf)
            res = res.float()
            buf[:, est._n_total:] = flow_up + res
            fine = self.context_networks(buf).float()
            return fin(res + fine)
        if torch.is_grad_enabled() and (Fn.requires_grad or Fwn.requires_grad):
            c = self._corr_leaky(Fn, Fwn)
            if est.train_in_buffer_ok([c, A, flow_up]):
                # training on the matrix cores: the estimator is one autograd node in the inference buffer layout; the
                # refined flow is appended to its buffer, which the context network reads whole (no concatenations)
                buf, res = est.forward_train([c, A, flow_up], flow_tail=flow_up)
                fine = self.context_networks(buf)
                if add_to_flow and flow_up.dtype == torch.float32 and fine.dtype == res.dtype and res.dtype != torch.float32:
                    return ops.flow_sum3(flow_up, res, fine)              # one launch each way
                return fin(res.float() + fine.float())
            x = torch.cat([c, A, flow_up.to(c.dtype)], dim=1)
        else:
            x = self._estimator_input(Fn, Fwn, A, flow_up)
        feat, res = est(x)
        res = res.float()
        fine = self.context_networks(torch.cat([feat, (flow_up + res).to(feat.dtype)], dim=1)).float()
        return fin(res + fine)

    def _corr_leaky(self, f_a, f_b_warp):
        """LeakyReLU(corr81(f_a, f_b_warp)) under autograd (model/upflow.py:561-563) as ONE node: the kernel applies the
        activation to its fp32 sums (fp32: the same bits as the two ops; 16-bit: one rounding instead of two), the backward
        masks the incoming gradient by the sign of the output."""
        c = self.correlation
        if (c.pad_size, c.kernel_size, c.max_displacement, c.stride1, c.stride2) == (4, 1, 4, 1, 1) and f_a.is_cuda:
            return ops.corr81(f_a, f_b_warp, float(self.leakyRELU.negative_slope))
        return self.leakyRELU(c(f_a, f_b_warp))

    def _estimator_input(self, f_a, f_b_warp, feat_1x1, flow):
        """cat[LeakyReLU(corr81(f_a, f_b_warp)), feat_1x1, flow]  (model/upflow.py:557-566).
        Inference: the kernel applies the LeakyReLU and writes the 81 channels straight into the
        115-channel buffer, so the cost volume is never re-read for an activation or a concat."""
        if torch.is_grad_enabled() and (f_a.requires_grad or f_b_warp.requires_grad):
            c = self._corr_leaky(f_a, f_b_warp)
            return torch.cat([c, feat_1x1, flow.to(c.dtype)], dim=1)
        B, _, H, W = f_a.shape
        buf = torch.empty((B, self.num_ch_in, H, W), dtype=f_a.dtype, device=f_a.device)
        ops.corr81_forward_raw(f_a.contiguous(), f_b_warp.contiguous(), out=buf[:, :self.dim_corr], leaky_slope=0.1)
        buf[:, self.dim_corr:self.dim_corr + 32] = feat_1x1
        buf[:, self.dim_corr + 32:] = flow
        return buf

    def decode_level_res(self, level, flow_1, flow_2, feature_1, feature_1_1x1, feature_2, feature_2_1x1, img_ori_1, img_ori_2):
        """One pyramid level, both directions (model/upflow.py:535-573)."""
        flow_1_up = upsample2d_flow_as(flow_1, feature_1, mode="bilinear", if_rate=True)
        flow_2_up = upsample2d_flow_as(flow_2, feature_2, mode="bilinear", if_rate=True)
        if level == 0:
            feature_2_warp, feature_1_warp = feature_2, feature_1
        else:
            if self.conf.if_sgu_upsample:
                flow_1_up = self.self_guided_upsample(flow_up_bilinear=flow_1_up, feature_1=feature_1_1x1, feature_2=feature_2_1x1)
                flow_2_up = self.self_guided_upsample(flow_up_bilinear=flow_2_up, feature_1=feature_2_1x1, feature_2=feature_1_1x1)
            feature_2_warp = self.warping_layer(feature_2, flow_1_up)
            feature_1_warp = self.warping_layer(feature_1, flow_2_up)
        if self.conf.if_norm_before_cost_volume:
            kw = dict(normalize=True, center=True, moments_across_channels=self.conf.norm_moments_across_channels,
                      moments_across_images=self.conf.norm_moments_across_images)
            feature_1, feature_2_warp = network_tools.normalize_features((feature_1, feature_2_warp), **kw)
            feature_2, feature_1_warp = network_tools.normalize_features((feature_2, feature_1_warp), **kw)
        in_1 = self._estimator_input(feature_1, feature_2_warp, feature_1_1x1, flow_1_up)
        in_2 = self._estimator_input(feature_2, feature_1_warp, feature_2_1x1, flow_2_up)
        feat_1, res_1 = self.flow_estimators(in_1)
        feat_2, res_2 = self.flow_estimators(in_2)
        res_1, res_2 = res_1.float(), res_2.float()
        fine_1 = self.context_networks(torch.cat([feat_1, (flow_1_up + res_1).to(feat_1.dtype)], dim=1)).float()
        fine_2 = self.context_networks(torch.cat([feat_2, (flow_2_up + res_2).to(feat_2.dtype)], dim=1)).float()
        return flow_1_up, flow_2_up, res_1 + fine_1, res_2 + fine_2

    def to_inference(self, dtype=torch.bfloat16, pyramid_dtype='auto', device=None):
        """Cast the network for 16-bit inference; with `pyramid_dtype` (torch.float16; 'auto', the default: fp16 under dtype = bfloat16
        — round 6: measured in bench.py's own line, 0.097 px to the reference instead of 0.179 px for -0.9 % throughput —, otherwise
        `dtype` itself; None or `dtype`: one type everywhere, like `net.to(dtype)`) the feature pyramid and the 1x1 projections —
        1.5 % of a step's flop, but > 60 % of the bf16 path's distance to the reference (profiles/r04_precision_localise.txt: their
        weights 0.156 px, their activations 0.091 px of 0.175 px at 384x1280) — keep fp16 weights, features and warped features, the
        cost volume reads fp16 features (its fp16 matrix instruction) and everything downstream (cost volume output, estimator,
        context network, SGU) stays `dtype`.  Call it on the fp32 network: the fp16 copies must be rounded from the fp32 weights, not
        from their bf16 roundings.  Returns self (in eval mode)."""
        if device is not None:
            self.to(device)
        if isinstance(pyramid_dtype, str):
            if pyramid_dtype != 'auto':
                raise ValueError("to_inference: pyramid_dtype must be a torch dtype, None or 'auto'")
            pyramid_dtype = torch.float16 if dtype == torch.bfloat16 else None
        if pyramid_dtype is None or pyramid_dtype == dtype:
            return self.to(dtype).eval()
        if dtype not in (torch.bfloat16, torch.float16) or pyramid_dtype not in (torch.bfloat16, torch.float16):
            raise ValueError('to_inference: dtype / pyramid_dtype must be torch.bfloat16 or torch.float16')
        src = self.feature_pyramid_extractor.convs[0][0][0].weight.dtype
        if src != torch.float32:
            import warnings
            warnings.warn('to_inference(pyramid_dtype=...) on a network that is already %s: the pyramid keeps that rounding' % src)
        pyr = {id(m) for part in (self.feature_pyramid_extractor, self.conv_1x1) for m in part.modules()}
        for m in self.modules():
            for name, p_ in list(m._parameters.items()):
                if p_ is not None:
                    p_.data = p_.data.to(pyramid_dtype if id(m) in pyr else dtype)
        self.invalidate_packed()
        return self.eval()

    def froze_PWC(self):
        for part in (self.feature_pyramid_extractor, self.flow_estimators, self.context_networks, self.conv_1x1):
            for p in part.parameters():
                p.requires_grad = False

    def self_guided_upsample(self, flow_up_bilinear, feature_1, feature_2, output_level_flow=None):
        return self.sgi_model(flow_up_bilinear, feature_1, feature_2, output_level_flow=output_level_flow)[1]

    @classmethod
    def demo(cls, device='cuda'):
        """Smoke run in the spirit of model/upflow.py:589-637, on the GPU."""
        conf = UPFlow_net.config()
        conf.update({'if_norm_before_cost_volume': True, 'norm_moments_across_channels': False,
                     'norm_moments_across_images': False, 'if_sgu_upsample': True})
        net = conf().to(device).eval()
        im = torch.rand(1, 3, 320, 320, device=device)
        with torch.no_grad():
            out = net({'im1': im, 'im2': im, 'if_loss': False})
        for k, v in out.items():
            tools.check_tensor(v, k)
