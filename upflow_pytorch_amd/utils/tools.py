"""Host-side mirror of the parts of the reference's `utils/tools.py` that sit on the hot path.

Same names and call signatures as `/root/reference/utils/tools.py` (class `tools` used as a
namespace), so code written against the reference (`tools.torch_warp(x, flo)`,
`tools.occ_check_model(...)(flow_f=..., flow_b=...)`, `tools.abstract_config`, `net.load_model(...)`)
keeps working; the arithmetic runs in the HIP kernels of libupflow_hip.so.
File formats (.flo, KITTI flow PNG) live in utils/flow_io.py, the KITTI readers and the evaluation bench in
dataset/kitti_dataset.py.  Out of scope: data prefetcher, visualisation, SP_transform.
"""
import torch
import torch.nn as nn

from .. import ops


class tools():
    # ------------------------------------------------------------------------------------------
    class abstract_config():
        """Attribute-bag configuration (utils/tools.py:32-105): defaults are set in __init__,
        `update(dict)` overrides only attributes that already exist, `get_name()` builds a tag."""
        name_filter_out_list = []

        def _public(self):
            return {k: getattr(self, k) for k in dir(self)
                    if not k.startswith('_') and not callable(getattr(self, k)) and k != 'name_filter_out_list'}

        def get_dict(self):
            return self._public()

        def update(self, data: dict, verbose=True):
            for k in self._public():
                if k in data:
                    setattr(self, k, data[k])
                    if verbose:
                        print('set param ====  %s:   %s' % (k, data[k]))

        def get_name(self, print_now=True):
            items = sorted((k, v) for k, v in self._public().items() if k not in self.name_filter_out_list)
            if print_now:
                print('=' * 10 + '\n{')
                for k, v in items:
                    print("\t%-50s: '%s,'," % ("'%s'" % k, v))
                print('}\n' + '=' * 10)
            return ''.join('%s|%s_' % kv for kv in items)

        def update_ex_name(self, ex_name: str):
            return ex_name

        @classmethod
        def check_length_of_file_name(cls, file_name):
            return len(file_name) < 255

        @classmethod
        def check_length_of_file_path(cls, filepath):
            return len(filepath) < 4096

    # ------------------------------------------------------------------------------------------
    class abstract_model(nn.Module):
        """state_dict-only checkpointing, utils/tools.py:107-155."""

        def save_model(self, save_path):
            torch.save(self.state_dict(), save_path)

        def load_model(self, load_path, if_relax=False, if_print=True):
            if if_print:
                print('loading protrained model from %s' % load_path)
            loaded = torch.load(load_path, map_location='cpu')
            if if_relax:
                # keep only keys that exist here with the same shape (utils/tools.py:115-125)
                own = self.state_dict()
                own.update({k: v for k, v in loaded.items() if k in own and v.shape == own[k].shape})
                loaded = own
            self.load_state_dict(loaded)

        def invalidate_packed(self):
            """Drop every packed-weight copy the 16-bit convolution path keeps (model/pwc_modules.py:_PackedConv3x3).
            Needed only after in-place parameter edits that bypass autograd's version counter (`p.data.copy_()`, EMA)."""
            from ..model.pwc_modules import packed_convs
            for pc in packed_convs(self):                # (incl. the channel-octet forms, `_packed8`)
                pc.invalidate()

        def load_state_dict(self, *args, **kwargs):
            r = super().load_state_dict(*args, **kwargs)
            self.invalidate_packed()
            return r

        @classmethod
        def choose_gpu(cls, model, gpu_opt=None):
            """The reference wraps in nn.DataParallel over all GPUs (utils/tools.py:130-148).  Here the
            multi-GPU form is one process per GPU: under torch.distributed the model is wrapped in
            DistributedDataParallel on this rank's device (RCCL all-reduce), otherwise it is moved to
            one GPU."""
            import torch.distributed as dist
            if gpu_opt is None:
                if dist.is_available() and dist.is_initialized():
                    from ..parallel import ddp_wrap
                    return ddp_wrap(model)
                return model.cuda()
            if type(gpu_opt) != int:
                raise ValueError('wrong gpu config, it show be int:  %s' % (str(gpu_opt)))
            torch.cuda.set_device(gpu_opt)
            return model.cuda(gpu_opt)

        @classmethod
        def save_model_gpu(cls, model, path):
            inner = getattr(model, 'module', model)      # unwrap DDP / DataParallel (utils/tools.py:150-155)
            inner.save_model(path)

    class AverageMeter():
        """utils/tools.py:282-297."""

        def __init__(self):
            self.reset()

        def reset(self):
            self.val = 0
            self.avg = 0
            self.sum = 0
            self.count = 0

        def update(self, val, num):
            self.val = val
            self.sum += val * num
            self.count += num
            self.avg = self.sum / self.count

    class time_clock():
        """Wall-clock stopwatch (utils/tools.py time_clock)."""

        def __init__(self):
            self.st = 0.0
            self.en = 0.0

        def start(self):
            import time
            self.st = time.time()

        def end(self):
            import time
            self.en = time.time()

        def get_during(self):
            return self.en - self.st

    # ---- file formats of the data / evaluation edge (utils/tools.py:1482-1632), see utils/flow_io.py ---------------
    @classmethod
    def write_flo(cls, flow, filename):
        from . import flow_io
        flow_io.write_flo(flow, filename)

    write_flow = write_flo

    @classmethod
    def read_flo(cls, filename):
        from . import flow_io
        return flow_io.read_flo(filename)

    read_flow = read_flo

    @classmethod
    def write_kitti_png_file(cls, flow_fn, flow_data, mask_data=None):
        from . import flow_io
        flow_io.write_kitti_png_file(flow_fn, flow_data, mask_data)

    @classmethod
    def write_flow_png(cls, filename, uv, v=None, mask=None):
        from . import flow_io
        flow_io.write_flow_png(filename, uv, v, mask)

    class abs_test_model():
        """Evaluation protocol of utils/tools.py:157-164."""
        save_dir = ''

        def eval_forward(self, im1, im2, gt, *args):
            return 0

        def eval_save_result(self, save_name, predflow, *args, **kwargs):
            pass

    # ------------------------------------------------------------------------------------------
    class boundary_dilated_warp():
        """Photometric-loss warp that samples the UN-cropped image, so flow leaving the crop still finds
        pixels (utils/tools.py:351-499).  Clamp-to-edge bilinear gather: indices are clamped to the
        image and the weights are computed from the CLAMPED corner coordinates (:409-412, :458-466).
        One HIP gather launch (csrc/loss.hip: upf_boundary_warp_*), differentiable wrt the flow."""

        @classmethod
        def warp_im(cls, I_nchw, flow_nchw, start_n211):
            return ops.boundary_warp(I_nchw, flow_nchw, start_n211)

    # ------------------------------------------------------------------------------------------
    @classmethod
    def torch_warp(cls, x, flo):
        """Backward warp without validity mask (utils/tools.py:1274-1319) — one HIP launch."""
        return ops.warp(x, flo, None)

    @classmethod
    def torch_warp_mask(cls, x, flo):
        """utils/tools.py:1229-1272: warp, and a mask with `mask < 0.9999 -> 0, else 1`.
        The reference thresholds grid_sample(ones) at 0.9999 here (not >= 1.0); that sum is a smooth
        function of the position, so it is evaluated from the warped ones tensor."""
        out = ops.warp(x, flo, None)
        m = ops.warp(torch.ones_like(x), flo, None)
        m = (m >= 0.9999).to(out.dtype)
        return out * m, m

    @classmethod
    def check_tensor(cls, data, name, print_data=False, print_in_txt=None):
        if data.is_cuda:
            data = data.detach().cpu()
        a = data.float().numpy()
        print(name, a.shape, 'max', a.max(), 'min', a.min(), 'mean', a.mean())

    # ------------------------------------------------------------------------------------------
    class occ_check_model():
        """Forward-backward consistency occlusion masks (utils/tools.py:501-677).
        `obj_out_all`: 'all' = consistency check only, 'out' = outgoing-flow mask only,
        'obj' = consistent OR leaving the image (the configuration model/upflow.py:364-365 uses)."""

        def __init__(self, occ_type='for_back_check', occ_alpha_1=1.0, occ_alpha_2=0.05, sum_abs_or_squar=True, obj_out_all='all'):
            self.occ_type_ls = ['for_back_check', 'forward_warp']
            assert occ_type in self.occ_type_ls
            assert obj_out_all in ['obj', 'out', 'all']
            self.occ_type = occ_type
            self.occ_alpha_1 = occ_alpha_1
            self.occ_alpha_2 = occ_alpha_2
            self.sum_abs_or_squar = True
            self.obj_out_all = obj_out_all

        def __call__(self, flow_f, flow_b, scale=1):
            if self.occ_type != 'for_back_check':
                raise ValueError('not implemented')
            if self.obj_out_all == 'obj':
                # fused kernel: 2 warps + magnitudes + thresholds + outgoing mask in one launch
                return ops.occ_check(flow_f, flow_b, self.occ_alpha_1, self.occ_alpha_2 / scale)
            if self.obj_out_all == 'all':
                return self._forward_backward_occ_check(flow_f, flow_b, scale)
            return self.torch_outgoing_occ_check(flow_f), self.torch_outgoing_occ_check(flow_b)

        def _forward_backward_occ_check(self, flow_fw, flow_bw, scale=1):
            def mag(v):                                    # utils/tools.py:559: sum_c sqrt(v_c^2)
                return v.abs().sum(dim=1, keepdim=True)
            m = mag(flow_fw) + mag(flow_bw)
            bw_w = tools.torch_warp(flow_bw, flow_fw)
            fw_w = tools.torch_warp(flow_fw, flow_bw)
            thr = self.occ_alpha_1 * m + self.occ_alpha_2 / scale
            return (mag(flow_fw + bw_w) < thr).float(), (mag(flow_bw + fw_w) < thr).float()

        @classmethod
        def torch_outgoing_occ_check(cls, flow):
            B, C, H, W = flow.shape
            xx = torch.arange(W, device=flow.device, dtype=torch.float32).view(1, 1, 1, W)
            yy = torch.arange(H, device=flow.device, dtype=torch.float32).view(1, 1, H, 1)
            px = xx + flow[:, 0:1].float()
            py = yy + flow[:, 1:2].float()
            return ((px <= W - 1) & (px >= 0) & (py <= H - 1) & (py >= 0)).float()

        @classmethod
        def torch_get_obj_occ_check(cls, occ_mask, out_occ):
            return ((occ_mask == 1) | (out_occ == 0)).to(occ_mask.dtype)
