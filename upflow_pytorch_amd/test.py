"""Evaluation entry point — mirror of `/root/reference/test.py`: `Test_model(pretrain_path)` with the reference's flags
(test.py:22-30), `eval_forward(im1, im2, gt, ...) -> flow_fw` (:40-47), `kitti_2015_test()` (:54-61).

What differs: the forward runs on the MI355X-native operators, and — because KITTI is evaluated at batch 1 on frames whose
size changes between sequences — through `runtime.ShapeCachedInference`: one captured hipGraph per frame size, replayed for
every further pair of that size (`graph=False` restores plain eager calls).  `dtype`: torch.float32 (the parity mode, the
reference's arithmetic within 1e-4 px), torch.float16 / torch.bfloat16 (the fast modes; README.md "which dtype")."""
import torch

from .utils.tools import tools
from .model.upflow import UPFlow_net

PARAM_DICT = {
    # use cost volume norm                                   (test.py:22-30)
    'if_norm_before_cost_volume': True,
    'norm_moments_across_channels': False,
    'norm_moments_across_images': False,
    'if_froze_pwc': False,
    'if_use_cor_pytorch': False,
    'if_sgu_upsample': True,
}


class Test_model(tools.abs_test_model):
    def __init__(self, pretrain_path='./scripts/upflow_kitti2015.pth', dtype=torch.float32, graph=True, device='cuda', net=None, streams=1):
        """streams > 1: `Evaluation_bench` keeps that many frame pairs in flight (runtime.PipelinedEvaluation, through
        eval_forward_stream) instead of one at a time — same results, ~1.7x the pairs per second at KITTI's frame size."""
        super(Test_model, self).__init__()
        supplied = net is not None
        if net is None:
            net_conf = UPFlow_net.config()
            net_conf.update(PARAM_DICT, verbose=False)
            net = net_conf()
            if pretrain_path is not None:
                net.load_model(pretrain_path, if_relax=True, if_print=True)
        # a supplied network that is already 16-bit — possibly MIXED (UPFlow_net.to_inference(dtype, pyramid_dtype=...)) — keeps its
        # types: `.to(dtype)` would flatten the fp16 pyramid into the decoder's type and evaluate another configuration
        if supplied and (dtype is None or any(p_.dtype != torch.float32 for p_ in net.parameters())):
            net = net.to(device).eval()
        else:
            net = net.to(device).to(dtype).eval()
        self.net_work = net
        self.runner = None
        self.pipe = None
        if graph:
            from .runtime import ShapeCachedInference, PipelinedEvaluation
            self.runner = ShapeCachedInference(net)
            if streams > 1:
                self.pipe = PipelinedEvaluation(net, streams=streams)

    def eval_forward_stream(self, pairs):
        """pairs: iterable of (im1, im2) -> generator of flow_fw, in order, with several pairs in flight; every yielded tensor is the
        caller's own copy, as eval_forward's is.  (Not in the reference: its loop is one pair at a time.)"""
        if self.pipe is None:
            for im1, im2 in pairs:
                yield self.eval_forward(im1, im2, 0)
            return
        with torch.no_grad():
            for out in self.pipe.map(pairs):
                # (a copy, like eval_forward's: the slot's output is overwritten when the generator advances, and eval_save_result
                # — a hook whose documented purpose is to store / save predflows — may keep it; 3.7 MB per KITTI pair.  ADVICE r5)
                yield out['flow_f_out'].clone()

    def eval_forward(self, im1, im2, gt, *args):
        # === network output                                 (test.py:40-47)
        with torch.no_grad():
            if self.runner is not None:
                output_dict = self.runner(im1, im2)
                return output_dict['flow_f_out'].clone()      # (the runner owns its outputs: the caller keeps predflow across calls)
            output_dict = self.net_work({'im1': im1, 'im2': im2, 'if_loss': False})
            return output_dict['flow_f_out']

    def eval_save_result(self, save_name, predflow, *args, **kwargs):
        print(save_name)


def kitti_2015_test(pretrain_path='./scripts/upflow_kitti2015.pth', dtype=torch.float32, root=None):
    from .dataset.kitti_dataset import kitti_flow
    # note that eval batch size should be 1 for KITTI 2012 and KITTI 2015 (image size may be different for different sequence)
    bench = kitti_flow.Evaluation_bench(name='2015_train', if_gpu=True, batch_size=1, root=root)
    testmodel = Test_model(pretrain_path=pretrain_path, dtype=dtype)
    epe_all, f1, epe_noc, epe_occ = bench(testmodel)
    print('EPE All = %.2f, F1 = %.2f, EPE Noc = %.2f, EPE Occ = %.2f' % (epe_all, f1, epe_noc, epe_occ))
    return epe_all, f1, epe_noc, epe_occ


if __name__ == '__main__':
    kitti_2015_test()
