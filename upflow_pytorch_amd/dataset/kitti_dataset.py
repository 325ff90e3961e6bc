"""KITTI data / evaluation edge of the hot path — mirror of `/root/reference/dataset/kitti_dataset.py` without
TensorFlow / cv2 / pypng.

Same names and call shapes: `img_func`, `kitti_train.kitti_data_with_start_point` (training crops + un-cropped frames +
crop offset `start`, :268-342), `kitti_flow.get_file_names`, `kitti_flow.kitti_train` (:574-630),
`kitti_flow.Evaluation_bench` (:379-499: average end-point error, KITTI outlier rate F1, occluded / non-occluded split).
The reference hard-codes its data roots (:31,38); here they are module attributes / constructor arguments, and nothing
is read at import time.  Metrics run on whatever device the tensors are on."""
import os
import random

import numpy as np
import torch

from ..utils import flow_io
from ..utils.tools import tools

mv_data_dir = os.environ.get('UPF_KITTI_MV_DIR', '/data/Optical_Flow_all/datasets/KITTI_data/KITTI_data_mv')
kitti_flow_dir = os.environ.get('UPF_KITTI_FLOW_DIR', '/data/Optical_Flow_all/datasets/KITTI_data')


class img_func():
    MEAN = [104.920005, 110.1753, 114.785955]
    STDDEV = 1 / 0.0039216

    @classmethod
    def get_process_img_only_img(cls, img, normalize=True, if_horizontal_flip=False):
        """[H,W,3] uint8 -> [3,H,W] float: (image - mean) / stddev (dataset/kitti_dataset.py:62-80)."""
        if if_horizontal_flip:
            img = np.flip(img, 1)
        if normalize:
            img = (img - cls.MEAN) / cls.STDDEV
        return np.transpose(img, [2, 0, 1])

    @classmethod
    def get_process_img(cls, img_name, normalize=True, if_horizontal_flip=False):
        return cls.get_process_img_only_img(flow_io.read_image(img_name), normalize, if_horizontal_flip)

    @classmethod
    def frame_name_to_num(cls, name):
        stripped = name.split('.')[0].lstrip('0')
        return 0 if stripped == '' else int(stripped)

    @classmethod
    def np_2_tensor(cls, *args):
        return [torch.from_numpy(np.ascontiguousarray(a)).float() for a in args]

    @classmethod
    def read_png_flow(cls, fpath):
        return flow_io.read_kitti_png_flow(fpath)

    read_flow = read_png_flow


def _pairs(image_dir, flow_dir_occ=None, flow_dir_noc=None):
    image_files = sorted(os.listdir(image_dir))
    assert len(image_files) % 2 == 0, 'expected pairs of images'
    out = []
    if flow_dir_occ is None:
        for i in range(len(image_files) // 2):
            out.append({'im1': os.path.join(image_dir, image_files[2 * i]), 'im2': os.path.join(image_dir, image_files[2 * i + 1])})
        return out
    occ, noc = sorted(os.listdir(flow_dir_occ)), sorted(os.listdir(flow_dir_noc))
    assert len(occ) == len(noc) == len(image_files) // 2, 'flow / image counts disagree'
    for i in range(len(occ)):
        out.append({'flow_occ': os.path.join(flow_dir_occ, occ[i]), 'flow_noc': os.path.join(flow_dir_noc, noc[i]),
                    'im1': os.path.join(image_dir, image_files[2 * i]), 'im2': os.path.join(image_dir, image_files[2 * i + 1])})
    return out


class kitti_train:
    @classmethod
    def mv_data_get_file_names(cls, root=None):
        """Consecutive frame pairs of the multi-view extensions, test frames 9..12 excluded (:193-227)."""
        root = root or mv_data_dir
        names = {}
        for key, sub in (('2012', os.path.join('stereo_flow_2012', 'data_stereo_flow_multiview')),
                         ('2015', os.path.join('stereo_flow_2015', 'data_scene_flow_multiview'))):
            samples = []
            for split in ('testing', 'training'):
                img_dir = os.path.join(root, sub, split, 'image_2')
                if not os.path.isdir(img_dir):
                    continue
                files = sorted(os.listdir(img_dir))
                for a, b in zip(files[:-1], files[1:]):
                    ia, ib = int(a[-6:-4]), int(b[-6:-4])
                    if ia != ib - 1 or 12 >= ia >= 9 or 12 >= ib >= 9:
                        continue
                    samples.append((os.path.join(img_dir, a), os.path.join(img_dir, b)))
            names[key] = samples
        return names

    class kitti_data_with_start_point(torch.utils.data.Dataset):
        class config(tools.abstract_config):
            def __init__(self, **kwargs):
                self.crop_size = (256, 832)
                self.rho = 8
                self.swap_images = True
                self.normalize = True
                self.repeat = None
                self.horizontal_flip_aug = True
                self.mv_type = None
                self.data_root = None
                self.update(kwargs, verbose=False)

            def __call__(self):
                return kitti_train.kitti_data_with_start_point(self)

        def __init__(self, conf):
            self.conf = conf
            if conf.mv_type not in ('2015', '2012'):
                raise ValueError('mv_type should be 2012 or 2015')
            self.filenames_extended = kitti_train.mv_data_get_file_names(conf.data_root)[conf.mv_type]
            self.N = len(self.filenames_extended)

        def __len__(self):
            if self.conf.repeat is None or self.conf.repeat <= 0:
                return self.N
            return self.N * int(self.conf.repeat)

        def __getitem__(self, index):
            im1, im2 = self.read_img(index)
            im1_crop, im2_crop, start = self.random_crop(im1, im2)
            return tuple(img_func.np_2_tensor(im1, im2, im1_crop, im2_crop, start))

        def random_crop(self, im1, im2):
            height, width = im1.shape[1:]
            ph, pw = self.conf.crop_size
            x = np.random.randint(self.conf.rho, width - self.conf.rho - pw)
            y = np.random.randint(self.conf.rho, height - self.conf.rho - ph)
            start = np.expand_dims(np.expand_dims(np.array([x, y]), 1), 2)
            return im1[:, y:y + ph, x:x + pw], im2[:, y:y + ph, x:x + pw], start

        def read_img(self, index):
            flip = bool(self.conf.horizontal_flip_aug and random.random() < 0.5)
            p1, p2 = self.filenames_extended[index % self.N]
            im1 = img_func.get_process_img(p1, normalize=self.conf.normalize, if_horizontal_flip=flip)
            im2 = img_func.get_process_img(p2, normalize=self.conf.normalize, if_horizontal_flip=flip)
            if self.conf.swap_images and random.random() < 0.5:
                return im2, im1
            return im1, im2


class kitti_flow:
    @classmethod
    def get_file_names(cls, root=None):
        """{'2012_train','2015_train','2012_test','2015_test'} -> list of path dicts (:509-572)."""
        root = root or kitti_flow_dir
        data = {}
        for key, sub, img in (('2012', 'data_stereo_flow', 'colored_0'), ('2015', 'data_scene_flow', 'image_2')):
            tr = os.path.join(root, sub, 'training')
            if os.path.isdir(os.path.join(tr, img)):
                data[key + '_train'] = _pairs(os.path.join(tr, img), os.path.join(tr, 'flow_occ'), os.path.join(tr, 'flow_noc'))
            te = os.path.join(root, sub, 'testing', img)
            if os.path.isdir(te):
                data[key + '_test'] = _pairs(te)
        return data

    class kitti_train():
        def __init__(self, name, root=None):
            assert name in ['2012_train', '2015_train', '2012_test', '2015_test']
            self.file_names = kitti_flow.get_file_names(root)[name]
            self.normalize = True
            self.name = name

        def __len__(self):
            return len(self.file_names)

        def __getitem__(self, index):
            d = self.file_names[index]
            im1 = img_func.get_process_img(d['im1'], normalize=self.normalize)
            im2 = img_func.get_process_img(d['im2'], normalize=self.normalize)
            if self.name.endswith('_test'):
                im1, im2 = img_func.np_2_tensor(im1, im2)
                return im1, im2, os.path.basename(d['im1']).replace('.png', '')
            occ, occmask = img_func.read_png_flow(d['flow_occ'])
            noc, nocmask = img_func.read_png_flow(d['flow_noc'])
            return tuple(img_func.np_2_tensor(im1, im2, occ, occmask, noc, nocmask))

    class Evaluation_bench():
        """`bench(test_model)` -> (all EPE, F1 %, non-occluded EPE, occluded-only EPE), dataset/kitti_dataset.py:379-451."""

        def __init__(self, name, if_gpu=True, batch_size=1, root=None, dataset=None):
            assert name in ['2012_train', '2015_train', '2012_test', '2015_test']
            self.name = name
            self.batch_size = batch_size
            self.if_gpu = if_gpu
            self.dataset = dataset if dataset is not None else kitti_flow.kitti_train(name=name, root=root)

        def _dev(self, *ts):
            return [t.cuda(non_blocking=True) if self.if_gpu else t for t in ts]

        def __call__(self, test_model):
            if self.name.endswith('_test'):
                for i in range(len(self.dataset)):
                    im1, im2, img_name = self.dataset[i]
                    im1, im2 = self._dev(im1.unsqueeze(0), im2.unsqueeze(0))
                    test_model.eval_save_result(img_name, test_model.eval_forward(im1, im2, 0))
                return None
            meters = [tools.AverageMeter() for _ in range(4)]
            index = [-1]

            def batches():
                for s in range(0, len(self.dataset), self.batch_size):
                    items = [self.dataset[i] for i in range(s, min(len(self.dataset), s + self.batch_size))]
                    if len({tuple(it[0].shape) for it in items}) != 1:            # KITTI frames differ in size: one by one then
                        for it in items:
                            yield self._dev(*[t.unsqueeze(0) for t in it])
                    else:
                        yield self._dev(*[torch.stack(col) for col in zip(*items)])

            def score(batch, predflow):
                index[0] += 1
                im1, im2, occ, occmask, noc, nocmask = batch
                num = im1.shape[0]
                vals = [self.flow_error_avg(occ, predflow, occmask), self.outlier_pct(occ, predflow, occmask),
                        self.flow_error_avg(noc, predflow, nocmask), self.flow_error_avg(occ, predflow, occmask - nocmask)]
                for m, v in zip(meters, vals):
                    m.update(val=float(v), num=num)
                save_name = 'all_%.2f f1_%.1f noc_%.2f occ_%.2f__%d' % (meters[0].val, meters[1].val, meters[2].val, meters[3].val, index[0])
                test_model.eval_save_result(save_name, predflow, occmask=occmask)

            stream_fn = getattr(test_model, 'eval_forward_stream', None)
            if stream_fn is not None and getattr(test_model, 'pipe', None) is not None:
                # several pairs in flight (not in the reference, whose loop is strictly one pair at a time): the forward of the next
                # pairs runs while this one is scored; same order, same numbers
                from collections import deque
                waiting = deque()

                def pairs():
                    for b in batches():
                        waiting.append(b)
                        yield b[0], b[1]
                for predflow in stream_fn(pairs()):
                    score(waiting.popleft(), predflow)
            else:
                for batch in batches():
                    score(batch, test_model.eval_forward(*batch))
            return meters[0].avg, meters[1].avg, meters[2].avg, meters[3].avg

        @classmethod
        def flow_error_avg(cls, flow_1, flow_2, mask):
            """Average end-point error over the masked pixels, torch n c h w (:464-475)."""
            diff = torch.sqrt(torch.sum((flow_1 - flow_2) ** 2, dim=(1,), keepdim=True)) * mask
            return torch.sum(diff) / (torch.sum(mask) + 1e-6)

        @classmethod
        def outlier_pct(cls, gt_flow, predflow, mask, threshold=3.0, relative=0.05):
            """KITTI outlier rate in percent: end-point error > max(3 px, 5 % of |gt|) among the masked pixels (:477-499)."""
            def euclidean(t):
                return torch.sqrt(torch.sum(t ** 2, dim=(1,), keepdim=True))
            diff = euclidean(gt_flow - predflow) * mask
            thr = torch.tensor(threshold).type_as(gt_flow)
            if relative is not None:
                thr = torch.max(thr, euclidean(gt_flow) * relative)
            return torch.sum(diff > thr) / torch.sum(mask) * 100
