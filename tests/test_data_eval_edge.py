"""Data / evaluation edge: .flo and KITTI flow-PNG round trips and known answers, the PNG codec
against an independent encoder (Pillow, all five scan-line filters), and the EPE / F1 evaluation bench against
hand-computed values and against the oracle's EPE (dataset/kitti_dataset.py:464-499).  CPU only."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch

import oracle
from upflow_pytorch_amd.utils import flow_io
from upflow_pytorch_amd.utils.tools import tools
from upflow_pytorch_amd.dataset.kitti_dataset import kitti_flow, kitti_train, img_func


def test_flo_round_trip_and_layout(tmp_path):
    g = np.random.default_rng(0)
    flow = g.normal(size=(7, 13, 2)).astype(np.float32) * 20
    p = str(tmp_path / 'a.flo')
    tools.write_flo(flow, p)
    raw = open(p, 'rb').read()
    assert len(raw) == 12 + 7 * 13 * 2 * 4
    magic, w, h = struct.unpack('<fii', raw[:12])                       # Middlebury header: magic, WIDTH, height
    assert magic == np.float32(202021.25) and (w, h) == (13, 7)
    assert struct.unpack('<ff', raw[12:20]) == (flow[0, 0, 0], flow[0, 0, 1])     # (u, v) interleaved, row-major
    assert np.array_equal(tools.read_flo(p), flow) and np.array_equal(flow_io.read_flow(p), flow)
    open(p, 'wb').write(raw[:40])
    with pytest.raises(ValueError):
        flow_io.read_flo(p)                                             # truncated payload
    open(p, 'wb').write(struct.pack('<fii', 1.0, 13, 7) + raw[12:])
    with pytest.raises(ValueError):
        flow_io.read_flo(p)                                             # wrong magic


def test_kitti_flow_png_known_answer_and_round_trip(tmp_path):
    flow = np.zeros((3, 4, 2))
    flow[0, 0] = (0.0, 0.0)
    flow[0, 1] = (1.0, -1.0)
    flow[1, 2] = (-512.0, 511.984375)          # the ends of the 16-bit range
    flow[2, 3] = (10.5, 0.015625)              # 1/64 px resolution
    mask = np.zeros((3, 4), dtype=np.uint8)
    mask[0, 1] = mask[1, 2] = mask[2, 3] = 1
    p = str(tmp_path / 'f.png')
    tools.write_kitti_png_file(p, flow, mask)
    raw = flow_io.read_png(p)
    assert raw.dtype == np.uint16 and raw.shape == (3, 4, 3)
    assert tuple(raw[0, 0]) == (32768, 32768, 0)                        # u*64 + 2^15, v*64 + 2^15, valid
    assert tuple(raw[0, 1]) == (32768 + 64, 32768 - 64, 1)
    assert tuple(raw[1, 2]) == (0, 65535, 1)
    assert tuple(raw[2, 3]) == (32768 + 672, 32769, 1)
    f, m = img_func.read_png_flow(p)
    assert f.shape == (2, 3, 4) and m.shape == (1, 3, 4) and m.dtype == np.uint8
    assert np.array_equal(np.transpose(f, (1, 2, 0)), flow) and np.array_equal(m[0], mask)
    # random field: quantised to 1/64 px on write
    g = np.random.default_rng(1)
    fl = g.normal(size=(37, 53, 2)) * 30
    flow_io.write_flow_png(p, fl[:, :, 0], fl[:, :, 1])
    f2, m2 = flow_io.read_kitti_png_flow(p)
    assert np.abs(np.transpose(f2, (1, 2, 0)) - fl).max() <= 1 / 64 and m2.min() == 1


@pytest.mark.parametrize('mode,dtype', [('RGB', np.uint8), ('L', np.uint8), ('RGBA', np.uint8), ('I;16', np.uint16)])
def test_png_reader_against_pillow_encoder(tmp_path, mode, dtype):
    """Files written by an independent encoder (Pillow, adaptive filtering => Sub / Up / Average / Paeth rows) decode to
    the same pixels; files written here decode in Pillow to the same pixels."""
    Image = pytest.importorskip('PIL.Image')
    g = np.random.default_rng(2)
    base = np.cumsum(g.integers(0, 5, size=(41, 67, 4)), axis=1)        # smooth-ish rows so that the encoder picks filters
    if mode == 'I;16':
        arr = (base[:, :, 0] * 97 % 65536).astype(np.uint16)
    else:
        c = {'RGB': 3, 'L': 1, 'RGBA': 4}[mode]
        arr = (base[:, :, :c] % 256).astype(np.uint8)
        arr = arr[:, :, 0] if c == 1 else arr
    p = str(tmp_path / 'p.png')
    Image.fromarray(arr, mode=mode if mode != 'I;16' else None).save(p, optimize=True)
    got = flow_io.read_png(p)
    assert got.dtype == dtype and np.array_equal(got.reshape(arr.shape), arr)
    rows = zlib.decompress(b''.join(_idat(p)))
    q = str(tmp_path / 'q.png')
    flow_io.write_png(q, arr)
    assert np.array_equal(np.asarray(Image.open(q)).reshape(arr.shape), arr)
    assert rows is not None


def _idat(path):
    data = open(path, 'rb').read()
    pos, out = 8, []
    while pos < len(data):
        n, tag = struct.unpack('>I4s', data[pos:pos + 8])
        if tag == b'IDAT':
            out.append(data[pos + 8:pos + 8 + n])
        pos += 12 + n
    return out


def test_png_all_filters_by_hand(tmp_path):
    """A file whose rows use filters 0..4 in turn (built here byte by byte, independently of write_png)."""
    g = np.random.default_rng(3)
    img = g.integers(0, 256, size=(5, 6, 3), dtype=np.uint8)
    bpp, stride = 3, 18
    raw = bytearray()
    prev = np.zeros(stride, dtype=np.int64)
    for y in range(5):
        cur = img[y].reshape(-1).astype(np.int64)
        left = np.concatenate([np.zeros(bpp, dtype=np.int64), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, dtype=np.int64), prev[:-bpp]])
        if y == 0:
            f = cur
        elif y == 1:
            f = cur - left
        elif y == 2:
            f = cur - prev
        elif y == 3:
            f = cur - (left + prev) // 2
        else:
            p = left + prev - ul
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            f = cur - pred
        raw += bytes([y]) + bytes((f % 256).astype(np.uint8))
        prev = cur
    p = str(tmp_path / 'h.png')
    with open(p, 'wb') as fh:
        fh.write(bytes([0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a]))
        for tag, body in ((b'IHDR', struct.pack('>IIBBBBB', 6, 5, 8, 2, 0, 0, 0)), (b'IDAT', zlib.compress(bytes(raw))), (b'IEND', b'')):
            fh.write(struct.pack('>I', len(body)) + tag + body + struct.pack('>I', zlib.crc32(tag + body) & 0xffffffff))
    assert np.array_equal(flow_io.read_png(p), img)
    bad = bytearray(open(p, 'rb').read())
    bad[40] ^= 0xff
    open(p, 'wb').write(bytes(bad))
    with pytest.raises(ValueError):
        flow_io.read_png(p)                                            # CRC mismatch is detected


def test_epe_and_f1_known_answers():
    E = kitti_flow.Evaluation_bench
    gt = torch.zeros(1, 2, 2, 3)
    gt[0, 0] = torch.tensor([[0.0, 10.0, 100.0], [0.0, 0.0, 0.0]])
    pred = gt.clone()
    pred[0, 0, 0, 0] += 3.0          # error 3.0, |gt| 0    -> NOT an outlier (strictly greater than 3)
    pred[0, 1, 0, 1] += 3.5          # error 3.5, |gt| 10   -> outlier (thr = max(3, 0.5) = 3)
    pred[0, 0, 0, 2] += 4.0          # error 4.0, |gt| 100  -> not an outlier (thr = 5)
    pred[0, 0, 1, 0] += 50.0         # masked out below
    mask = torch.ones(1, 1, 2, 3)
    mask[0, 0, 1, 0] = 0
    epe = float(E.flow_error_avg(gt, pred, mask))
    assert abs(epe - (3.0 + 3.5 + 4.0) / 5) <= 1e-6
    assert abs(float(E.outlier_pct(gt, pred, mask)) - 100.0 * 1 / 5) <= 1e-5
    # with a full mask the bench's EPE is the oracle's EPE (dataset/kitti_dataset.py:464-475 with mask == 1)
    g = torch.Generator().manual_seed(4)
    a, b = torch.randn(2, 2, 9, 11, generator=g), torch.randn(2, 2, 9, 11, generator=g)
    assert abs(float(E.flow_error_avg(a, b, torch.ones(2, 1, 9, 11))) - oracle.epe(a, b)) <= 1e-6


def test_evaluation_bench_end_to_end(tmp_path):
    """A synthetic KITTI-2015 tree written with this package's writers, read back through kitti_flow.kitti_train and
    scored by Evaluation_bench with a model that returns the ground truth shifted by a known error."""
    Image = pytest.importorskip('PIL.Image')
    root = tmp_path / 'data_scene_flow' / 'training'
    for d in ('image_2', 'flow_occ', 'flow_noc'):
        os.makedirs(root / d)
    g = np.random.default_rng(5)
    for i in range(3):
        for k in (10, 11):
            flow_io.write_png(str(root / 'image_2' / ('%06d_%d.png' % (i, k))), g.integers(0, 256, size=(32, 64, 3), dtype=np.uint8))
        fl = np.round(g.normal(size=(32, 64, 2)) * 10 * 64) / 64
        occ_mask = (g.random((32, 64)) > 0.2).astype(np.uint8)
        noc_mask = occ_mask * (g.random((32, 64)) > 0.3).astype(np.uint8)
        tools.write_kitti_png_file(str(root / 'flow_occ' / ('%06d_10.png' % i)), fl, occ_mask)
        tools.write_kitti_png_file(str(root / 'flow_noc' / ('%06d_10.png' % i)), fl, noc_mask)
    ds = kitti_flow.kitti_train('2015_train', root=str(tmp_path))
    assert len(ds) == 3
    im1, im2, occ, occmask, noc, nocmask = ds[0]
    assert im1.shape == (3, 32, 64) and occ.shape == (2, 32, 64) and occmask.shape == (1, 32, 64)
    assert abs(float(im1.max())) < 1.0                                  # normalised like dataset/kitti_dataset.py:45-54

    class Model(tools.abs_test_model):
        saved = []

        def eval_forward(self, im1, im2, gt, *args):
            return gt + torch.tensor([4.0, 0.0]).view(1, 2, 1, 1)       # constant 4 px error

        def eval_save_result(self, save_name, predflow, *args, **kwargs):
            self.saved.append(save_name)
    bench = kitti_flow.Evaluation_bench('2015_train', if_gpu=False, batch_size=2, root=str(tmp_path))
    all_epe, f1, noc_epe, occ_epe = bench(Model())
    assert abs(all_epe - 4.0) <= 1e-4 and abs(noc_epe - 4.0) <= 1e-4 and abs(occ_epe - 4.0) <= 1e-4
    assert 0.0 < f1 <= 100.0 and len(Model.saved) == 2                  # batches of 2 + 1


def test_training_dataset_crops(tmp_path):
    root = tmp_path / 'stereo_flow_2015' / 'data_scene_flow_multiview' / 'training' / 'image_2'
    os.makedirs(root)
    g = np.random.default_rng(6)
    for k in range(0, 21):
        flow_io.write_png(str(root / ('000000_%02d.png' % k)), g.integers(0, 256, size=(96, 160, 3), dtype=np.uint8))
    names = kitti_train.mv_data_get_file_names(str(tmp_path))['2015']
    assert len(names) == 20 - 5                                          # pairs touching frames 9..12 are excluded
    conf = kitti_train.kitti_data_with_start_point.config(mv_type='2015', crop_size=(64, 128), data_root=str(tmp_path))
    ds = conf()
    im1, im2, c1, c2, start = ds[3]
    assert im1.shape == (3, 96, 160) and c1.shape == (3, 64, 128) and start.shape == (2, 1, 1)
    x, y = int(start[0]), int(start[1])
    assert torch.equal(c1, im1[:, y:y + 64, x:x + 128]) and x >= 8 and y >= 8


# ---- pinned on the reference (tests/golden/eval_edge.npz, generated by make_golden.py `eval` from the imported
# /root/reference/utils/tools.py:1482-1632 and dataset/kitti_dataset.py:67-147, :464-499) ---------------------------------
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'eval_edge.npz')


@pytest.fixture(scope='module')
def edge():
    return np.load(GOLD)


def test_flo_bytes_equal_the_references(edge, tmp_path):
    """The build's writer produces the reference's bytes, and its reader returns the reference's array from them."""
    for i in range(3):
        flow, ref_bytes = edge['flo_flow_%d' % i], edge['flo_bytes_%d' % i].tobytes()
        p = str(tmp_path / ('w%d.flo' % i))
        for wr in (tools.write_flo, tools.write_flow, flow_io.write_flo):
            wr(flow, p)
            assert open(p, 'rb').read() == ref_bytes
        open(p, 'wb').write(ref_bytes)
        for rd in (tools.read_flo, tools.read_flow, flow_io.read_flo):
            got = rd(p)
            assert got.dtype == np.float32 and np.array_equal(got, flow)


def test_kitti_png_quantisation_equals_the_references(edge, tmp_path):
    """What the reference hands to pypng (write_flow_png, RGB) / cv2 (write_kitti_png_file, BGR) == the samples in our file."""
    p = str(tmp_path / 'q.png')
    uv, mask = edge['png_uv'], edge['png_mask']
    tools.write_flow_png(p, uv, mask=mask)
    assert np.array_equal(flow_io.read_png(p), edge['png_raw_pypng'])
    tools.write_flow_png(p, uv[:, :, 0], uv[:, :, 1])
    assert np.array_equal(flow_io.read_png(p), edge['png_raw_pypng_nomask'])
    tools.write_kitti_png_file(p, edge['png_uv_cv2'], mask)
    assert np.array_equal(flow_io.read_png(p), edge['png_raw_cv2_bgr'][:, :, ::-1])      # cv2 stores BGR -> the file's RGB


def test_kitti_png_decode_equals_the_references(edge, tmp_path):
    p = str(tmp_path / 'k.png')
    open(p, 'wb').write(edge['png_file_bytes'].tobytes())
    for rd in (img_func.read_png_flow, img_func.read_flow, flow_io.read_kitti_png_flow):
        f, m = rd(p)
        assert f.dtype == np.float64 and np.array_equal(f, edge['png_read_flow'])
        assert m.dtype == np.uint8 and np.array_equal(m, edge['png_read_mask'])


def test_frame_normalisation_and_names_equal_the_references(edge):
    img = edge['img']
    for key, kw in (('img_norm', dict(normalize=True)), ('img_norm_flip', dict(normalize=True, if_horizontal_flip=True)),
                    ('img_raw', dict(normalize=False))):
        got = img_func.get_process_img_only_img(img, **kw)
        assert got.shape == edge[key].shape and np.array_equal(np.asarray(got, dtype=np.float64), edge[key].astype(np.float64))
    assert [img_func.frame_name_to_num(str(n)) for n in edge['frame_names']] == list(edge['frame_nums'])


def test_epe_and_f1_equal_the_references(edge):
    EB = kitti_flow.Evaluation_bench
    gt, pred = torch.from_numpy(edge['ev_gt']), torch.from_numpy(edge['ev_pred'])

    def same(got, want):
        got, want = float(got), float(want[0])
        return (np.isnan(got) and np.isnan(want)) or got == pytest.approx(want, rel=1e-6, abs=1e-7)
    for k in ('rand', 'full', 'empty'):
        m = torch.from_numpy(edge['ev_mask_' + k])
        assert same(EB.flow_error_avg(gt, pred, m), edge['ev_epe_' + k]), k
        assert same(EB.outlier_pct(gt, pred, m), edge['ev_f1_' + k]), k          # empty mask: nan on both sides (0 / 0)
    m = torch.from_numpy(edge['ev_mask_rand'])
    assert same(EB.outlier_pct(gt, pred, m, threshold=2.0, relative=None), edge['ev_f1_rand_abs'])
    assert same(EB.outlier_pct(gt, pred, m, threshold=1.0, relative=0.1), edge['ev_f1_rand_t1'])
    assert same(EB.flow_error_avg(gt, gt + 100.0, m), edge['ev_epe_all_outliers'])
    assert same(EB.outlier_pct(gt, gt + 100.0, m), edge['ev_f1_all_outliers']) and float(edge['ev_f1_all_outliers'][0]) == pytest.approx(100.0)
    assert same(EB.outlier_pct(gt, gt, m), edge['ev_f1_exact']) and float(edge['ev_f1_exact'][0]) == 0.0
