"""File formats at the data / evaluation edge of the hot path, without cv2 / pypng / imageio /
TensorFlow: Middlebury `.flo`, KITTI 16-bit flow PNG, 8-bit frame PNG.

Mirrors the reference's readers / writers (all numpy, H x W x C host arrays):
    read_flo / read_flow / write_flo / write_flow        utils/tools.py:1556-1632
    write_kitti_png_file / write_flow_png                utils/tools.py:1482-1525   (R = u*64+2^15, G = v*64+2^15, B = valid)
    read_kitti_png_flow                                  dataset/kitti_dataset.py:104-145 (img_func.read_flow / read_png_flow)
    read_image                                           dataset/kitti_dataset.py:45-54   (tf.image.decode_image)
The PNG codec below is the subset those files use: non-interlaced, bit depth 8 / 16, colour types 0 / 2 / 4 / 6, all
five scan-line filters on read; filter 0 / Up on write.  zlib does the (de)compression.
"""
import struct
import zlib

import numpy as np

FLO_MAGIC = 202021.25
_PNG_SIG = b'\x89PNG\r\n\x1a\n'
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


# ---------------------------------------------------------------------------------------------- Middlebury .flo
def write_flo(flow, filename):
    """flow [H,W,2] float -> .flo (magic 202021.25, int32 width, int32 height, row-major (u,v) float32)."""
    flow = np.asarray(flow)
    if flow.ndim != 3 or flow.shape[2] != 2:
        raise ValueError('write_flo: [H,W,2] expected, got %s' % (flow.shape,))
    h, w = flow.shape[:2]
    with open(filename, 'wb') as f:
        np.array([FLO_MAGIC], dtype=np.float32).tofile(f)
        np.array([w], dtype=np.int32).tofile(f)
        np.array([h], dtype=np.int32).tofile(f)
        flow.astype(np.float32).tofile(f)


def read_flo(filename):
    """.flo -> [H,W,2] float32.  Raises on a bad magic number or a truncated file (the reference prints and returns None)."""
    with open(filename, 'rb') as f:
        head = f.read(12)
        if len(head) < 12:
            raise ValueError('%s: truncated .flo header' % filename)
        magic, w, h = struct.unpack('<fii', head)
        if magic != np.float32(FLO_MAGIC):
            raise ValueError('%s: magic number incorrect (%r), invalid .flo file' % (filename, magic))
        if w <= 0 or h <= 0 or w * h > (1 << 28):
            raise ValueError('%s: implausible size %dx%d' % (filename, w, h))
        data = np.fromfile(f, np.float32, count=2 * w * h)
    if data.size != 2 * w * h:
        raise ValueError('%s: truncated .flo payload (%d of %d floats)' % (filename, data.size, 2 * w * h))
    return data.reshape(h, w, 2)


write_flow, read_flow = write_flo, read_flo


# ---------------------------------------------------------------------------------------------- PNG subset
def _chunk(tag, payload):
    return struct.pack('>I', len(payload)) + tag + payload + struct.pack('>I', zlib.crc32(tag + payload) & 0xffffffff)


def write_png(filename, img, compression=3):
    """img: [H,W] or [H,W,C] uint8 / uint16, C in 1..4 -> PNG (16-bit samples big-endian, as the format requires)."""
    img = np.asarray(img)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    if img.dtype not in (np.uint8, np.uint16) or c not in (1, 2, 3, 4):
        raise ValueError('write_png: uint8 / uint16 image with 1..4 channels expected, got %s %s' % (img.dtype, img.shape))
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    depth = 8 * img.dtype.itemsize
    rows = np.ascontiguousarray(img.astype('>u2') if depth == 16 else img).view(np.uint8).reshape(h, -1)
    # scan-line filter "Up" (2): byte-wise difference to the previous row — vectorised, and it compresses flow fields well
    up = np.empty_like(rows)
    up[0] = rows[0]
    up[1:] = rows[1:] - rows[:-1]
    raw = np.concatenate([np.full((h, 1), 2, dtype=np.uint8), up], axis=1).tobytes()
    with open(filename, 'wb') as f:
        f.write(_PNG_SIG)
        f.write(_chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, depth, ctype, 0, 0, 0)))
        f.write(_chunk(b'IDAT', zlib.compress(raw, compression)))
        f.write(_chunk(b'IEND', b''))


def _unfilter(raw, h, stride, bpp):
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.uint8)
    pos = 0
    for y in range(h):
        ft = raw[pos]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=pos + 1)
        pos += stride + 1
        if ft == 0:
            cur = line.copy()
        elif ft == 2:                                       # Up
            cur = line + prev
        elif ft == 1:                                       # Sub: running sum per byte lane
            cur = np.cumsum(line.reshape(-1, bpp), axis=0, dtype=np.uint64).astype(np.uint8).reshape(-1) if stride % bpp == 0 else None
            if cur is None:
                raise ValueError('PNG: row length is not a multiple of the pixel size')
        elif ft in (3, 4):                                  # Average / Paeth: sequential along the row
            cur = bytearray(line.tobytes())
            pv = prev.tobytes()
            if ft == 3:
                for i in range(stride):
                    a = cur[i - bpp] if i >= bpp else 0
                    cur[i] = (cur[i] + ((a + pv[i]) >> 1)) & 0xff
            else:
                for i in range(stride):
                    a = cur[i - bpp] if i >= bpp else 0
                    b = pv[i]
                    c = pv[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                    cur[i] = (cur[i] + pr) & 0xff
            cur = np.frombuffer(bytes(cur), dtype=np.uint8)
        else:
            raise ValueError('PNG: unknown scan-line filter %d' % ft)
        out[y] = cur
        prev = out[y]
    return out


def read_png(filename):
    """PNG -> [H,W,C] uint8 / uint16 array (C = 1, 2, 3, 4).  Non-interlaced, bit depth 8 / 16, no palette."""
    with open(filename, 'rb') as f:
        data = f.read()
    if data[:8] != _PNG_SIG:
        raise ValueError('%s: not a PNG file' % filename)
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        n, tag = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if len(body) != n:
            raise ValueError('%s: truncated chunk %r' % (filename, tag))
        (crc,) = struct.unpack('>I', data[pos + 8 + n:pos + 12 + n])
        if crc != (zlib.crc32(tag + body) & 0xffffffff):
            raise ValueError('%s: CRC mismatch in chunk %r' % (filename, tag))
        if tag == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif tag == b'IDAT':
            idat.append(body)
        elif tag == b'IEND':
            break
        pos += 12 + n
    if hdr is None or not idat:
        raise ValueError('%s: missing IHDR / IDAT' % filename)
    w, h, depth, ctype, _, _, interlace = hdr
    if depth not in (8, 16) or ctype not in _CHANNELS or interlace != 0:
        raise ValueError('%s: unsupported PNG (depth %d, colour type %d, interlace %d)' % (filename, depth, ctype, interlace))
    c = _CHANNELS[ctype]
    bpp = c * depth // 8
    stride = w * bpp
    raw = zlib.decompress(b''.join(idat))
    if len(raw) != h * (stride + 1):
        raise ValueError('%s: decompressed size %d, expected %d' % (filename, len(raw), h * (stride + 1)))
    rows = _unfilter(raw, h, stride, bpp)
    if depth == 16:
        return rows.view('>u2').astype(np.uint16).reshape(h, w, c)
    return rows.reshape(h, w, c)


def read_image(filename):
    """8-bit frame -> [H,W,3] uint8 RGB (what tf.image.decode_image returns, dataset/kitti_dataset.py:45-54)."""
    img = read_png(filename)
    if img.dtype != np.uint8:
        img = (img >> 8).astype(np.uint8)
    if img.shape[2] == 1:
        img = np.repeat(img, 3, axis=2)
    return img[:, :, :3]


# ---------------------------------------------------------------------------------------------- KITTI flow PNG
def write_kitti_png_file(flow_fn, flow_data, mask_data=None):
    """flow [H,W,2] (u,v) float, mask [H,W] -> KITTI 16-bit PNG: R = u*64 + 2^15, G = v*64 + 2^15, B = valid
    (utils/tools.py:1515-1525 through cv2's BGR order; :1482-1513 through pypng's RGB order — the same file)."""
    flow_data = np.asarray(flow_data, dtype=np.float64)
    if flow_data.ndim != 3 or flow_data.shape[2] != 2:
        raise ValueError('write_kitti_png_file: [H,W,2] expected, got %s' % (flow_data.shape,))
    h, w = flow_data.shape[:2]
    valid = np.ones((h, w), dtype=np.uint16) if mask_data is None else np.asarray(mask_data).reshape(h, w).astype(np.uint16)
    enc = np.clip(flow_data * 64.0 + 2 ** 15, 0.0, 65535.0).astype(np.uint16)     # (write_flow_png clips; so do we)
    write_png(flow_fn, np.stack([enc[:, :, 0], enc[:, :, 1], valid], axis=-1))


def write_flow_png(filename, uv, v=None, mask=None):
    uv = np.asarray(uv)
    flow = uv if v is None else np.stack([uv, np.asarray(v)], axis=-1)
    write_kitti_png_file(filename, flow, mask)


def read_kitti_png_flow(fpath):
    """KITTI flow PNG -> (flow [2,H,W] float64, mask [1,H,W] uint8), channel-first like img_func.read_png_flow
    (dataset/kitti_dataset.py:129-145)."""
    gt = read_png(fpath)
    if gt.dtype != np.uint16 or gt.shape[2] < 3:
        raise ValueError('%s: a 16-bit RGB PNG expected, got %s %s' % (fpath, gt.dtype, gt.shape))
    flow = (gt[:, :, 0:2].astype('float64') - 2 ** 15) / 64.0
    mask = np.uint8(gt[:, :, 2:3])
    return np.transpose(flow, [2, 0, 1]), np.transpose(mask, [2, 0, 1])
