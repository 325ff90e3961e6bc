#!/usr/bin/env python3
"""The warp's backward kernel with and without the neighbour-lane merge of its fixed-point atomics (csrc/warp.hip, round 4):
both builds of warp.hip (-DUPF_WARP_MERGE=1 / =0) as stand-alone libraries, the same inputs, every output bit compared, both timed.

    python tools/warp_bwd_ab.py        # on the GPU box (needs hipcc)
"""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, 'upflow_pytorch_amd', 'csrc')


def build(merge):
    out = '/tmp/libwarp_merge%d.so' % merge
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
           '-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops', '-DUPF_WARP_MERGE=%d' % merge, '-I', CS, '-I', os.path.join(ROOT, 'include'),
           os.path.join(CS, 'warp.hip'), os.path.join(CS, 'api.hip'), '-o', out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    L = C.CDLL(out)
    L.upf_warp_backward_workspace_bytes.restype = C.c_longlong
    L.upf_warp_backward.argtypes = [C.c_void_p] * 6 + [C.c_int] * 7 + [C.c_void_p]
    return L


def run(L, x, flow, gy, mask_mode, shift, reps=0):
    B, Cc, H, W = x.shape
    gx = torch.empty_like(x)
    gf = torch.empty(B, 2, H, W, device='cuda')
    ws = torch.empty(L.upf_warp_backward_workspace_bytes(B, Cc, H, W), dtype=torch.uint8, device='cuda')
    code = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[x.dtype]
    st = torch.cuda.current_stream().cuda_stream

    def go():
        rc = L.upf_warp_backward(x.data_ptr(), flow.data_ptr(), gy.data_ptr(), gx.data_ptr(), gf.data_ptr(), ws.data_ptr(), B, Cc, H, W, code, mask_mode, shift, st)
        assert rc == 0, rc
    go()
    torch.cuda.synchronize()
    t = None
    if reps:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            go()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e3
    return gx, gf, t


def main():
    from upflow_pytorch_amd import _lib
    codes = {0: 'fp32', 1: 'fp16', 2: 'bf16'}
    assert _lib.dtype_code(torch.zeros(1, dtype=torch.bfloat16)) == 2 and _lib.dtype_code(torch.zeros(1, dtype=torch.float16)) == 1, codes
    L1, L0 = build(1), build(0)
    g = torch.Generator().manual_seed(5)
    cases = [(8, 32, 64, 208, torch.bfloat16, 3.0), (8, 64, 32, 104, torch.bfloat16, 2.0), (8, 96, 16, 52, torch.bfloat16, 1.0), (8, 196, 4, 13, torch.bfloat16, 0.5),
             (8, 3, 256, 832, torch.float32, 8.0), (2, 5, 17, 23, torch.float32, 30.0), (2, 7, 9, 64, torch.float16, 0.0), (1, 4, 33, 1, torch.float32, 2.0),
             (4, 32, 64, 208, torch.float32, 3.0)]
    for (B, Cc, H, W, dt, mag) in cases:
        x = torch.randn(B, Cc, H, W, generator=g).cuda().to(dt)
        gy = torch.randn(B, Cc, H, W, generator=g).cuda().to(dt)
        smooth = torch.nn.functional.interpolate(torch.randn(B, 2, max(H // 8, 1), max(W // 8, 1), generator=g), size=(H, W), mode='bilinear', align_corners=True)
        flow = (smooth * mag + 0.05 * torch.randn(B, 2, H, W, generator=g)).cuda()
        flow[:, :, H // 2:, W // 3:] += 2.5 * mag                               # a motion boundary
        for mask_mode in (0, 1, 2):
            for shift in (0, B // 2):
                a = run(L1, x, flow, gy, mask_mode, shift, reps=20 if mask_mode == 1 and shift == 0 else 0)
                b = run(L0, x, flow, gy, mask_mode, shift, reps=20 if mask_mode == 1 and shift == 0 else 0)
                assert torch.equal(a[0].view(torch.uint8), b[0].view(torch.uint8)), ('gx differs', B, Cc, H, W, dt, mask_mode, shift)
                assert torch.equal(a[1].view(torch.uint8), b[1].view(torch.uint8)), ('gflow differs', B, Cc, H, W, dt, mask_mode, shift)
                if a[2] is not None:
                    print('[%d,%3d,%3d,%3d] %-8s |flow| ~ %4.1f px: merged %7.2f us   one atomic per tap %7.2f us   x%.2f   (bit-identical)' %
                          (B, Cc, H, W, str(dt).replace('torch.', ''), mag, a[2], b[2], b[2] / a[2]), flush=True)
    # overflow / NaN still surface
    x = torch.randn(1, 2, 8, 64).cuda(); gy = torch.randn(1, 2, 8, 64).cuda(); flow = torch.zeros(1, 2, 8, 64).cuda()
    gy[0, 0, 3, 10] = float('nan'); gy[0, 1, 4, 20] = 1e30
    a = run(L1, x, flow, gy, 0, 0); b = run(L0, x, flow, gy, 0, 0)
    assert torch.isnan(a[0][0, 0, 3, 10]) and torch.isnan(a[0][0, 1, 4, 20]) and torch.equal(torch.isnan(a[0]), torch.isnan(b[0]))
    print('all cases bit-identical; NaN / overflow poison the same elements')


if __name__ == '__main__':
    main()
