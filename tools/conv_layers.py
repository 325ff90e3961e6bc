#!/usr/bin/env python3
"""Per-layer table of every convolution of one config-2 inference step (384x1280, bf16, 2B = 8 stacked items):
us / TFLOP/s / GB/s per launch (hipGraph replay of NREP launches, tools/kbench.py's clock), per pyramid level.

    python tools/conv_layers.py                 # the table with the library's launch heuristics
    python tools/conv_layers.py --sweep         # additionally every layer under the experiment switches
                                                # (force_sk, force_mtw, ph_fit): which launch shape is fastest
Writes gpurun_out/conv_layers.json.  Layer list: /root/reference/model/pwc_modules.py:122-142 (pyramid), :250-286
(estimator), :396-412 (context network), model/upflow.py:24-60 (SGU estimator + guidance stem), :349-353 (1x1).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops

NREP = 20


def graph_time(fn, iters=6):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(NREP):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(iters):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / NREP * 1e3)
    return best


def layers(H, W):
    """(name, Cin, Cout, k, dilation, stride, H, W) of the decoder convolutions at one pyramid level."""
    L = []
    cin = 115
    for i, f in enumerate((128, 128, 96, 64, 32)):
        L.append(('est.conv%d' % (i + 1), cin, f, 3, 1, 1, H, W)); cin += f
    L.append(('est.conv_last', cin, 2, 3, 1, 1, H, W))
    ch = (565, 128, 128, 128, 96, 64, 32, 2)
    for i, d in enumerate((1, 2, 4, 8, 16, 1, 1)):
        L.append(('ctx.conv%d' % i, ch[i], ch[i + 1], 3, d, 1, H, W))
    cin = 64
    for i, f in enumerate((32, 32, 32, 16, 8)):
        L.append(('sgu.conv%d' % (i + 1), cin, f, 3, 1, 1, H, W)); cin += f
    L.append(('sgu.conv_last', cin, 3, 3, 1, 1, H, W))
    return L


def stem_layers(H, W):
    L = []
    chs = [3, 16, 32, 64, 96, 128, 196]
    h, w = H, W
    for i in range(6):
        L.append(('pyr%d.s2' % i, chs[i], chs[i + 1], 3, 1, 2, h, w))
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        L.append(('pyr%d.s1' % i, chs[i + 1], chs[i + 1], 3, 1, 1, h, w))
    L += [('sgu_stem0', 3, 16, 3, 1, 1, H, W), ('sgu_stem1.s2', 16, 16, 3, 1, 2, H, W),
          ('sgu_stem2', 16, 32, 3, 1, 1, H // 2, W // 2), ('sgu_stem3.s2', 32, 32, 3, 1, 2, H // 2, W // 2)]
    for c, (h, w) in zip((196, 128, 96, 64, 32), [(6, 20), (12, 40), (24, 80), (48, 160), (96, 320)]):
        L.append(('conv_1x1[%d]' % c, c, 32, 1, 1, 1, h, w))
    return L


def bench_layer(B, Cin, Cout, k, d, stride, H, W, dt=torch.bfloat16):
    x = torch.randn(B, Cin, H, W, device='cuda').to(dt)
    w = (torch.randn(Cout, Cin, k, k, device='cuda') * 0.02).to(dt)
    b = torch.randn(Cout, device='cuda')
    Ho, Wo = ops.conv3x3_out_hw(H, W, stride)
    y = torch.empty(B, Cout, Ho, Wo, device='cuda', dtype=dt)
    packed = ops.conv3x3_pack(w)
    t = graph_time(lambda: ops.conv3x3_forward_raw(x, packed, b, y, d, 0.1, stride, k))
    flop = 2.0 * B * Ho * Wo * Cin * Cout * k * k
    byt = 2.0 * B * (Cin * H * W + Cout * Ho * Wo)
    return t, flop / t / 1e6, byt / t / 1e3


def bench_layer_c8(B, C8c, C2, Cout, d, H, W, y_c8, dt=torch.bfloat16):
    """The same layer through upf_conv_forward_c8: input channels [0, C8c) in a C8 buffer, the last C2 as NCHW planes."""
    Cin = C8c + C2
    x8 = torch.randn(B, (C8c + 7) // 8, H, W, 8, device='cuda').to(dt)
    x2 = torch.randn(B, C2, H, W, device='cuda').to(dt) if C2 else None
    w = (torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.02).to(dt)
    b = torch.randn(Cout, device='cuda')
    y = ops.c8_empty(B, Cout, H, W, dt, 'cuda') if y_c8 else torch.empty(B, Cout, H, W, device='cuda', dtype=dt)
    c8_map = list(range(C8c)) + [-1] * ((C8c + 7) // 8 * 8 - C8c)
    packed = ops.conv_c8_pack(w, c8_map, list(range(C8c, Cin)))
    t = graph_time(lambda: ops.conv_c8_forward_raw(x8, x2, packed, b, y, d, 0.1))
    flop = 2.0 * B * H * W * Cin * Cout * 9
    return t, flop / t / 1e6, 2.0 * B * H * W * (Cin + Cout) / t / 1e3


def main_c8():
    """Decoder layers at the two fine levels in the mixed layout: est.* read [C8 suffix | corr81 + flow planes]."""
    B = 8
    rows = []
    for (H, W) in [(96, 320), (48, 160)]:
        items = []
        c8 = 32
        for i, f in enumerate((128, 128, 96, 64, 32)):
            items.append(('est.conv%d' % (i + 1), c8, 83, f, 1, True)); c8 += f
        items.append(('est.conv_last', c8, 83, 2, 1, False))
        items.append(('ctx.conv0', 480, 85, 128, 1, True))
        if '--pure' in sys.argv:                     # what the estimator would cost if corr81 / features / flows were octets too
            c8 = 120                                 # 81 + 32 + 2 + 2 = 117 channels -> 15 octets
            for i, f in enumerate((128, 128, 96, 64, 32)):
                items.append(('est.conv%d/pure' % (i + 1), c8, 0, f, 1, True)); c8 += f
            items.append(('est.conv_last/pure', c8, 0, 2, 1, False))
            items.append(('ctx.conv0/pure', 568, 0, 128, 1, True))
        ch = (128, 128, 128, 96, 64, 32, 2)
        for i, d in enumerate((2, 4, 8, 16, 1)):
            items.append(('ctx.conv%d' % (i + 1), ch[i], 0, ch[i + 1], d, True))
        items.append(('ctx.conv6', 32, 0, 2, 1, False))
        cin = 64
        for i, f in enumerate((32, 32, 32, 16, 8)):
            items.append(('sgu.conv%d' % (i + 1), cin, 0, f, 1, True)); cin += f
        items.append(('sgu.conv_last', cin, 0, 3, 1, False))
        tot_c8 = tot_nchw = 0.0
        for (name, C8c, C2, Cout, d, y_c8) in items:
            t, tf, gb = bench_layer_c8(B, C8c, C2, Cout, d, H, W, y_c8)
            t0, tf0, gb0 = bench_layer(B, C8c + C2, Cout, 3, d, 1, H, W)
            alt = ''
            if Cout <= 32:
                prev = ops.conv_c8_set_option('rpw4', 0)
                alt = '  (8-row tiles %.1f us)' % bench_layer_c8(B, C8c, C2, Cout, d, H, W, y_c8)[0]
                ops.conv_c8_set_option('rpw4', prev)
            tot_c8 += t; tot_nchw += t0
            rows.append(dict(level='%dx%d' % (H, W), layer=name, C8=C8c, C2=C2, Cout=Cout, d=d, us_c8=t, us_nchw=t0, TFs_c8=tf))
            print('%4dx%-4d %-14s %3d+%2d->%3d d%-2d  C8 %7.1f us %7.1f TF/s (%4.1f%%) %6.0f GB/s | NCHW %7.1f us  x%.2f%s' %
                  (H, W, name, C8c, C2, Cout, d, t, tf, tf / 25.0, gb, t0, t0 / t, alt), flush=True)
        print('sum %dx%d: C8 %.1f us, NCHW %.1f us' % (H, W, tot_c8, tot_nchw))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(rows, open('gpurun_out/conv_layers_c8.json', 'w'), indent=1)


def main_fp32():
    """The same layer list through the split-precision kernel of the parity mode (csrc/conv_x3.hip), beside the bf16 kernel:
    effective TFLOP/s = the layer's algorithmic flop / time (the kernel issues three matrix products per operand pair)."""
    B = 8
    levels = [(96, 320), (48, 160), (24, 80), (12, 40), (6, 20)]
    items = [('stem',) + l for l in stem_layers(384, 1280)]
    for (H, W) in levels:
        items += [('%dx%d' % (H, W),) + l for l in layers(H, W)]
    sums = {}
    for (lvl, name, Cin, Cout, k, d, s, H, W) in items:
        t, tf, gb = bench_layer(B, Cin, Cout, k, d, s, H, W, dt=torch.float32)
        t16, tf16, _ = bench_layer(B, Cin, Cout, k, d, s, H, W)
        a = sums.setdefault(lvl, [0.0, 0.0]); a[0] += t; a[1] += t16
        extra = ''
        if '--sweep' in sys.argv:                    # the two kernels forced (tiled 8x32 / split-K 2x32), to set "sk_max_tiles"
            alt = {}
            for nm, v in (('tiled', 0), ('splitk', 1 << 30)):
                prev = ops.conv_x3_set_option('sk_max_tiles', v)
                try:
                    alt[nm] = bench_layer(B, Cin, Cout, k, d, s, H, W, dt=torch.float32)[0]
                finally:
                    ops.conv_x3_set_option('sk_max_tiles', prev)
            wgs = B * ((W + 31) // 32) * ((H + 7) // 8) * ((Cout + 31) // 32)
            extra = '  | tiled %.1f splitk %.1f (8x32-tile workgroups %d)' % (alt['tiled'], alt['splitk'], wgs)
        print('%-8s %-14s %3d->%3d k%d d%-2d s%d %4dx%-4d  x3 %8.1f us %7.1f TF/s eff. (x3 issued: %4.1f%% of 2.5 PF) %7.0f GB/s | bf16 %7.1f us  x%.2f%s' %
              (lvl, name, Cin, Cout, k, d, s, H, W, t, tf, 3 * tf / 25.0, 2 * gb, t16, t / t16, extra), flush=True)
    for lvl, (a, b) in sums.items():
        print('sum %-8s x3 %8.1f us   bf16 %8.1f us   x%.2f' % (lvl, a, b, a / b))
    print('sum all x3 %8.1f us   bf16 %8.1f us' % (sum(a for a, _ in sums.values()), sum(b for _, b in sums.values())))


def main():
    if '--c8' in sys.argv:
        return main_c8()
    if '--fp32' in sys.argv:
        return main_fp32()
    sweep = '--sweep' in sys.argv
    B = 8
    levels = [(96, 320), (48, 160), (24, 80), (12, 40), (6, 20)]
    rows = []
    items = [('stem',) + l for l in stem_layers(384, 1280)]
    for (H, W) in levels:
        items += [('%dx%d' % (H, W),) + l for l in layers(H, W)]
    variants = [('auto', {})]
    if sweep:
        variants += [('sk0', {'force_sk': 0}), ('sk1', {'force_sk': 1}),
                     ('sk0_mtw1', {'force_sk': 0, 'force_mtw': 1}), ('sk0_mtw2', {'force_sk': 0, 'force_mtw': 2}),
                     ('sk0_mtw4', {'force_sk': 0, 'force_mtw': 4}), ('ph0', {'ph_fit': 0}), ('rpw4_0', {'rpw4_min': 1 << 30})]
    for (lvl, name, Cin, Cout, k, d, s, H, W) in items:
        res = {}
        for vn, opts in variants:
            prev = {kk: ops.conv_set_option(kk, vv) for kk, vv in opts.items()}
            try:
                res[vn] = bench_layer(B, Cin, Cout, k, d, s, H, W)
            finally:
                for kk, vv in prev.items():
                    ops.conv_set_option(kk, vv)
        t, tf, gb = res['auto']
        best = min(res, key=lambda v: res[v][0])
        rows.append(dict(level=lvl, layer=name, Cin=Cin, Cout=Cout, k=k, d=d, stride=s, H=H, W=W, us=t, TFs=tf, GBs=gb,
                         variants={v: r[0] for v, r in res.items()}, best=best))
        extra = ''
        if sweep:
            extra = '  best %-9s %7.1f us  | ' % (best, res[best][0]) + ' '.join('%s %.1f' % (v, r[0]) for v, r in res.items() if v != 'auto')
        print('%-8s %-14s %3d->%3d k%d d%-2d s%d %4dx%-4d %8.1f us %7.1f TF/s (%4.1f%% of 2.5 PF) %7.0f GB/s%s' %
              (lvl, name, Cin, Cout, k, d, s, H, W, t, tf, tf / 25.0, gb, extra), flush=True)
    for lvl in ['stem'] + ['%dx%d' % l for l in levels]:
        sel = [r for r in rows if r['level'] == lvl]
        print('sum %-8s %8.1f us   (best variants: %8.1f us)' % (lvl, sum(r['us'] for r in sel), sum(min(r['variants'].values()) for r in sel)))
    print('sum all %8.1f us' % sum(r['us'] for r in rows))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(rows, open('gpurun_out/conv_layers.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
