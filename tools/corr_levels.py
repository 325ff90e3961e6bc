#!/usr/bin/env python3
"""Per-level timing of the cost volume (forward, bf16) at the five pyramid levels of BASELINE configs 2-5:
the all-channels-in-LDS kernel (auto / each forced tile geometry), the channel-chunked round-1 kernels, and the fused
normalisation path.  Times: graph replay of NREP launches (kernel + ~1.5 us dependent-launch boundary), like tools/kbench.py;
`ev` = upf_corr81_forward_timed (HIP events around each launch, includes the ~4 us event floor).
Writes a table to stdout and JSON to gpurun_out/corr_levels.json.  Run under gpurun."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upflow_pytorch_amd import ops
from tools.kbench import graph_time

LEVELS = {2: (4, [(196, 6, 20), (128, 12, 40), (96, 24, 80), (64, 48, 160), (32, 96, 320)]),
          3: (4, [(196, 4, 13), (128, 8, 26), (96, 16, 52), (64, 32, 104), (32, 64, 208)]),
          4: (8, [(196, 7, 16), (128, 14, 32), (96, 28, 64), (64, 56, 128), (32, 112, 256)]),
          5: (1, [(196, 15, 45), (128, 30, 90), (96, 60, 180), (64, 120, 360), (32, 240, 720)])}


def main(cfgs, stacked):
    res = []
    for cfg in cfgs:
        B, lv = LEVELS[cfg]
        if stacked:
            B *= 2                                   # what the model launches: both flow directions stacked along the batch
        tot_bytes, tot = 0, {}
        for (C, H, W) in lv:
            f = torch.randn(2, B, C, H, W, device='cuda').bfloat16()
            out = torch.empty(B, 81, H, W, device='cuda', dtype=torch.bfloat16)
            byt = 2 * B * H * W * (2 * C + 81)
            row = dict(cfg=cfg, B=B, C=C, H=H, W=W, bytes=byt)
            ops.corr_set_option('old_path', 1)
            row['old'] = graph_time(lambda: ops.corr81_forward_raw(f[0], f[1], out=out, leaky_slope=0.1))
            ops.corr_set_option('old_path', 0)
            row['new'] = graph_time(lambda: ops.corr81_forward_raw(f[0], f[1], out=out, leaky_slope=0.1))
            row['new_ev'] = ops.corr81_forward_timed(f[0], f[1], out, 0.1, nrep=100)[0]
            for v in range(4):
                ops.corr_set_option('variant', v)
                row['v%d' % v] = graph_time(lambda: ops.corr81_forward_raw(f[0], f[1], out=out, leaky_slope=0.1))
            ops.corr_set_option('variant', -1)
            row['norm+corr'] = graph_time(lambda: ops.corr81_forward_raw(*ops.normalize(f.view(2 * B, C, H, W)).view(2, B, C, H, W).unbind(0), out=out, leaky_slope=0.1))
            row['fused'] = graph_time(lambda: ops.corr81_norm_forward_raw(f[0], f[1], out=out, leaky_slope=0.1))
            res.append(row)
            tot_bytes += byt
            for k in ('old', 'new', 'norm+corr', 'fused'):
                tot[k] = tot.get(k, 0.0) + row[k]
            print('cfg%d B%-2d C%3d %4dx%-4d  old %7.2f  new %7.2f us (%5.1f%% of 8 TB/s; ev %6.2f)  tiles 8x32 %6.2f 4x32 %6.2f 2x32 %6.2f 4x16 %6.2f | normalize+corr %7.2f  fused %7.2f'
                  % (cfg, B, C, H, W, row['old'], row['new'], byt / row['new'] / 80e3, row['new_ev'], row['v0'], row['v1'], row['v2'], row['v3'],
                     row['norm+corr'], row['fused']), flush=True)
        print('cfg%d five levels: %.1f MB; old %.1f us = %.1f%%   new %.1f us = %.1f%% of 8 TB/s;  normalize+corr %.1f us -> fused %.1f us'
              % (cfg, tot_bytes / 1e6, tot['old'], tot_bytes / tot['old'] / 80e3, tot['new'], tot_bytes / tot['new'] / 80e3, tot['norm+corr'], tot['fused']), flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(res, open('gpurun_out/corr_levels%s.json' % ('_stacked' if stacked else ''), 'w'), indent=1)


if __name__ == '__main__':
    a = [x for x in sys.argv[1:] if x != '--stacked']
    main([int(x) for x in a] or [2, 3, 4, 5], '--stacked' in sys.argv)
