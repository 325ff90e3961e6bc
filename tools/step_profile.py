#!/usr/bin/env python3
"""Per-kernel breakdown of ONE steady-state training step from a rocprofv3 kernel-trace CSV: steps are delimited by the
optimiser's first kernel (multi_tensor_apply of Adam); the last complete step is reported."""
import csv
import sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'occ_check' in r['Kernel_Name']]          # one per forward
k = len(marks) - 2
seg = rows[marks[k]:marks[k + 1]]
agg = defaultdict(lambda: [0, 0.0, 1e18])
for r in seg:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Kernel_Name'][:118]
    agg[n][0] += 1; agg[n][1] += d; agg[n][2] = min(agg[n][2], d)
tot = sum(v[1] for v in agg.values())
wall = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3
print('# one steady-state training step (occ_check to occ_check): %d kernels, GPU time %.1f us, wall %.1f us' % (len(seg), tot, wall))
print('%-118s %6s %10s %6s %9s' % ('kernel', 'calls', 'total_us', 'pct', 'avg_us'))
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-118s %6d %10.1f %6.2f %9.2f' % (n, v[0], v[1], 100 * v[1] / tot, v[1] / v[0]))
