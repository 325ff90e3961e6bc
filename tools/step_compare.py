#!/usr/bin/env python3
"""Which kernels of a training step do NOT scale with the batch: per-kernel totals of one steady-state step from two rocprofv3
kernel traces (batch A, batch B), side by side, sorted by the time that stays.   python tools/step_compare.py a.csv b.csv"""
import csv, sys, re
from collections import defaultdict


def load(f):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    occ = [i for i, r in enumerate(rows) if 'occ_check' in r['Kernel_Name']]
    k = len(occ) - 2
    seg = rows[occ[k - 1] + 1:occ[k] + 1]
    agg = defaultdict(lambda: [0, 0.0])
    for r in seg:
        n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '').replace('upf::', '').replace('at::native::', 'at::')[:100]
        agg[n][0] += 1; agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    return agg


a, b = load(sys.argv[1]), load(sys.argv[2])
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print('# GPU time: A %.1f us, B %.1f us' % (ta, tb))
print('%-100s %5s %9s %9s %7s' % ('kernel', 'calls', 'A_us', 'B_us', 'B/A'))
for n in sorted(set(a) | set(b), key=lambda n: -min(a.get(n, [0, 0])[1], b.get(n, [0, 0])[1])):
    va, vb = a.get(n, [0, 0.0]), b.get(n, [0, 0.0])
    print('%-100s %5d %9.1f %9.1f %7.2f' % (n, vb[0], va[1], vb[1], vb[1] / va[1] if va[1] else 0))
