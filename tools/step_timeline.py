#!/usr/bin/env python3
"""ONE steady-state step of a rocprofv3 kernel trace CSV as an ordered timeline: start offset, duration, gap to the previous
kernel, grid (workgroups), short kernel name — and, at the end, the time per PHASE (runs of the same kernel family).
    python tools/step_timeline.py trace.csv [step index] [--full]"""
import csv, re, sys
args = [a for a in sys.argv[1:] if not a.startswith('--')]
full = '--full' in sys.argv
rows = list(csv.DictReader(open(args[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
occ = [i for i, r in enumerate(rows) if 'occ_check' in r['Kernel_Name']]
k = int(args[1]) if len(args) > 1 else len(occ) - 2
seg = rows[occ[k - 1] + 1:occ[k] + 1]


def short(n):
    n = re.sub(r'\(.*', '', n)
    n = n.replace('void ', '').replace('upf::', '').replace('at::native::', 'at::')
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return n[:90]


def wgs(r):
    try:
        g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
        w = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
        return g // max(w, 1)
    except Exception:
        return -1


t0 = int(seg[0]['Start_Timestamp'])
prev_end = t0
gaps = 0.0
print('# kernels %d  wall %.1f us' % (len(seg), (int(seg[-1]['End_Timestamp']) - t0) / 1e3))
print('%9s %8s %7s %7s  %s' % ('start_us', 'dur_us', 'gap_us', 'wgs', 'kernel'))
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3
    gaps += max(gap, 0.0)
    if full or 'wgrad' in r['Kernel_Name'] or (e - s) > 20000:
        print('%9.1f %8.2f %7.2f %7d  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, wgs(r), short(r['Kernel_Name'])))
    prev_end = max(prev_end, e)
print('# sum of gaps %.1f us' % gaps)
# coarse phases: cumulative kernel time in windows of 500 us, split into "big" (>= 128 workgroups) and "small" launches
win = 500.0
acc = {}
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    w = int(((s - t0) / 1e3) // win)
    a = acc.setdefault(w, [0, 0.0, 0, 0.0])
    if wgs(r) >= 128:
        a[0] += 1; a[1] += (e - s) / 1e3
    else:
        a[2] += 1; a[3] += (e - s) / 1e3
print('# window_start_us  big_launches big_us  small_launches small_us')
for w in sorted(acc):
    a = acc[w]
    print('%8.0f %6d %8.1f %6d %8.1f' % (w * win, a[0], a[1], a[2], a[3]))
