#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of tools/prof_sampling.py: per kernel, the mean of the last launches of every counter.
python tools/pmc_sampling_summary.py <dir> [<dir> ...]"""
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void upf::', '')[:48]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(d + '/**/*_kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r['Kernel_Name'].split('(')[0].replace('void upf::', '')[:48]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(agg):
    if not any(s in k for s in ('warp_fwd', 'blend_fwd', 'occ_check')):
        continue
    m = {c: sum(v[-4:]) / len(v[-4:]) for c, v in agg[k].items()}
    t = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else float('nan')
    print('%s   median %.1f us' % (k, t))
    cyc = m.get('GRBM_GUI_ACTIVE', 0) / 8
    wc = m.get('SQ_WAVE_CYCLES', 0)
    if cyc and wc:
        print('   clock %.2f GHz | of wave time: active %.1f%% (VALU %.1f%%, VMEM %.1f%%, SALU/other rest)  issue-stall %.1f%%  parked on waitcnt %.1f%%' % (
            cyc / t / 1e3, 100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc, 100 * m.get('SQ_ACTIVE_INST_VALU', 0) / wc, 100 * m.get('SQ_ACTIVE_INST_VMEM', 0) / wc,
            100 * m.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * m.get('SQ_WAIT_ANY', 0) / wc))
    if 'SQ_INSTS_VALU' in m:
        w = m.get('SQ_WAVES', 1)
        print('   per wave: %.0f VALU, %.0f SALU, %.0f VMEM read, %.0f VMEM write instructions; VALU busy %.1f%% of CU cycles' % (
            m['SQ_INSTS_VALU'] / w, m.get('SQ_INSTS_SALU', 0) / w, m.get('SQ_INSTS_VMEM_RD', 0) / w, m.get('SQ_INSTS_VMEM_WR', 0) / w,
            100 * m.get('SQ_ACTIVE_INST_VALU', 0) / 4 / (m.get('SQ_BUSY_CU_CYCLES', 1) or 1)))
    if 'TCC_HIT_sum' in m:
        print('   L2: hit %.0f miss %.0f (%.1f%% hits); FETCH_SIZE x2 = %.1f MB, WRITE_SIZE %.1f MB' % (
            m['TCC_HIT_sum'], m['TCC_MISS_sum'], 100 * m['TCC_HIT_sum'] / max(m['TCC_HIT_sum'] + m['TCC_MISS_sum'], 1), m.get('FETCH_SIZE', 0) * 2 / 1024, m.get('WRITE_SIZE', 0) / 1024))
    if 'TA_BUSY_avr' in m or 'TA_BUSY_sum' in m:
        print('   TA busy: %s' % {c: round(v, 1) for c, v in m.items() if c.startswith('TA_')})
