#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_* pass of tools/prof_conv.py: python tools/pmc_summary.py <dir>"""
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + '/*_counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        if 'conv_kernel' in r['Kernel_Name'] or 'conv_x3' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    kt = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f.replace('counter_collection', 'kernel_trace'))) if 'conv_kernel' in r['Kernel_Name'] or 'conv_x3' in r['Kernel_Name']]
    m = {k: sum(v[-3:]) / len(v[-3:]) for k, v in agg.items()}
    if 'SQ_WAVE_CYCLES' not in m:
        continue
    wc = m['SQ_WAVE_CYCLES']
    cyc = m['GRBM_GUI_ACTIVE'] / 8
    print(f.split('/')[-1], 'dur us', [round(v, 1) for v in kt[-3:]], 'clock %.2f GHz' % (cyc / kt[-1] / 1e3))
    print('   MFMA util %.1f%%  | of wave time: parked(waitcnt/barrier) %.1f%%  issue-stall %.1f%% (LDS %.1f%%)  active %.1f%% | LDS active %.1f%% of kernel cycles, conflicts %.0f%% of LDS active' % (
        100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc, 100 * m['SQ_WAIT_ANY'] / wc, 100 * m['SQ_WAIT_INST_ANY'] / wc, 100 * m['SQ_WAIT_INST_LDS'] / wc,
        100 * m['SQ_ACTIVE_INST_ANY'] / wc, 100 * m['SQ_LDS_IDX_ACTIVE'] / 256 / cyc, 100 * m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']))
