#!/bin/bash
# PMC picture of wgrad_kernel for one layer shape: bash tools/prof_wgrad_run.sh "567 128 1"  (writes gpurun_out/prof_wgrad/*)
# (SQ counters only: a pass with TCC_*_sum counters did not terminate on this pool.)
R=$(pwd); export TMPDIR=/tmp
cfg=${1:-"567 128 1"}; tag=$(echo $cfg | tr ' ' '_')
OUT=$R/gpurun_out/prof_wgrad; mkdir -p $OUT
cd /tmp
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/a_$tag -- python $R/tools/prof_wgrad.py $cfg > $OUT/a_$tag.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/b_$tag -- python $R/tools/prof_wgrad.py $cfg > $OUT/b_$tag.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/prof_wgrad/*/*/*_counter_collection.csv')):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        if 'wgrad_' in r['Kernel_Name'] and 'reduce' not in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    kt = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f.replace('counter_collection', 'kernel_trace'))) if 'wgrad_' in r['Kernel_Name'] and 'reduce' not in r['Kernel_Name']]
    print(f.split('/')[2], 'dur us', [round(v, 1) for v in kt[-3:]], {k: round(sum(v[-2:]) / len(v[-2:])) for k, v in agg.items()})
PY
