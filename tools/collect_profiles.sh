#!/bin/bash
# Everything under profiles/${ROUND}_* in one go (run on the GPU box; results land in gpurun_out/profiles/).
R=$(pwd); export TMPDIR=/tmp
P=$R/gpurun_out/profiles; mkdir -p $P
# 1. the default bench command under rocprofv3 (kernel trace + stats), and the same bench line un-profiled
python bench.py > $P/r04_bench_config2_default.json 2> /dev/null
for st in 1 2 3 4; do python bench.py --streams $st --no-cpu-baseline --no-train-probe --no-literal-split 2>/dev/null | tail -1 > $P/r04_bench_config2_streams$st.json; done
rm -rf $R/gpurun_out/prof_bench; mkdir -p $R/gpurun_out/prof_bench
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench/bench.log 2>&1)
python - <<PY
import csv, glob
f = glob.glob('$R/gpurun_out/prof_bench/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
out = open('$P/r04_bench_default_rocprof_stats.txt', 'w')
out.write('# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline   (MI355X; config 2, hipGraph replay with 4 steps in flight on 4 HIP streams, 10 warm-up + 50 timed steps + 50 steps one at a time,\n# then the roofline probes: 220 + 30 launches of the cost volume at [8,32,96,320], 55 of the 565->128 convolution)\n')
out.write('# bench line of this run: ' + [l for l in open('$R/gpurun_out/prof_bench/bench.log').read().split('\\n') if l.startswith('{')][-1] + '\\n')
out.write('%-150s %8s %14s %12s %8s\\n' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'pct'))
for r in rows[:45]:
    out.write('%-150s %8s %14s %12s %8s\\n' % (r['Name'][:150], r['Calls'], r['TotalDurationNs'], r['AverageNs'].split('.')[0], r['Percentage']))
PY
# 2. one steady-state forward, eager (per-kernel table)
rm -rf $R/gpurun_out/prof_eager; mkdir -p $R/gpurun_out/prof_eager
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_eager -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-probe --no-graph > $R/gpurun_out/prof_eager/bench.log 2>&1)
(echo "# rocprofv3 --kernel-trace --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-probe --no-graph   (tools/steady_profile.py on the trace)"; python tools/steady_profile.py $(ls $R/gpurun_out/prof_eager/*/*kernel_trace.csv | head -1)) > $P/r04_bench_config2_eager_kernel_stats.txt
# 2b. the same for the fp32 parity mode (split-precision convolutions)
rm -rf $R/gpurun_out/prof_eager32; mkdir -p $R/gpurun_out/prof_eager32
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_eager32 -- python $R/bench.py --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-train-probe --no-graph > $R/gpurun_out/prof_eager32/bench.log 2>&1)
(echo "# rocprofv3 --kernel-trace --output-format csv -- python bench.py --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-train-probe --no-graph   (tools/steady_profile.py on the trace)"; python tools/steady_profile.py $(ls $R/gpurun_out/prof_eager32/*/*kernel_trace.csv | head -1)) > $P/r04_bench_config2_fp32_eager_kernel_stats.txt
# 3. training steps
tools/train_profile.sh train_bf16 --no-graph > /dev/null 2>&1
tools/train_profile.sh train_fp32 --dtype fp32 --no-graph > /dev/null 2>&1
python bench.py --mode train > $P/r04_train_bf16_bench.json 2>/dev/null
python bench.py --mode train --dtype fp32 > $P/r04_train_fp32_bench.json 2>/dev/null
# 4. cost volume: rocprof + PMC of the normalising, octet-writing variant (the kernel inside the step) at the 1/4-resolution level of configs 2 and 5
if [ -z "$SKIP_CORR" ]; then
tools/prof_corr_run.sh 8 32 96 320 bf16 normc8_l4_cfg2_stacked_bf16 norm_c8 r04 > /dev/null 2>&1
tools/prof_corr_run.sh 2 32 240 720 bf16 normc8_l4_cfg5_stacked_bf16 norm_c8 r04 > /dev/null 2>&1
rm -rf $R/gpurun_out/prof_normc8_*
fi
# 4b. the other inference workloads of BASELINE.json (parity-test cases, not bench lines: recorded for DESIGN §6)
for wl in config4 config5 kitti_native; do python bench.py --workload $wl --no-cpu-baseline --no-train-probe 2>/dev/null | tail -1 > $P/r04_bench_$wl.json; done
# 4c. the fp32 parity mode under every convolution back end, and its bench lines
python tools/x3_modes.py > $P/r04_fp32_conv_modes_final.txt 2>/dev/null
for m in hip_x3 hip_x3s miopen; do python bench.py --dtype fp32 --fp32-conv $m --no-cpu-baseline --no-train-probe 2>/dev/null | tail -1 > $P/r04_bench_config2_fp32_$m.json; done
python bench.py --dtype fp16 --no-cpu-baseline --no-train-probe 2>/dev/null | tail -1 > $P/r04_bench_config2_fp16.json
# 5. every other operator at the level shapes
python tools/kbench.py > $P/r04_kbench.txt 2>&1
# 6. every convolution of one config-2 step, per layer (and the C8 variants of the layers the model runs in C8)
python tools/conv_layers.py > $P/r04_conv_layers.txt 2>&1
python tools/conv_layers.py --c8 > $P/r04_conv_layers_c8.txt 2>&1
# the raw traces are hundreds of MB: only the summaries travel back (gpurun merges <= 64 MiB)
rm -rf $R/gpurun_out/prof_bench/*/ $R/gpurun_out/prof_eager/*/ $R/gpurun_out/prof_eager32/*/ $R/gpurun_out/prof_train_bf16/trace $R/gpurun_out/prof_train_fp32/trace
du -sh $R/gpurun_out
ls -la $P
