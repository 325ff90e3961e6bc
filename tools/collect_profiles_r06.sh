#!/bin/bash
# The round-6 profile set in one go (run on the GPU box; results land in gpurun_out/profiles/r06_*).
R=$(pwd); export TMPDIR=/tmp; export ROUND=r06
P=$R/gpurun_out/profiles; mkdir -p $P
# 1. the default bench command, un-profiled (the driver's form) and under rocprofv3 (kernel trace + stats)
python bench.py > $P/r06_bench_config2_default.json 2> /dev/null
rm -rf $R/gpurun_out/prof_bench; mkdir -p $R/gpurun_out/prof_bench
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench/bench.log 2>&1)
python - <<PY
import csv, glob
f = glob.glob('$R/gpurun_out/prof_bench/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
out = open('$P/r06_bench_default_rocprof_stats.txt', 'w')
out.write('# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline   (MI355X; config 2, bf16 decoder + fp16 pyramid, hipGraph replay with 4 steps in flight on 4 HIP streams, 10 warm-up + 3 windows of 50 timed steps + 50 steps one at a time,\n# then the roofline probes: 220 + 30 launches of the cost volume at [8,32,96,320] (fp16 features -> bf16 octets), 55 of the 565->128 convolution, the EPE probe, the literal-split and training probes)\n')
out.write('# bench line of this run: ' + [l for l in open('$R/gpurun_out/prof_bench/bench.log').read().split('\\n') if l.startswith('{')][-1] + '\\n')
out.write('%-150s %8s %14s %12s %8s\\n' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'pct'))
for r in rows[:50]:
    out.write('%-150s %8s %14s %12s %8s\\n' % (r['Name'][:150], r['Calls'], r['TotalDurationNs'], r['AverageNs'].split('.')[0], r['Percentage']))
PY
# 2. one steady-state forward, eager (per-kernel tables + the ordered timeline): config 2 and the native KITTI frame
bash tools/eager_profile_r06.sh > /dev/null 2>&1
bash tools/eager_profile_any.sh kitti_native_b4 --workload kitti_native > /dev/null 2>&1
# 3. the pure-bf16 configuration (rounds 1-5's headline) and fp16, KITTI-native frames, the other workloads
python bench.py --pyramid-dtype bf16 --no-cpu-baseline --no-train-probe 2>/dev/null | tail -1 > $P/r06_bench_config2_pure_bf16.json
python bench.py --dtype fp16 --no-cpu-baseline --no-train-probe 2>/dev/null | tail -1 > $P/r06_bench_config2_fp16.json
python bench.py --workload kitti_native --no-cpu-baseline --no-train-probe --no-literal-split 2>/dev/null | tail -1 > $P/r06_bench_kitti_native.json
for wl in config4 config5; do python bench.py --workload $wl --no-cpu-baseline --no-train-probe 2>/dev/null | tail -1 > $P/r06_bench_$wl.json; done
# 4. cost volume: rocprof + PMC of the launch inside the step (fp16 features -> bf16 octets, normalising) at the 1/4-resolution level of config 2,
#    the single-type bf16 form beside it, config 5 and the native KITTI frame (row-pitched features, PADW kernel)
tools/prof_corr_run.sh 8 32 96 320 bf16 normc8_l4_cfg2_stacked_f16in_bf16 norm_c8_mixed r06 > /dev/null 2>&1
tools/prof_corr_run.sh 8 32 96 320 bf16 normc8_l4_cfg2_stacked_bf16 norm_c8 r06 > /dev/null 2>&1
tools/prof_corr_run.sh 8 32 94 311 bf16 normc8_l4_kitti_stacked_f16in_bf16 norm_c8_mixed r06 > /dev/null 2>&1
tools/prof_corr_run.sh 2 32 240 720 bf16 normc8_l4_cfg5_stacked_f16in_bf16 norm_c8_mixed r06 > /dev/null 2>&1
rm -rf $R/gpurun_out/prof_normc8_*
# 5. training step
tools/train_profile.sh train_bf16 --no-graph > /dev/null 2>&1
python bench.py --mode train > $P/r06_train_bf16_bench.json 2>/dev/null
# 6. per-layer / per-operator tables, the merged tails, the vendor-GEMM ceiling
python tools/conv_layers.py > $P/r06_conv_layers.txt 2>&1
python tools/conv_layers.py --c8 --pure > $P/r06_conv_layers_c8.txt 2>&1
python tools/tail_bench.py > $P/r06_tail_bench.txt 2>&1
python tools/kbench.py > $P/r06_kbench.txt 2>&1
python tools/gemm_ceiling.py > /dev/null 2>&1; cp gpurun_out/gemm_ceiling.txt $P/r06_gemm_ceiling.txt
(python tools/ab_bench.py "" "no_merge=1"; UPF_AB_STREAMS=4 python tools/ab_bench.py "" "no_merge=1") 2>/dev/null | grep -v amdgpu.ids > $P/r06_merged_tail_ab.txt
rm -rf $R/gpurun_out/prof_bench/*/ $R/gpurun_out/prof_eager*/*/ $R/gpurun_out/prof_train_bf16/trace
du -sh $R/gpurun_out
ls -la $P
