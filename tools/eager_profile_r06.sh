# rocprofv3 kernel trace of a few eager (ungraphed) config-2 forwards -> per-kernel table of one steady-state forward (tools/steady_profile.py) and its ordered timeline (tools/step_timeline.py): profiles/r06_bench_config2_b4_eager_kernel_stats.txt, r06_step_timeline_full.txt
R=$(pwd); export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_eager; mkdir -p $R/gpurun_out/prof_eager $R/gpurun_out/profiles
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_eager -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-probe --no-literal-split --no-graph > $R/gpurun_out/prof_eager/bench.log 2>&1)
T=$(ls $R/gpurun_out/prof_eager/*/*kernel_trace.csv | head -1)
(echo "# rocprofv3 --kernel-trace --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-probe --no-literal-split --no-graph   (tools/steady_profile.py on the trace)"; python tools/steady_profile.py $T) > $R/gpurun_out/profiles/r06_bench_config2_b4_eager_kernel_stats.txt
python tools/step_timeline.py $T --full > $R/gpurun_out/profiles/r06_step_timeline_full.txt
rm -rf $R/gpurun_out/prof_eager/*/
