# rocprofv3 kernel trace of a few eager (ungraphed) config-2 forwards -> per-kernel table of one steady-state forward (tools/steady_profile.py)
R=$(pwd); export TMPDIR=/tmp
P=$R/gpurun_out/profiles; mkdir -p $P
rm -rf $R/gpurun_out/prof_eager; mkdir -p $R/gpurun_out/prof_eager
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_eager -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-probe --no-graph > $R/gpurun_out/prof_eager/bench.log 2>&1)
(echo "# rocprofv3 --kernel-trace --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-probe --no-graph   (tools/steady_profile.py on the trace)"; python tools/steady_profile.py $(ls $R/gpurun_out/prof_eager/*/*kernel_trace.csv | head -1)) > $P/r03_bench_config2_eager_kernel_stats.txt
rm -rf $R/gpurun_out/prof_eager/*/
