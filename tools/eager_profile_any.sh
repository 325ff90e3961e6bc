#!/bin/bash
# rocprofv3 kernel trace of a few eager (ungraphed) forwards of any bench workload -> per-kernel table of one steady-state forward
#   bash tools/eager_profile_any.sh <tag> <bench args...>     e.g.  bash tools/eager_profile_any.sh kitti_native_b4 --workload kitti_native
R=$(pwd); export TMPDIR=/tmp
TAG=$1; shift
P=$R/gpurun_out/profiles; mkdir -p $P
D=$R/gpurun_out/prof_eager_$TAG
rm -rf $D; mkdir -p $D
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $D -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-probe --no-literal-split --no-eval-probe --no-graph "$@" > $D/bench.log 2>&1)
(echo "# rocprofv3 --kernel-trace --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-probe --no-literal-split --no-graph $*   (tools/steady_profile.py on the trace)"; python tools/steady_profile.py $(ls $D/*/*kernel_trace.csv | head -1)) > $P/${ROUND:-r05}_bench_${TAG}_eager_kernel_stats.txt
tail -2 $D/bench.log
rm -rf $D/*/
