#!/bin/bash
# rocprofv3 kernel trace of a few config-3 training steps -> per-kernel table of ONE steady-state step (tools/step_profile.py)
set -e
R=$(pwd); TAG=${1:-train}; shift || true
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --mode train --steps 4 --warmup 3 "$@" > $OUT/bench.log 2>&1
cd $R
mkdir -p gpurun_out/profiles
python tools/step_profile.py $(ls $OUT/trace/*/*_kernel_trace.csv | head -1) > gpurun_out/profiles/${ROUND:-r05}_${TAG}_step_kernels.txt
head -60 gpurun_out/profiles/${ROUND:-r05}_${TAG}_step_kernels.txt
