#!/bin/bash
# rocprofv3 passes for the cost volume at one shape (run on the GPU box):  tools/prof_corr_run.sh B C H W dtype tag [plain|norm] [round]
# kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own --pmc passes (never combined with other trace domains).
set -e
B=$1; C=$2; H=$3; W=$4; DT=$5; TAG=$6; VAR=${7:-plain}; RND=${8:-r03}
R=$(pwd)
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/tools/prof_corr.py $B $C $H $W $DT $VAR > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $R/tools/prof_corr.py $B $C $H $W $DT $VAR > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $R/tools/prof_corr.py $B $C $H $W $DT $VAR > $OUT/write.log 2>&1
cd $R
mkdir -p gpurun_out/profiles
python tools/prof_corr_summary.py $OUT/trace $OUT/fetch $OUT/write $B $C $H $W $DT gpurun_out/profiles/${RND}_corr81_${TAG}
