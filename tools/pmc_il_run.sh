cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
for cfg in "565 128 96 320 1" "531 32 96 320 1" "128 128 96 320 2"; do
  tag=$(echo $cfg | tr ' ' '_')
  for pp in 0 1; do
    UPF_CONV_OPTS=il=$pp rocprofv3 --kernel-trace --pmc $PMC -d $R/gpurun_out/r3g/pmc_il${pp}_$tag --output-format csv -- python $R/tools/prof_conv.py $cfg 8 6 > /dev/null 2>&1
    echo "== $cfg il=$pp"; python $R/tools/pmc_summary.py $(dirname $(find $R/gpurun_out/r3g/pmc_il${pp}_$tag -name '*counter_collection.csv' | head -1))
  done
done
for pp in 0 1; do
  UPF_ZERO_X=1 UPF_CONV_OPTS=il=$pp rocprofv3 --kernel-trace --pmc $PMC -d $R/gpurun_out/r3g/pmc_zero_il${pp} --output-format csv -- python $R/tools/prof_conv.py 565 128 96 320 1 8 6 > /dev/null 2>&1
  echo "== 565 128 96 320 1 ZERO INPUT il=$pp"; python $R/tools/pmc_summary.py $(dirname $(find $R/gpurun_out/r3g/pmc_zero_il${pp} -name '*counter_collection.csv' | head -1))
done
