# PMC passes (wait / active / instruction-mix counters) over the sampling operators at the config-2 shapes (tools/prof_sampling.py), summarised by tools/pmc_sampling_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_sampling
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_WAVES -d $O/p1 --output-format csv -- python $R/tools/prof_sampling.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU -d $O/p2 --output-format csv -- python $R/tools/prof_sampling.py > /dev/null 2>&1
python $R/tools/pmc_sampling_summary.py $O/p1; python $R/tools/pmc_sampling_summary.py $O/p2
