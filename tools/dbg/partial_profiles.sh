R=$(pwd); export TMPDIR=/tmp
P=$R/gpurun_out/profiles; mkdir -p $P
tools/train_profile.sh train_bf16 --dtype bf16 --no-graph > /dev/null 2>&1
python bench.py --mode train --dtype bf16 > $P/r02_train_bf16_bench.json 2>/dev/null
python tools/kbench.py > $P/r02_kbench.txt 2>&1
