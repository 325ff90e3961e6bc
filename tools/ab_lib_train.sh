#!/bin/bash
# A/B of two BUILDS of the library on one box, TRAINING step (config 3): the default build against
# upflow_pytorch_amd/libupflow_hip_alt.so (UPF_HIP_LIB), bench.py --mode train in separate processes, alternated.
#   bash tools/ab_lib_train.sh [bench args]
ROUNDS=${ROUNDS:-3}
ALT=$(pwd)/upflow_pytorch_amd/libupflow_hip_alt.so
for r in $(seq $ROUNDS); do
  for v in default alt; do
    if [ $v = alt ]; then export UPF_HIP_LIB=$ALT; else unset UPF_HIP_LIB; fi
    python bench.py --mode train --steps 50 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('round $r  %-8s' % '$v', 'ms_per_step', d['ms_per_step'], 'loss', d['final_loss']['loss'])"
  done
done
