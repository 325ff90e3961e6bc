#!/bin/bash
# A/B of two BUILDS of the library on one box: the default build against upflow_pytorch_amd/libupflow_hip_alt.so (UPF_HIP_LIB), bench.py
# in separate processes, alternated.   bash tools/ab_lib.sh [bench args]
ROUNDS=${ROUNDS:-3}
ALT=$(pwd)/upflow_pytorch_amd/libupflow_hip_alt.so
for r in $(seq $ROUNDS); do
  for v in default alt; do
    if [ $v = alt ]; then export UPF_HIP_LIB=$ALT; else unset UPF_HIP_LIB; fi
    python bench.py --no-cpu-baseline --no-train-probe --no-literal-split --no-eval-probe "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('round $r  %-8s' % '$v', d['value'], d['value_min'], d['value_max'], 'one-in-flight ms', d['one_step_in_flight']['ms_per_step'])"
  done
done
