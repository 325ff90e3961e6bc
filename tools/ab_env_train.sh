#!/bin/bash
# A/B of one ENVIRONMENT switch on one box, TRAINING step (config 3): bench.py --mode train with and without `VAR=1`, separate
# processes, alternated.     VAR=UPF_NO_GATED_DGRAD bash tools/ab_env_train.sh [bench args]
ROUNDS=${ROUNDS:-3}
VAR=${VAR:?name of the switch}
for r in $(seq $ROUNDS); do
  for v in default "$VAR"; do
    if [ $v = default ]; then unset $VAR; else export $VAR=1; fi
    python bench.py --mode train --steps 50 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('round $r  %-22s' % '$v', 'ms_per_step', d['ms_per_step'], 'loss', d['final_loss']['loss'])"
  done
done
