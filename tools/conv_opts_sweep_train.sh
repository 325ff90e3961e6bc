#!/bin/bash
# The config-3 training step under variations of the convolution launch heuristics (UPF_CONV_OPTS -> upf_conv_set_option; they were
# tuned on the inference shapes of config 2): one bench process per setting, one box.    bash tools/conv_opts_sweep_train.sh
for o in "" "sk_grid=24" "sk_grid=96" "sk_grid=160" "sk_grid_narrow=48" "sk_grid_narrow=192" "small_grid=128" "small_grid=512" "rpw4_min=128" "rpw4_min=512" "sk_grid_d4=8" "sk_grid_d4=48" "force_sk=0" ""; do
  if [ -n "$o" ]; then export UPF_CONV_OPTS=$o; else unset UPF_CONV_OPTS; fi
  python bench.py --mode train --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-24s ms_per_step %s' % ('${o:-(defaults)}', d['ms_per_step']))"
done
