# A/B on one box, alternated: bench.py with every tensor bf16 against bf16 + fp16 pyramid (the decision of DESIGN 4.4: throughput and epe_vs_reference of both)
for i in 1 2; do
python bench.py --no-cpu-baseline --no-train-probe --no-literal-split > gpurun_out/ab_pyr_bf16_$i.json 2>/dev/null
python bench.py --no-cpu-baseline --no-train-probe --no-literal-split --pyramid-dtype fp16 > gpurun_out/ab_pyr_fp16_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_pyr_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['value_min'], d['value_max'], d['one_step_in_flight']['ms_per_step'], d.get('epe_vs_reference'))
PY
