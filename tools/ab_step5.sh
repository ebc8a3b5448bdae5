#!/bin/bash
# GPU box: A/B of the bf16 config-5 step on ONE box (see ab_step.sh)
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%% *}; envs=${spec#* }
    out=$(env $envs python bench.py --workload 5 --dtype bf16 --no-cpu-baseline --steps 30 --warmup 6 2>/dev/null)
    echo "$label rep$rep $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('wall %.3f median %.3f min %.3f' % (d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min']))" "$out")"
  done
done
