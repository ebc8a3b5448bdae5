#!/bin/bash
# GPU box: A/B of the training step on ONE box.  Each argument is "label ENV=VAL ENV=VAL ..."; every variant runs
# `bench.py --no-cpu-baseline` twice, interleaved, and the wall / median / min step times are printed.
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%% *}; envs=${spec#* }
    out=$(env $envs python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null)
    echo "$label rep$rep $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('wall %.3f median %.3f min %.3f dom %.1f us' % (d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'], 1e3*d['roofline']['launch_ms']))" "$out")"
  done
done
