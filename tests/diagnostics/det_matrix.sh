#!/bin/bash
# GPU box: run tools/determinism.py under a few switches (step-0 gradient difference of two identical models)
cd "$(dirname "$0")/../.."
for cfg in "A_default:" "B_no_defer:JVAE_DEFER_BN=0" "A_default_again:"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo -n "$name  "
  env $envs timeout -k 10 100 python tests/diagnostics/determinism.py 2>&1 | grep "step 0 worst grad" | cut -c1-90
done
