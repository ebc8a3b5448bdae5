# GPU box: A/B of the BatchNorm-backward traversal orders (JVAE_BN_ORDER) - the stand-alone probe and the whole step
cd $GRAFT_REPO_ROOT
for rnd in 1 2; do
for o in 0 1 2 3; do
  JVAE_BN_ORDER=$o python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/bn_order_$o.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/bn_order_$o.json')); print('order $o round $rnd: step', round(d['ms_per_step'],3), 'median', round(d['ms_per_step_median'],3), 'bn alone us', round(d['roofline_hbm']['launch_ms']*1e3,1))
PY
done
done
