# scratch GPU job of the current iteration (edited per run)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r3g
timeout -k 10 600 python -m pytest tests/test_0_ops_gpu.py -x -q -k "conv or gemm or linear" > gpurun_out/r3g/t.log 2>&1; rc=$?; tail -3 gpurun_out/r3g/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_2_model_gpu.py -x -q > gpurun_out/r3g/t2.log 2>&1; rc=$?; tail -3 gpurun_out/r3g/t2.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 40 --warmup 8 > gpurun_out/r3g/eager$i.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 40 --warmup 8 --graph > gpurun_out/r3g/graph$i.json 2>/dev/null
done
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --workload eval > gpurun_out/r3g/eval.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3g/*.json')):
    d=json.load(open(f)); print(f, round(d['value']), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), round(d['ms_per_step_min'],3))
PY
