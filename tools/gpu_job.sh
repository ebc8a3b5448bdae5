# GPU box: the checks of a finished change, in the order the driver runs them (usage: gpurun -- 'bash tools/gpu_job.sh').
# Any failing stage (suite, smoke, bench) makes the job exit non-zero; the bench line read back is THIS run's.
set -e -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/check; mkdir -p $O
rm -f $O/bench_n1.json $O/bench_n1.err $O/smoke.log
rc=0
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/all.log 2>&1 || rc=$?
tail -3 $O/all.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { rc=$?; tail -5 $O/smoke.log; exit $rc; }
tail -1 $O/smoke.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { rc=$?; tail -5 $O/bench_n1.err; exit $rc; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/check/bench_n1.json')); print(round(d['value']), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), d['steps'], round(d['roofline']['frac'],3), round(d['roofline']['launch_ms']*1e3,1), round(d['cpu_baseline']['value'],1), d['cpu_baseline']['cores'])
PY
