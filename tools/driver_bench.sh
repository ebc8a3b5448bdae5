# GPU box: bench.py exactly as the driver launches it at N=1 (--gpus 1 --steps 20 --warmup 5), REPS fresh processes; prints the
# per-step lists in launch order (GPU events and host enqueue) so that a slow step can be located.  usage: driver_bench.sh TAG [REPS] [ENV=VAL ...]
R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-driver}; REPS=${2:-3}; shift 2 2>/dev/null
for kv in "$@"; do export "$kv"; done
O=gpurun_out/$TAG; mkdir -p $O
for rep in $(seq 1 $REPS); do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$rep.json 2> $O/bench_$rep.err || { tail -5 $O/bench_$rep.err; exit 1; }
  python - $O/bench_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('wall %.3f median %.3f min %.3f max %.3f drain %.2f' % (d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'], d['ms_per_step_max'], d['drain_ms_after_last_enqueue']))
print(' gpu :', ' '.join('%.2f' % t for t in d['per_step_ms']))
print(' host:', ' '.join('%.2f' % t for t in d['per_step_host_enqueue_ms']))
PY
done
