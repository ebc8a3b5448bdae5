# GPU box, round 5 job 1: the new 4-phase kernel (tests + A/B), host-time profile, graph vs eager at the driver's arguments,
# the train_model tests.
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05_job1; mkdir -p $O
export JVAE_KEEP_JOB_DIR=$R/gpurun_out/r05_jobs
timeout -k 10 120 python tools/t2_probe.py > $O/t2_new.txt 2>&1 || { tail -20 $O/t2_new.txt; exit 1; }
cat $O/t2_new.txt
JVAE_T2_V1=1 timeout -k 10 120 python tools/t2_probe.py > $O/t2_v1.txt 2>&1; grep us $O/t2_v1.txt
timeout -k 10 120 python tools/t2_probe.py | grep us
JVAE_T2_V1=1 timeout -k 10 120 python tools/t2_probe.py | grep us
timeout -k 10 600 python -m pytest tests/test_0_ops_gpu.py -x -q -k "stride2_transposed or conv_all_directions or deferred_batchnorm or deterministic" > $O/ops.log 2>&1; tail -3 $O/ops.log
timeout -k 10 600 python -m pytest tests/test_2_model_gpu.py -x -q -k "train_model or golden" > $O/model.log 2>&1; tail -5 $O/model.log
timeout -k 10 200 python tools/host_profile.py > $O/host_profile.txt 2>&1; head -45 $O/host_profile.txt
timeout -k 10 200 python tools/host_time.py 2>&1 | tail -3
for rep in 1 2; do
  for mode in "" "--graph"; do
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $mode > $O/bench_${rep}_${mode:-eager}.json 2>$O/bench.err || tail -5 $O/bench.err
    python - $O/bench_${rep}_${mode:-eager}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d['config']['launch'], 'wall %.3f median %.3f min %.3f max %.3f' % (d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'], d['ms_per_step_max']), 'host', ' '.join('%.2f' % t for t in d['per_step_host_enqueue_ms'][:8]))
PY
  done
done
