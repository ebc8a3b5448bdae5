# scratch GPU job of the current iteration (edited per run)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r3h; mkdir -p $O
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.json; echo
python bench.py --no-cpu-baseline --workload 3 > $O/bench_cfg3.json 2>/dev/null
python bench.py --no-cpu-baseline --workload 5 --dtype bf16 > $O/bench_cfg5_bf16.json 2>/dev/null
python bench.py --no-cpu-baseline --workload 5 > $O/bench_cfg5_f32.json 2>/dev/null
python bench.py --no-cpu-baseline --workload eval --steps 10 --warmup 3 > $O/bench_eval.json 2>/dev/null
python bench.py --no-cpu-baseline --graph > $O/bench_graph.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3h/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value']), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3))
    except Exception as e: print(f, 'ERR', e)
PY
bash tools/prof_bench.sh r3h_prof > $O/step_breakdown.txt 2>&1; cd $R
bash tools/prof_noov.sh r3h_noov > $O/step_breakdown_no_overlap.txt 2>&1; cd $R
cp $(find gpurun_out/r3h_prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
cp $(find gpurun_out/r3h_noov -name "*kernel_stats.csv" | head -1) $O/bench_no_overlap_kernel_stats.csv
head -16 $O/step_breakdown_no_overlap.txt
