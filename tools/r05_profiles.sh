# GPU box, round-5 evidence: suite + smoke + the driver's bench line, kernel stats of the default (graph replay) bench and of the
# no-overlap eager loop, one-step launch traces, PMC passes of the dominant forward kernel, of the weight-gradient kernel and of the
# 4-phase kernel, the diagnostic bench lines of the other configs.   usage: gpurun -- 'bash tools/r05_profiles.sh [a|b]'
set -e -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r05; mkdir -p $O
part=${1:-a}
if [ $part = a ]; then
  bash tools/gpu_job.sh
  tail -3 gpurun_out/check/all.log > $O/r05_gpu_suite.txt
  tail -1 gpurun_out/check/smoke.log >> $O/r05_gpu_suite.txt
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_n1.json 2> $O/bench_n1.err
  python bench.py --gpus 1 --steps 20 --warmup 5 --eager --no-cpu-baseline > $O/r05_bench_eager.json 2>/dev/null
  bash tools/prof_bench.sh r05_bench > $O/r05_bench_step_breakdown.txt 2>&1
  cp $(find gpurun_out/r05_bench -name "*kernel_stats.csv") $O/r05_bench_kernel_stats.csv
  bash tools/prof_noov.sh r05_noov > $O/r05_bench_no_overlap_step_breakdown.txt 2>&1
  cp $(find gpurun_out/r05_noov -name "*kernel_stats.csv") $O/r05_bench_no_overlap_kernel_stats.csv
  bash tools/prof_trace.sh r05_trace
  cp gpurun_out/r05_trace/step_trace.txt $O/r05_step_trace_no_overlap.txt
  bash tools/prof_gaps.sh r05_gaps
  cp gpurun_out/r05_gaps/step_trace_two_stream.txt $O/r05_step_trace_graph_replay.txt
else
  bash tools/prof_kernel.sh r05_x3_fwd conv5_x3_kernel tools/dominant_kernel.py > $O/prof_x3.log 2>&1
  cp gpurun_out/r05_x3_fwd/summary.json $O/r05_x3_fwd_pmc.json
  cp $(find gpurun_out/r05_x3_fwd/trace -name "*kernel_stats.csv") $O/r05_x3_fwd_kernel_stats.csv
  bash tools/prof_kernel.sh r05_wgrad_x3 conv5_wgrad_x3_kernel tools/wgrad_probe.py > $O/prof_wg.log 2>&1
  cp gpurun_out/r05_wgrad_x3/summary.json $O/r05_wgrad_x3_pmc.json
  cp $(find gpurun_out/r05_wgrad_x3/trace -name "*kernel_stats.csv") $O/r05_wgrad_x3_kernel_stats.csv
  bash tools/prof_kernel.sh r05_t2s convt2s_x3_kernel tools/t2_probe.py > $O/prof_t2s.log 2>&1
  cp gpurun_out/r05_t2s/summary.json $O/r05_t2_x3_pmc.json
  python bench.py --workload 3 --no-cpu-baseline > $O/r05_bench_cfg3.json 2>/dev/null
  python bench.py --workload 5 --dtype bf16 --no-cpu-baseline > $O/r05_bench_cfg5_bf16.json 2>/dev/null
  python bench.py --workload 5 --dtype f32 --no-cpu-baseline > $O/r05_bench_cfg5_f32.json 2>/dev/null
  python bench.py --workload eval --steps 20 --warmup 3 --no-cpu-baseline > $O/r05_bench_eval.json 2>/dev/null
fi
