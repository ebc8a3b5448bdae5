# GPU box, round-4 evidence, part A: suite + smoke + bench line, two-stream and no-overlap kernel stats, one-step launch trace.
set -e -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
bash tools/gpu_job.sh
mkdir -p gpurun_out/r04
cp gpurun_out/check/bench_n1.json gpurun_out/r04/r04_bench_n1.json
tail -3 gpurun_out/check/all.log > gpurun_out/r04/r04_gpu_suite.txt
cat gpurun_out/check/smoke.log | tail -1 >> gpurun_out/r04/r04_gpu_suite.txt
bash tools/prof_bench.sh r04_bench > gpurun_out/r04/r04_bench_step_breakdown.txt 2>&1
cp $(find gpurun_out/r04_bench -name "*kernel_stats.csv") gpurun_out/r04/r04_bench_kernel_stats.csv
bash tools/prof_noov.sh r04_noov > gpurun_out/r04/r04_bench_no_overlap_step_breakdown.txt 2>&1
cp $(find gpurun_out/r04_noov -name "*kernel_stats.csv") gpurun_out/r04/r04_bench_no_overlap_kernel_stats.csv
bash tools/prof_trace.sh r04_trace
cp gpurun_out/r04_trace/step_trace.txt gpurun_out/r04/r04_step_trace_no_overlap.txt
bash tools/prof_gaps.sh r04_gaps
cp gpurun_out/r04_gaps/step_trace_two_stream.txt gpurun_out/r04/r04_step_trace_two_stream.txt
