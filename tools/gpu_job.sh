# scratch GPU job of the current iteration (edited per run)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r3y; mkdir -p $O
bash tools/prof_trace.sh r3y_t; cd $R
grep "gemm" gpurun_out/r3y_t/step_trace.txt | cut -c1-12,40-90 | head -30; tail -1 gpurun_out/r3y_t/step_trace.txt
JVAE_GEMM_DEPTH=1 bash tools/prof_trace.sh r3y_t1; cd $R
grep "gemm" gpurun_out/r3y_t1/step_trace.txt | cut -c1-12,40-90 | head -30; tail -1 gpurun_out/r3y_t1/step_trace.txt
