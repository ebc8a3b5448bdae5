# scratch GPU job of the current iteration (edited per run)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r3x; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_0_ops_gpu.py -x -q -k "gemm or linear or conv or head or point" > $O/t.log 2>&1; rc=$?; tail -2 $O/t.log
[ $rc -ne 0 ] && exit $rc
bash tools/prof_trace.sh r3x_t; cd $R
grep "gemm" gpurun_out/r3x_t/step_trace.txt | cut -c1-12,40-90 | head -30; tail -1 gpurun_out/r3x_t/step_trace.txt
bash tools/ab_step.sh "depth3 JVAE_GEMM_DEPTH=3" "db JVAE_GEMM_DEPTH=2" 2>&1 | tee $O/ab.log
