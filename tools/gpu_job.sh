# scratch GPU job of the current iteration (edited per run)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r3k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_0_ops_gpu.py -x -q -k "conv or gemm or linear or dense" > $O/t.log 2>&1; rc=$?; tail -3 $O/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_2_model_gpu.py -x -q > $O/t2.log 2>&1; rc=$?; tail -3 $O/t2.log
[ $rc -ne 0 ] && exit $rc
bash tools/ab_step.sh "unfold JVAE_DENSE_WINDOW=0" "dense JVAE_DENSE_WINDOW=1" 2>&1 | tee $O/ab.log
bash tools/prof_trace.sh r3k_trace
cd $R; grep -n "gemm\|unfold\|fold\|split_planes\|dense\|pack_refresh" gpurun_out/r3k_trace/step_trace.txt | head -50; tail -1 gpurun_out/r3k_trace/step_trace.txt
