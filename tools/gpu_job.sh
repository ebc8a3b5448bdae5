# scratch GPU job of the current iteration (edited per run)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r3s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_0_ops_gpu.py -x -q -k "conv or head" > $O/t.log 2>&1; rc=$?; tail -2 $O/t.log
[ $rc -ne 0 ] && exit $rc
bash tools/prof_trace.sh r3s_t; cd $R
grep "gemm\|unfold\|fold" gpurun_out/r3s_t/step_trace.txt | cut -c1-12,40-90 | head -30; tail -1 gpurun_out/r3s_t/step_trace.txt
