# scratch GPU job of the current iteration (edited per run)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
bash tools/prof_trace.sh r3m_trace5 --workload 5 --dtype bf16
cd $R; tail -1 gpurun_out/r3m_trace5/step_trace.txt
