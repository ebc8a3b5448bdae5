# scratch GPU job of the current iteration (edited per run)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r3l; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/all.log 2>&1; rc=$?; tail -3 $O/all.log
exit $rc
