# GPU box: per-dispatch kernel trace of ONE training step (weight gradients on the main stream: every kernel alone), in launch order:
# name, grid, duration.  usage: prof_trace.sh TAG [bench args]   -> gpurun_out/TAG/step_trace.txt
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
export JVAE_BENCH_NO_PROBES=1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace -d $O/trace -- python3 $R/tools/bench_no_overlap.py --steps 6 --warmup 3 "$@" > $O/bench.json 2> $O/bench.err
python3 $R/tools/trace_step.py $(find $O -name "*_kernel_trace.csv") > $O/step_trace.txt
find $O -name "*.csv" -delete
