# GPU box: launch-by-launch trace of ONE step of the real (two-stream) bench with the idle gap in front of every dispatch.
# usage: prof_gaps.sh TAG [bench args]   -> gpurun_out/TAG/step_trace_two_stream.txt
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
export JVAE_BENCH_NO_PROBES=1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 3 "$@" > $O/bench.json 2> $O/bench.err
python3 $R/tools/trace_step.py $(find $O -name "*_kernel_trace.csv") > $O/step_trace_two_stream.txt
find $O -name "*.csv" -delete
