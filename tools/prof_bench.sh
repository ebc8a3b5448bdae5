# GPU box: rocprofv3 kernel trace of the default bench.py run; keeps kernel_stats.csv + the bench line.  usage: prof_bench.sh TAG [bench args]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
export JVAE_BENCH_NO_PROBES=1      # the trace holds the training steps only (no roofline probe launches)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $O/bench.json 2> $O/bench.err
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*_agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete
python3 $R/tools/step_stats.py $(find $O -name "*kernel_stats.csv")
