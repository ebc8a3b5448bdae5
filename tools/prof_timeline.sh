set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_tl; rm -rf $O; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 $R/tools/timeline.py $(find $O -name "*kernel_trace.csv") 10 14
find $O -name "*.csv" -delete
