# GPU box: kernel-trace timelines of the graph-replayed and the eager training step (tools/timeline.py on both)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s4_graph; rm -rf $O; mkdir -p $O
for mode in ${MODES:-graph eager}; do
  export MODE=$mode
  rocprofv3 --output-format csv --kernel-trace -d $O/$mode -- python3 $R/tools/graph_probe.py > $O/$mode.log 2>&1
  echo "== $mode: $(tail -1 $O/$mode.log)"
  python3 $R/tools/timeline.py $(find $O/$mode -name "*kernel_trace.csv") 12 15
done
unset MODE; python3 $R/tools/graph_probe.py; MODE=eager python3 $R/tools/graph_probe.py
find $O -name "*.csv" -size +20M -delete
