# scratch GPU job of the current iteration (edited per run)
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r3q
python tools/aten_ops.py > gpurun_out/r3q/aten.txt 2>&1; tail -40 gpurun_out/r3q/aten.txt
