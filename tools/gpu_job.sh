# scratch GPU job of the current iteration (edited per run)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r3p; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_0_ops_gpu.py -x -q -k "conv" > $O/t.log 2>&1; rc=$?; tail -3 $O/t.log
[ $rc -ne 0 ] && exit $rc
for v in 0 1; do JVAE_X3_S2=$v AFF=1 python tools/conv_bench.py 2>/dev/null | grep "^E1\|^E3\|^D2\|^D4" | sed "s/^/s2x3=$v /"; done
bash tools/ab_step.sh "f32s2 JVAE_X3_S2=0" "x3s2 JVAE_X3_S2=1" 2>&1 | tee $O/ab.log
