# GPU box: A/B of two builds of the library on ONE box: libjvae_base.so (built from another revision, git-ignored) against the
# tree's libjvae_hip.so - conv_bench lines of the given layers and the step, interleaved.  usage: ab_lib.sh "E2|D1|D3|D5"
R=$GRAFT_REPO_ROOT; cd $R
B=$R/joint-vae_amd/jvae_hip/libjvae_base.so
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export JVAE_HIP_LIB=$B; else unset JVAE_HIP_LIB; fi
    AFF=1 python tools/conv_bench.py 2>/dev/null | grep -E "^($1)" | sed "s/^/$lib /"
    out=$(python bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null)
    echo "$lib rep$rep $(python -c "import json,sys; d=json.loads(sys.argv[1]); print('wall %.3f median %.3f min %.3f dom %.1f us' % (d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'], 1e3*d['roofline']['launch_ms']))" "$out")"
  done
done
