# scratch GPU job of the current iteration (edited per run)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r3e
timeout -k 10 300 python -m pytest tests/test_1_b8_gpu.py -x -q -k "per_rank or training_sequence" > gpurun_out/r3e/t.log 2>&1; tail -3 gpurun_out/r3e/t.log
for i in 1 2; do
  JVAE_HIP_LIB=$R/joint-vae_amd/jvae_hip/libjvae_base.so python tools/x3_time.py > gpurun_out/r3e/x3_base$i.log 2>&1
  python tools/x3_time.py > gpurun_out/r3e/x3_new$i.log 2>&1
done
cat gpurun_out/r3e/x3_base2.log gpurun_out/r3e/x3_new2.log | cut -c1-120
bash tools/ab_step.sh "noxcd JVAE_HIP_LIB=$R/joint-vae_amd/jvae_hip/libjvae_base.so" "xcd A=1" 2>&1 | tee gpurun_out/r3e/ab.log
bash tools/prof_kernel.sh r3e_fwd conv5_x3_kernel tools/dominant_kernel.py
cd $R
python3 -c "
import json; d=json.load(open('gpurun_out/r3e_fwd/summary.json')); print({k:v for k,v in d.items() if k not in ('passes','counters','kernel_filter')})"
bash tools/prof_trace.sh r3e_trace
cd $R; tail -5 gpurun_out/r3e_trace/step_trace.txt
