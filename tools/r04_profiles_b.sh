# GPU box, round-4 evidence, part B: PMC passes of the dominant forward kernel and of the weight-gradient kernel, the s_memtime
# anatomy (diagnostic build), the diagnostic bench lines of the other configs.
set -e -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/r04
bash tools/prof_kernel.sh r04_x3_fwd conv5_x3_kernel tools/dominant_kernel.py > gpurun_out/r04/prof_x3.log 2>&1
cp gpurun_out/r04_x3_fwd/summary.json gpurun_out/r04/r04_x3_fwd_pmc.json
cp $(find gpurun_out/r04_x3_fwd/trace -name "*kernel_stats.csv") gpurun_out/r04/r04_x3_fwd_kernel_stats.csv
bash tools/prof_kernel.sh r04_wgrad_x3 conv5_wgrad_x3_kernel tools/wgrad_probe.py > gpurun_out/r04/prof_wg.log 2>&1
cp gpurun_out/r04_wgrad_x3/summary.json gpurun_out/r04/r04_wgrad_x3_pmc.json
cp $(find gpurun_out/r04_wgrad_x3/trace -name "*kernel_stats.csv") gpurun_out/r04/r04_wgrad_x3_kernel_stats.csv
JVAE_X3_TPW=1 JVAE_HIP_LIB=$R/joint-vae_amd/jvae_hip/libjvae_stamps.so python tools/x3_stamps.py > gpurun_out/r04/r04_x3_stamps_raw.txt 2>&1
python bench.py --workload 3 --no-cpu-baseline > gpurun_out/r04/r04_bench_cfg3.json 2>/dev/null
python bench.py --workload 5 --dtype bf16 --no-cpu-baseline > gpurun_out/r04/r04_bench_cfg5_bf16.json 2>/dev/null
python bench.py --workload 5 --dtype f32 --no-cpu-baseline > gpurun_out/r04/r04_bench_cfg5_f32.json 2>/dev/null
python bench.py --workload eval --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r04/r04_bench_eval.json 2>/dev/null
python bench.py --graph --no-cpu-baseline > gpurun_out/r04/r04_bench_graph.json 2>/dev/null
