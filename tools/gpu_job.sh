# scratch GPU job of the current iteration (edited per run)
set -o pipefail
R=$GRAFT_REPO_ROOT; cd $R
O=gpurun_out/r3o; mkdir -p $O
JVAE_BENCH_NO_PROBES=1 python bench.py --no-cpu-baseline --steps 3000 --warmup 10 > $O/soak.json 2> $O/soak.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3o/soak.json')); print('soak', round(d['value']), round(d['ms_per_step'],3), round(d['ms_per_step_median'],3), round(d['ms_per_step_min'],3), d['final_loss'])
PY
python - <<'PY'
import sys, os, torch, time
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),'joint-vae_amd')]
import bench
net = bench.build_model(torch.device('cuda',0), 2)
x = torch.rand(512,3,32,32,device='cuda'); y = torch.randint(0,10,(512,),device='cuda')
m=None
for i in range(50): _, m = net.train_step(x,y,batch=i,current_measures=m)
torch.cuda.synchronize(); a0=torch.cuda.memory_allocated(); r0=torch.cuda.memory_reserved()
for i in range(1500): _, m = net.train_step(x,y,batch=i,current_measures=m)
torch.cuda.synchronize(); a1=torch.cuda.memory_allocated(); r1=torch.cuda.memory_reserved()
print('memory allocated MB', a0>>20, a1>>20, 'reserved MB', r0>>20, r1>>20, 'rmse', m['rmse'])
PY
