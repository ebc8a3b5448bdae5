import os, sys, torch
REPO = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
from jvae_hip import ops
N,K,C=512,64,10
mu=torch.randn(N,K,device='cuda',requires_grad=True); lv=torch.randn(N,K,device='cuda',requires_grad=True)
eps=torch.randn(2,N,K,device='cuda'); eps[0]=0
y=torch.randint(0,C,(N,),device='cuda'); means=torch.randn(C,K,device='cuda',requires_grad=True); T=torch.ones(C,device='cuda')
def run():
    out=ops.latent(mu,lv,eps,y,means,T)
    (out[1].sum()+out[2].sum()).backward()
for _ in range(3): run()
torch.cuda.synchronize()
import time
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5): run()
    torch.cuda.synchronize()
for e in prof.key_averages():
    if 'latent' in e.key or 'means_grad' in e.key: print(e.key[:60], e.count, e.device_time_total/e.count)
