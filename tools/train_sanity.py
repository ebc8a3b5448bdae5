"""GPU box: a few hundred optimiser steps on a small synthetic 'dataset' through train_model(): loss must fall, no NaN."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'joint-vae_amd')]
import bench
net = bench.build_model(torch.device('cuda', 0))
g = torch.Generator().manual_seed(0)
# 10 class prototypes + noise, so that there is something to learn
protos = torch.rand(10, 3, 32, 32, generator=g)
y = torch.randint(0, 10, (4096,), generator=g)
x = (0.7 * protos[y] + 0.3 * torch.rand(4096, 3, 32, 32, generator=g)).clamp(0, 1)
ds = torch.utils.data.TensorDataset(x, y)
t0 = time.time()
hist = net.train_model(ds, epochs=6, batch_size=512, warmup=[0, 2])
torch.cuda.synchronize()
for e in range(6):
    h = hist[e]
    print('epoch', e, 'total %.2f kl %.3f cross_x %.2f' % (h['train_loss']['total'], h['train_loss']['kl'], h['train_loss']['cross_x']),
          'rmse %.4f sigma %.4f' % (h['train_measures']['rmse'], h['train_measures']['sigma']))
print('time %.1fs' % (time.time() - t0), 'finite:', all(bool(torch.isfinite(p).all()) for p in net.parameters()))
