"""Per-iteration host issue time vs GPU completion time of the eager loop (is the GPU ever waiting for the host?)"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from argparse import Namespace
from semantic_pyramid_for_image_generation_amd import loop, optim, train_model as tm
dev = torch.device('cuda:0')
G, D, C = bench.build_models(dev, 'res', 'instance_norm')
tm.args = Namespace(use_diversity_loss=False, gradient_penalty_weight=10.0)
g_opt = optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True)
d_opt = optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True)
gen = torch.Generator().manual_seed(1234)
pool = [torch.rand(16, 3, 256, 256, generator=gen).to(dev) for _ in range(5)]
torch.manual_seed(100); random.seed(0)
lp = loop.TrainLoop(G, D, C, g_opt, d_opt, 16, dev, use_graphs=False)
lp.prepare()
for i in range(15):
    lp.iteration(pool[i % 5])
torch.cuda.synchronize()
N = 30
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
ev[0].record()
t0 = time.perf_counter()
for i in range(N):
    lp.iteration(pool[i % 5])
    ev[i + 1].record()
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
gpu = [ev[0].elapsed_time(ev[i + 1]) for i in range(N)]
print("iter  do_g  host_end  gpu_end  lead   d_host  d_gpu")
for i in range(N):
    it = lp.iterations - N + i + 1
    print(f"{i:3d}   {int(it % 5 == 1)}   {host[i]:8.2f} {gpu[i]:8.2f} {gpu[i]-host[i]:6.2f}  {host[i]-(host[i-1] if i else 0):6.2f} {gpu[i]-(gpu[i-1] if i else 0):6.2f}")
