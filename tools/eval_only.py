"""eval_model.py loop only (batch 8, 4 levels x 4 noises), for profiling: python tools/eval_only.py [reps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from semantic_pyramid_for_image_generation_amd import eval_model, train_model as tm
dev = torch.device('cuda:0')
G, D, C = bench.build_models(dev, 'res', 'instance_norm')
G.eval()
norm_clf = tm.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
src = torch.rand(8, 3, 256, 256, generator=torch.Generator().manual_seed(1234)).to(dev)
for _ in range(2):
    eval_model.synthesize_batch(G, C, src, norm_clf)
torch.cuda.synchronize()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
t0 = time.perf_counter()
for _ in range(reps):
    eval_model.synthesize_batch(G, C, src, norm_clf)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('eval: %.1f images/s (%.2f ms per source batch of 8 -> 128 images)' % (reps * 128 / dt, dt / reps * 1e3))
