"""One rank of the data-parallel product path (not a test module: tests/test_dp_gpu.py launches N of these under
torch.distributed.run).  The rank goes through exactly what `train_model._main` / `bench.py` do for WORLD_SIZE > 1:
parallel.init_device -> broadcast_module -> flatten_grads -> optim.Adam(grad_scale=1/N, zero_grads=True) -> TrainLoop
iterations (classifier taps, masks, critic step, generator step) with the gradient all-reduce inside the step functions.
It writes its inputs, the draws of every iteration and the resulting parameters to <out>/rank<r>.pt.

usage: dp_worker.py <out_dir> <iterations> <batch>
backend: SEMPYR_DIST_BACKEND (nccl = RCCL, one GPU per rank; gloo + SEMPYR_SHARE_GPU=1 = rehearsal of the same code with
both ranks on one card and host-staged collectives)."""
import os
import random
import sys
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

SEED = 41


def main():
    out_dir, iters, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    from oracle import recipes          # test infrastructure: seeded inputs / ResNet-18 weights shared with the parent's oracle
    from semantic_pyramid_for_image_generation_amd import loop, optim, parallel
    from semantic_pyramid_for_image_generation_amd import train_model as tm
    from semantic_pyramid_for_image_generation_amd.classifier import Classifier, resnet18
    from semantic_pyramid_for_image_generation_amd.discriminator import Discriminator
    from semantic_pyramid_for_image_generation_amd.generator import Generator
    dev = parallel.init_device()
    rank, world = parallel.rank(), parallel.world_size()
    assert world == int(os.environ['WORLD_SIZE']) and world > 1
    gen_type = os.environ.get('DP_GEN_TYPE', 'res')
    norm = os.environ.get('DP_NORM', 'instance_norm')
    torch.manual_seed(SEED)
    G = Generator(gen_type); G.init_weights()
    D = Discriminator(norm, gen_type); D.init_weights()
    C = Classifier(resnet18(365)); C.load_state_dict(recipes.make_resnet(SEED).state_dict())
    init = {'G': {k: v.detach().clone() for k, v in G.state_dict().items()},
            'D': {k: v.detach().clone() for k, v in D.state_dict().items()}}
    if rank != 0:                        # diverge on purpose: the broadcast must repair parameters and buffers
        with torch.no_grad():
            for p in list(G.parameters()) + list(D.parameters()):
                p.add_(0.25)
            for b in G.buffers():
                if b.is_floating_point():
                    b.add_(1.0)
    G.to(dev); D.to(dev); C.to(dev).eval()
    for p in C.parameters():
        p.requires_grad_(False)
    parallel.broadcast_module(G); parallel.broadcast_module(D); parallel.broadcast_module(C)
    tm.args = Namespace(use_diversity_loss=os.environ.get('DP_DIVERSITY', '0') == '1', gradient_penalty_weight=10.0)
    parallel.flatten_grads(G)
    parallel.flatten_grads(D, bucket_bytes=int(os.environ.get('DP_BUCKET_BYTES', str(8 << 20))))
    dp = dict(grad_scale=1.0 / world, zero_grads=True)
    g_opt = optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True, **dp)
    d_opt = optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True, **dp)
    lp = loop.TrainLoop(G, D, C, g_opt, d_opt, B, dev, use_graphs=False)

    draws = []
    orig = lp._draw_device

    def recording(key, images, static):
        inp = orig(key, images, static)
        draws.append({'key': key, 'L': lp.features_to_train, 'specs': [list(s) for s in lp.specs],
                      'z_d': inp.z_d.clone(), 'alpha': inp.alpha.clone(),
                      'z_g': None if inp.z_g is None else inp.z_g.clone(),
                      'z2': None if inp.z2 is None else inp.z2.clone()})
        return inp
    lp._draw_device = recording

    images = recipes.image_batch(SEED + 10 * rank, B)          # rank-distinct data
    torch.manual_seed(100 + rank)                               # rank-distinct z / alpha
    random.seed(0)                                              # identical layer / crop draws on every rank
    snaps = []
    x = images.to(dev)
    for _ in range(iters):
        out = lp.iteration(x, log=True)
        torch.cuda.synchronize()
        snaps.append({'G': {k: v.detach().cpu().clone() for k, v in G.state_dict().items()},
                      'D': {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}})
    logged = lp.pop_logged()
    flat_zero = float(D._sempyr_flat_grads.flat.abs().max()) == 0.0 and float(G._sempyr_flat_grads.flat.abs().max()) == 0.0
    views_ok = all(p.grad is v for m in (G, D) for p, v in zip(m._sempyr_flat_grads.params, m._sempyr_flat_grads.views))
    cpu = lambda t: None if t is None else t.detach().cpu()
    torch.save({'rank': rank, 'world': world, 'backend': torch.distributed.get_backend(), 'images': images, 'init': init,
                'draws': [{k: (cpu(v) if torch.is_tensor(v) else v) for k, v in d.items()} for d in draws],
                'snaps': snaps, 'logged': [(it, dict(s)) for it, s in logged], 'flat_zero': flat_zero, 'views_ok': views_ok,
                'device': torch.cuda.current_device(), 'bucket_stats': dict(D._sempyr_flat_grads.stats),
                'n_buckets': len(D._sempyr_flat_grads.buckets)},
               os.path.join(out_dir, f'rank{rank}.pt'))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
