"""Graph-replayed TrainLoop vs eager: bit-exactness over a few iterations and cycle time at the bench configuration."""
import os, sys, time, random
from argparse import Namespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_models
from semantic_pyramid_for_image_generation_amd import ops, optim, loop
from semantic_pyramid_for_image_generation_amd import train_model as tm

dev = torch.device('cuda:0')
torch.cuda.set_device(0)
B = int(os.environ.get('B', '16'))
norm = os.environ.get('NORM', 'instance_norm')
gen_type = os.environ.get('GEN', 'res')
tm.args = Namespace(use_diversity_loss=os.environ.get('DIV', '0') == '1', gradient_penalty_weight=10.0)


def make(use_graphs):
    G, D, C = build_models(dev, gen_type, norm)
    g_opt = optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True)
    d_opt = optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True)
    lp = loop.TrainLoop(G, D, C, g_opt, d_opt, B, dev, use_graphs=use_graphs)
    return G, D, lp


def run(use_graphs, iters, prepare):
    G, D, lp = make(use_graphs)
    gen = torch.Generator().manual_seed(1234)
    pool = [torch.rand(B, 3, 256, 256, generator=gen).to(dev) for _ in range(3)]
    if prepare:
        t0 = time.time(); lp.prepare(); print('prepare: %.1f s, %d graphs' % (time.time() - t0, len(lp.graphs)), flush=True)
    torch.manual_seed(7); random.seed(7)
    losses = []
    for i in range(iters):
        out = lp.iteration(pool[i % 3])
        losses.append([float(v) for v in out['d'].values()] + ([float(v) for v in out['g'].values()] if out['g'] else []))
    torch.cuda.synchronize()
    return losses, {k: v.detach().clone() for k, v in G.state_dict().items()}, {k: v.detach().clone() for k, v in D.state_dict().items()}, lp, pool


if os.environ.get('CHECK', '1') == '1':
    n = int(os.environ.get('CHECK_ITERS', '7'))
    le, ge, de, _, _ = run(False, n, False)
    lg, gg, dg, _, _ = run(True, n, True)
    bad = 0
    for i, (a, b) in enumerate(zip(le, lg)):
        if a != b:
            print('iter', i, 'losses differ', a, b); bad += 1
    for k in ge:
        if not torch.equal(ge[k], gg[k]):
            print('G', k, float((ge[k].float() - gg[k].float()).abs().max())); bad += 1
    for k in de:
        if not torch.equal(de[k], dg[k]):
            print('D', k, float((de[k].float() - dg[k].float()).abs().max())); bad += 1
    print('bit-exact:', bad == 0, flush=True)

for mode in ([True, False] if os.environ.get('TIME', '1') == '1' else []):
    G, D, lp = make(mode)

    gen = torch.Generator().manual_seed(1234)
    pool = [torch.rand(B, 3, 256, 256, generator=gen).to(dev) for _ in range(5)]
    lp.prepare()
    if mode:
        print('graph stats [kernels, memsets, memcpys, empty, streams, waits, events, edges]:', {k: c.stats for k, c in lp.graphs.items()}, flush=True)
    torch.manual_seed(7); random.seed(0)
    for i in range(15):
        lp.iteration(pool[i % 5])
    torch.cuda.synchronize()
    cycles = int(os.environ.get('CYCLES', '6'))
    t0 = time.perf_counter()
    for i in range(5 * cycles):
        lp.iteration(pool[i % 5])
    issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print('graphs=%s: %.2f ms/cycle (%.1f iters/s), host issue %.2f ms/cycle' % (mode, 1e3 * el / cycles, 5 * cycles / el, 1e3 * issue / cycles), flush=True)
