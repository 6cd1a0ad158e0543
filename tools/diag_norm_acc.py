"""Accuracy of the norm kernels (forward, backward, backward-of-backward) against fp64, next to torch's own fp32 CPU ops."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops

dev = torch.device('cuda:0')


def tn(kind, x, gam, bet, eps=1e-5):
    if kind == 'batch':
        return F.batch_norm(x, None, None, gam, bet, True, 0.1, eps)
    if kind == 'instance':
        return F.instance_norm(x, None, None, gam, bet, True, 0.1, eps)
    return F.group_norm(x, 1, gam, bet, eps)


def run(fn, x, gam, bet, gy, u):
    x = x.clone().requires_grad_(True); gam = gam.clone().requires_grad_(True); bet = bet.clone().requires_grad_(True)
    gy = gy.clone().requires_grad_(True)
    y = fn(x, gam, bet)
    gx, ggam, gbet = torch.autograd.grad(y, [x, gam, bet], gy, create_graph=True)
    L = (gx * u).sum()
    r_gy, r_x, r_gam = torch.autograd.grad(L, [gy, x, gam])
    return y, gx, ggam, gbet, r_gy, r_x, r_gam


def rel(o, r):
    o = o.detach().double().cpu(); r = r.detach().double().cpu()
    return float((o - r).abs().max() / r.abs().max().clamp_min(1e-30))


names = ['y', 'gx', 'ggam', 'gbet', 'dd_gy', 'dd_x', 'dd_gam']
for kind in ['batch', 'instance', 'layer']:
    KIND = {'batch': ops.NORM_BATCH, 'instance': ops.NORM_INSTANCE, 'layer': ops.NORM_LAYER}[kind]
    for shape in [(2, 512, 4, 4), (2, 512, 8, 8), (2, 256, 16, 16), (2, 64, 64, 64), (16, 512, 8, 8)]:
        for mean in (0.3, 3.0):
            g = torch.Generator().manual_seed(5)
            x = torch.randn(shape, generator=g) * 1.5 + mean
            gam = torch.rand(shape[1], generator=g) + 0.5; bet = torch.randn(shape[1], generator=g) * 0.1
            gy0 = torch.randn(shape, generator=g); u0 = torch.randn(shape, generator=g)
            f = lambda x, gam, bet: F.leaky_relu(tn(kind, x, gam, bet), 0.2)
            ref = run(f, x.double(), gam.double(), bet.double(), gy0.double(), u0.double())
            cpu = run(f, x, gam, bet, gy0, u0)
            hip = run(lambda x, gam, bet: ops.norm_act(x, gam, bet, KIND, ops.ACT_LRELU, 0.2), x.to(dev), gam.to(dev), bet.to(dev),
                      gy0.to(dev), u0.to(dev))
            print(kind, shape, 'mean', mean, ' '.join(f'{n}:{rel(h, r):.1e}/{rel(c, r):.1e}' for n, h, c, r in zip(names, hip, cpu, ref)), flush=True)
