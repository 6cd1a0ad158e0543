import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from semantic_pyramid_for_image_generation_amd import ops
dev = torch.device('cuda:0')
for kind in ('batch', 'instance'):
    g = torch.Generator().manual_seed(13)
    shape = (16, 64, 256, 256)
    x = (torch.randn(shape, generator=g) * 1.7 + 3.0).to(dev).requires_grad_(True)
    gam = (torch.randn(64, generator=g) * 0.3 + 1).to(dev).requires_grad_(True)
    bet = (torch.randn(64, generator=g) * 0.3).to(dev).requires_grad_(True)
    gy = torch.randn(shape, generator=g).to(dev)
    code = ops.NORM_BATCH if kind == 'batch' else ops.NORM_INSTANCE
    y = ops.norm_act(x, gam, bet, code, ops.ACT_LRELU, 0.2)
    gx, gg, gb = torch.autograd.grad(y, (x, gam, bet), gy)
    xd = x.detach().double().requires_grad_(True); gd = gam.detach().double().requires_grad_(True); bd = bet.detach().double().requires_grad_(True)
    dims = (0, 2, 3) if kind == 'batch' else (2, 3)
    mu = xd.mean(dims, keepdim=True); var = xd.var(dims, unbiased=False, keepdim=True)
    pre = (xd - mu) / torch.sqrt(var + 1e-5) * gd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1)
    yr = torch.nn.functional.leaky_relu(pre, 0.2)
    gxr, ggr, gbr = torch.autograd.grad(yr, (xd, gd, bd), gy.double())
    e = (gx.double() - gxr).abs(); s = float(gxr.abs().max())
    bad = e > 1e-3 * s
    print(kind, 'gx scale', s, 'max err', float(e.max()), 'outliers', int(bad.sum()), 'of', e.numel(),
          'max err without outliers', float(e[~bad].max()), 'min |pre| at outliers', float(pre.detach().abs()[bad].max()) if bad.any() else None)
    print('  gg rel', float((gg.double()-ggr).abs().max()/ggr.abs().max()), 'gb rel', float((gb.double()-gbr).abs().max()/gbr.abs().max()))
