"""split-bf16 (SEMPYR_MATH_BF16X3) vs exact-fp32 3x3 convolutions: error against fp64 and time per launch"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)


def run(shape, cout, ups, math, reps=0):
    B, C, H, W = shape
    x = torch.randn(shape, generator=g) ; w = torch.randn(cout, C, 3, 3, generator=g) * 0.05
    gy = torch.randn(B, cout, H << ups, W << ups, generator=g)
    ops.CONV_MATH = math
    xd = x.to(dev).requires_grad_(True); wd = w.to(dev).requires_grad_(True)
    y = ops.conv2d(xd, wd, None, 1, 1, upsample=bool(ups))
    gx, gw = torch.autograd.grad(y, [xd, wd], gy.to(dev))
    out = None
    if reps:
        with torch.no_grad():
            for _ in range(3): ops.conv2d(xd, wd, None, 1, 1, upsample=bool(ups))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): ops.conv2d(xd, wd, None, 1, 1, upsample=bool(ups))
            torch.cuda.synchronize(); out = (time.perf_counter() - t0) / reps * 1e6
    ops.CONV_MATH = 0
    return (x, w, gy), (y.detach().cpu(), gx.cpu(), gw.cpu()), out


def ref64(x, w, gy, ups):
    x = x.double().requires_grad_(True); w = w.double().requires_grad_(True)
    xi = F.interpolate(x, scale_factor=2, mode='nearest') if ups else x
    y = F.conv2d(xi, w, None, 1, 1)
    gx, gw = torch.autograd.grad(y, [x, w], gy.double())
    return y.detach(), gx, gw


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


for shape, cout, ups, reps in [((2, 64, 64, 64), 64, 0, 0), ((2, 64, 32, 32), 64, 1, 0), ((2, 128, 32, 32), 64, 0, 0),
                               ((2, 512, 8, 8), 256, 1, 0), ((16, 64, 256, 256), 64, 0, 20), ((16, 64, 128, 128), 64, 1, 20),
                               ((16, 128, 64, 64), 64, 0, 20), ((16, 64, 128, 128), 64, 0, 20)]:
    g.manual_seed(1)
    (x, w, gy), o32, t32 = run(shape, cout, ups, 0, reps)
    g.manual_seed(1)
    _, o16, t16 = run(shape, cout, ups, 1, reps)
    r = ref64(x, w, gy, ups) if shape[0] <= 2 else None
    line = f"{shape} -> {cout} ups {ups}: "
    if r is not None:
        line += "err vs fp64 (y, gx, gw): f32 " + ' '.join(f'{rel(a, b):.1e}' for a, b in zip(o32, r)) + " | bf16x3 " + ' '.join(f'{rel(a, b):.1e}' for a, b in zip(o16, r))
    else:
        line += "bf16x3 vs f32 (y, gx): " + ' '.join(f'{rel(a, b.double()):.1e}' for a, b in zip(o16[:2], o32[:2]))
    if t32:
        line += f" | fwd time f32 {t32:.0f} us, bf16x3 {t16:.0f} us"
    print(line, flush=True)
