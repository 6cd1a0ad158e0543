"""Correctness + timing of the weight-gradient kernels on the long-reduction shapes against torch CPU fp64 (tuning aid):
python tools/wp_check.py [nocheck] [small]      (SEMPYR_WPATCH=0 in the environment: the generic gather kernel)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops
from semantic_pyramid_for_image_generation_amd._lib import CONV_WGRAD

dev = torch.device('cuda:0')
SHAPES = [   # name, B, Cin, H, W (stored input), Cout, k, stride, ups
    ('G.b7.conv2 64->64 @256', 16, 64, 256, 256, 64, 3, 1, 0),
    ('G.b7.conv1 ups 64->64 @128', 16, 64, 128, 128, 64, 3, 1, 1),
    ('G.b6.conv2 64->64 @128', 16, 64, 128, 128, 64, 3, 1, 0),
    ('G.b6.conv1 ups 128->64 @64', 16, 128, 64, 64, 64, 3, 1, 1),
    ('G.b5 128->128 @64', 16, 128, 64, 64, 128, 3, 1, 0),
    ('G.b5.conv1 ups 256->128 @32', 16, 256, 32, 32, 128, 3, 1, 1),
    ('D.b1 64->64 @128', 16, 64, 128, 128, 64, 4, 2, 0),
    ('D.b1 64->64 @128 B32', 32, 64, 128, 128, 64, 4, 2, 0),
    ('D.b2 64->128 @64', 16, 64, 64, 64, 128, 4, 2, 0),
    ('D.b2 64->128 @64 B32', 32, 64, 64, 64, 128, 4, 2, 0),
]
if 'small' in sys.argv:
    SHAPES = [
        ('3x3 64->64 10x64', 2, 64, 10, 64, 64, 3, 1, 0),
        ('3x3 128->64 7x128', 3, 128, 7, 128, 64, 3, 1, 0),
        ('3x3 ups 64->128 5x32', 2, 64, 5, 32, 128, 3, 1, 1),
        ('3x3 ups 64->64 6x64', 2, 64, 6, 64, 64, 3, 1, 1),
        ('4x4 64->64 12x64', 3, 64, 12, 64, 64, 4, 2, 0),
        ('4x4 64->128 18x128', 2, 64, 18, 128, 128, 4, 2, 0),
    ]


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if 'one' in sys.argv:
    SHAPES = [SHAPES[0], SHAPES[6]]
check = 'nocheck' not in sys.argv
torch.manual_seed(0)
print('%-30s %3s %4s %9s %7s %9s' % ('shape', 'cfg', 'spl', 'wgrad ms', 'TF', 'err'))
for name, b, Cin, H, W, Cout, k, s, ups in SHAPES:
    plan = ops.get_plan((b, Cin, H, W, Cout, k, k, s, 1, ups), dev)
    x = torch.randn(b, Cin, H, W, device=dev)
    gy = torch.randn(b, Cout, plan.Ho, plan.Wo, device=dev)
    gw = ops._raw_conv_wgrad(plan, gy, x)
    err = float('nan')
    if check:
        xd = x.double().cpu()
        if ups:
            xd = F.interpolate(xd, scale_factor=2, mode='nearest')
        gwr = torch.nn.grad.conv2d_weight(xd, (Cout, Cin, k, k), gy.double().cpu(), stride=s, padding=1)
        err = ((gw.double().cpu() - gwr).abs().max() / gwr.abs().max()).item()
    fl = 2.0 * b * plan.Ho * plan.Wo * Cout * Cin * k * k
    t = timeit(lambda: ops._raw_conv_wgrad(plan, gy, x))
    wi = plan.mode(CONV_WGRAD).info
    print('%-30s %3d %4d %9.3f %7.1f %9.2e' % (name, wi.cfg, wi.splits[0], t, fl / t / 1e9, err), flush=True)
