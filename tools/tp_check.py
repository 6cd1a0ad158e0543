"""Correctness + timing of the 4x4 / stride-2 conv primitives on the critic's shapes against torch CPU fp64 (tuning aid):
python tools/tp_check.py [nocheck] [b32]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops
from semantic_pyramid_for_image_generation_amd._lib import CONV_FWD, CONV_DGRAD, CONV_WGRAD

dev = torch.device('cuda:0')
B = 32 if 'b32' in sys.argv else 16
SHAPES = [
    ('D.b1 64->64 @128', B, 64, 128, 128, 64),
    ('D.b2 64->128 @64', B, 64, 64, 64, 128),
    ('D.b3 128->256 @32', B, 128, 32, 32, 256),
    ('D.b4 256->512 @16', B, 256, 16, 16, 512),
    ('D.b5 512->512 @8', B, 512, 8, 8, 512),
    ('odd 64->64 @24x40', 6, 64, 24, 40, 64),
    ('odd 72->192 @12x20', 4, 72, 12, 20, 192),
]


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if 'ksweep' in sys.argv:
    SHAPES = [('b1 C=%d' % c, B, c, 128, 128, 64) for c in (32, 64, 128, 256)] + [('b3 C=%d' % c, B, c, 32, 32, 256) for c in (64, 128, 256, 512)]
check = 'nocheck' not in sys.argv
torch.manual_seed(0)
print('%-22s %5s %3s %8s %8s %8s | %7s %7s %7s | %9s %9s %9s' % ('shape', 'cfg', 'spl', 'fwd', 'dgrad', 'wgrad', 'TF f', 'TF d', 'TF w', 'err f', 'err d', 'err w'))
for name, b, Cin, H, W, Cout in SHAPES:
    plan = ops.get_plan((b, Cin, H, W, Cout, 4, 4, 2, 1, 0), dev)
    x = torch.randn(b, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, 4, 4, device=dev) * 0.05
    bias = torch.randn(Cout, device=dev)
    y = ops._raw_conv_fwd(plan, x, w, bias)
    gy = torch.randn_like(y)
    gx = ops._raw_conv_dgrad(plan, gy, w, None)
    gw = ops._raw_conv_wgrad(plan, gy, x)
    ef = ed = ew = float('nan')
    if check:
        xd, wd, gyd = x.double().cpu(), w.double().cpu(), gy.double().cpu()
        yr = F.conv2d(xd, wd, bias.double().cpu(), stride=2, padding=1)
        ef = ((y.double().cpu() - yr).abs().max() / yr.abs().max()).item()
        gxr = F.conv_transpose2d(gyd, wd, stride=2, padding=1)
        ed = ((gx.double().cpu() - gxr).abs().max() / gxr.abs().max()).item()
        gwr = torch.nn.grad.conv2d_weight(xd, wd.shape, gyd, stride=2, padding=1)
        ew = ((gw.double().cpu() - gwr).abs().max() / gwr.abs().max()).item()
    fl = 2.0 * b * plan.Ho * plan.Wo * Cout * Cin * 16
    t_f = timeit(lambda: ops._raw_conv_fwd(plan, x, w, bias))
    t_d = timeit(lambda: ops._raw_conv_dgrad(plan, gy, w, None))
    t_w = timeit(lambda: ops._raw_conv_wgrad(plan, gy, x))
    fi, di = plan.mode(CONV_FWD).info, plan.mode(CONV_DGRAD).info
    print('%-22s %2d/%2d %3d %8.3f %8.3f %8.3f | %7.1f %7.1f %7.1f | %9.2e %9.2e %9.2e' % (
        name, fi.cfg, di.cfg, fi.splits[0], t_f, t_d, t_w, fl / t_f / 1e9, fl / t_d / 1e9, fl / t_w / 1e9, ef, ed, ew), flush=True)
