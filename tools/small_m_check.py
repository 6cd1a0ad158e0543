"""Input gradients / forwards with <= 4 GEMM rows against torch fp64, with timing (tuning aid): python tools/small_m_check.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops

dev = torch.device('cuda:0')
torch.manual_seed(0)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# name, B, Cin, H, W, Cout, k, s, p
for name, B, Cin, H, W, Cout, k, s, p in [('classifier stem 7x7/2 dgrad @256', 16, 3, 256, 256, 64, 7, 2, 3),
                                          ('7x7/2 dgrad odd 2x3x30x50', 2, 3, 30, 50, 64, 7, 2, 3),
                                          ('4x4/2 dgrad 3<-64 @64', 4, 3, 64, 64, 64, 4, 2, 1),
                                          ('5x5/1 dgrad 2<-32 @40', 2, 2, 40, 40, 32, 5, 1, 2)]:
    plan = ops.get_plan((B, Cin, H, W, Cout, k, k, s, p, 0), dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    gy = torch.randn(B, Cout, plan.Ho, plan.Wo, device=dev)
    gx = ops._raw_conv_dgrad(plan, gy, w, None)
    op = (H + 2 * p - k) % s, (W + 2 * p - k) % s
    ref = F.conv_transpose2d(gy.double().cpu(), w.double().cpu(), stride=s, padding=p, output_padding=op)
    err = ((gx.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    t = timeit(lambda: ops._raw_conv_dgrad(plan, gy, w, None))
    print('%-36s %8.3f ms  err %.2e' % (name, t, err), flush=True)
