"""Micro-benchmark of single conv primitives on the GPU (tuning aid): python tools/bench_conv.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_pyramid_for_image_generation_amd import ops
from semantic_pyramid_for_image_generation_amd._lib import CONV_FWD, CONV_DGRAD, CONV_WGRAD

dev = torch.device('cuda:0')
# (name, B, Cin, H, W, Cout, k, s, p, ups)
MID = [
    ('G.b6.conv2 64->64 3x3 @128', 16, 64, 128, 128, 64, 3, 1, 1, 0),
    ('G.b6.conv1 64->64 3x3 ups @64', 16, 64, 64, 64, 64, 3, 1, 1, 1),
    ('D.b1 64->64 4x4s2 @128', 16, 64, 128, 128, 64, 4, 2, 1, 0),
    ('C.l1 64->64 3x3 @64', 16, 64, 64, 64, 64, 3, 1, 1, 0),
    ('G.b6 128->64 3x3 ups @32', 16, 128, 32, 32, 64, 3, 1, 1, 1),
    ('D.b2 64->128 4x4s2 @64', 16, 64, 64, 64, 128, 4, 2, 1, 0),
    ('C.l2 128->128 3x3 @32', 16, 128, 32, 32, 128, 3, 1, 1, 0),
    ('G.b5 256->128 3x3 ups @16', 16, 256, 16, 16, 128, 3, 1, 1, 1),
    ('C.l3 256->256 3x3 @16', 16, 256, 16, 16, 256, 3, 1, 1, 0),
]
SMALL = [
    ('D.b3 128->256 4x4s2 @32', 16, 128, 32, 32, 256, 4, 2, 1, 0),
    ('D.b4 256->512 4x4s2 @16', 16, 256, 16, 16, 512, 4, 2, 1, 0),
    ('D.b5 512->512 4x4s2 @8', 16, 512, 8, 8, 512, 4, 2, 1, 0),
    ('D.last 512->1 4x4 @4', 16, 512, 4, 4, 1, 4, 1, 0, 0),
    ('G.c2 512->512 4x4s2 (convT) @8', 16, 512, 8, 8, 512, 4, 2, 1, 0),
    ('G.b3 512->256 3x3 ups @8', 16, 512, 8, 8, 256, 3, 1, 1, 1),
    ('G.b3b 256->256 3x3 @16', 16, 256, 16, 16, 256, 3, 1, 1, 0),
    ('G.b4 256->128 3x3 ups @16', 16, 256, 16, 16, 128, 3, 1, 1, 1),
    ('G.b5 128->64 3x3 ups @32', 16, 128, 32, 32, 64, 3, 1, 1, 1),
    ('C.l4 512->512 3x3 @8', 16, 512, 8, 8, 512, 3, 1, 1, 0),
    ('C.l3 256->256 3x3 @16', 16, 256, 16, 16, 256, 3, 1, 1, 0),
    ('C.l2 128->128 3x3 @32', 16, 128, 32, 32, 128, 3, 1, 1, 0),
]
SHAPES = [
    ('G.b7.conv2 64->64 3x3 @256', 16, 64, 256, 256, 64, 3, 1, 1, 0),
    ('G.b7.conv1 64->64 3x3 ups @128', 16, 64, 128, 128, 64, 3, 1, 1, 1),
    ('G.b7.skip 64->64 1x1 ups @128', 16, 64, 128, 128, 64, 1, 1, 0, 1),
    ('D.first 3->64 3x3 @256', 16, 3, 256, 256, 64, 3, 1, 1, 0),
    ('G.last 64->3 3x3 @256', 16, 64, 256, 256, 3, 3, 1, 1, 0),
    ('D.b1 64->64 4x4s2 @128', 16, 64, 128, 128, 64, 4, 2, 1, 0),
    ('D.b2 64->128 4x4s2 @64', 16, 64, 64, 64, 128, 4, 2, 1, 0),
    ('D.b4 256->512 4x4s2 @16', 16, 256, 16, 16, 512, 4, 2, 1, 0),
    ('D.b5 512->512 4x4s2 @8', 16, 512, 8, 8, 512, 4, 2, 1, 0),
    ('G.b3 512->256 3x3 ups @8', 16, 512, 8, 8, 256, 3, 1, 1, 1),
    ('C.l1 64->64 3x3 @64', 16, 64, 64, 64, 64, 3, 1, 1, 0),
    ('C.stem 3->64 7x7s2 @256', 16, 3, 256, 256, 64, 7, 2, 3, 0),
]


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print('%-34s %8s %8s %8s   (ms / TFLOP/s)' % ('shape', 'fwd', 'dgrad', 'wgrad'))
import sys
for name, B, Cin, H, W, Cout, k, s, p, ups in (SMALL if 'small' in sys.argv else MID if 'mid' in sys.argv else SHAPES):
    plan = ops.get_plan((B, Cin, H, W, Cout, k, k, s, p, ups), dev)
    x = torch.randn(B, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    y = ops._raw_conv_fwd(plan, x, w, None)
    gy = torch.randn_like(y)
    fl = 2.0 * B * plan.Ho * plan.Wo * Cout * Cin * k * k
    t_f = timeit(lambda: ops._raw_conv_fwd(plan, x, w, None))
    t_d = timeit(lambda: ops._raw_conv_dgrad(plan.noups(), gy, w, None))
    t_w = timeit(lambda: ops._raw_conv_wgrad(plan, gy, x))
    print('%-34s %8.3f %8.3f %8.3f   %6.1f %6.1f %6.1f' % (name, t_f, t_d, t_w, fl / t_f / 1e9, fl / t_d / 1e9, fl / t_w / 1e9), flush=True)
