"""time the stem kernels alone (diagnostic)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_pyramid_for_image_generation_amd import ops
dev = torch.device('cuda:0')
B, H, W = 16, 256, 256
x = torch.randn(B, 3, H, W, device=dev); w = torch.randn(64, 3, 3, 3, device=dev) * 0.2; b = torch.randn(64, device=dev)
gp = torch.randn(B, 64, H // 2, W // 2, device=dev)
plan = ops.get_plan((B, 3, H, W, 64, 3, 3, 1, 1, 0), dev)
bits = torch.empty(B * 64 * H * W // 32, dtype=torch.int32, device=dev)
pooled = torch.empty_like(gp)
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
print('fwd   %.1f us' % t(lambda: ops._raw_conv_fwd(plan, x, w, b, ops.ACT_LRELU, 0.2, pooled, bits=bits, want_y=False)))
print('wgrad %.1f us (incl. reduce)' % t(lambda: ops._raw_stem_wgrad(plan, gp, bits, 0.2, x, want_b=True)))
print('dgrad %.1f us' % t(lambda: ops._raw_stem_dgrad(plan, gp, bits, 0.2, w)))
