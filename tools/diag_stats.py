"""conv + BatchNorm statistics: epilogue partials + finalize vs plain conv + row_moments pass (diagnostic)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_pyramid_for_image_generation_amd import ops
dev = torch.device('cuda:0')
def t(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for (B, C, H, ups) in [(16, 64, 256, 0), (16, 64, 128, 1), (16, 64, 128, 0), (16, 64, 64, 1), (16, 64, 64, 0), (16, 128, 32, 0)]:
    x = torch.randn(B, C, H, H, device=dev); w = torch.randn(64, C, 3, 3, device=dev) * 0.05
    gam = torch.rand(64, device=dev) + 0.5; bet = torch.randn(64, device=dev) * 0.1
    with torch.no_grad():
        y = ops.conv2d(x, w, None, 1, 1, upsample=bool(ups))
        a = t(lambda: ops.conv2d(x, w, None, 1, 1, upsample=bool(ups)))
        b = t(lambda: ops.conv2d(x, w, None, 1, 1, upsample=bool(ups), stats=True))
        c = t(lambda: ops._norm_stats(y, gam, bet, ops.NORM_BATCH, 1e-5, None))
        ys = ops.conv2d(x, w, None, 1, 1, upsample=bool(ups), stats=True)
        d = t(lambda: ops._norm_stats(ys, gam, bet, ops.NORM_BATCH, 1e-5, None, 1, ops.tile_stats_of(ys))) if ops.tile_stats_of(ys) else float('nan')
    print(f"B{B} C{C} {H}x{H} ups{ups}: conv {a:7.1f}  conv+stats {b:7.1f} (+{b-a:5.1f}) | row_moments+finalize {c:6.1f}  tile finalize {d:6.1f} | net {a+c:7.1f} -> {b+d:7.1f}", flush=True)
