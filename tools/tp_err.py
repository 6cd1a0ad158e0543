"""Error statistics of the 4x4/s2 forward conv against fp64 (run with SEMPYR_TPATCH=0 / 1 to compare kernels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops
from semantic_pyramid_for_image_generation_amd._lib import CONV_FWD
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (B, Cin, H, W, Cout) in [(2, 64, 128, 128, 64), (4, 64, 32, 32, 64), (4, 64, 16, 16, 128), (4, 128, 8, 8, 256), (16, 64, 128, 128, 64)]:
    plan = ops.get_plan((B, Cin, H, W, Cout, 4, 4, 2, 1, 0), dev)
    x = torch.randn(B, Cin, H, W, device=dev).abs()          # post-activation-like: positive mean
    w = torch.randn(Cout, Cin, 4, 4, device=dev) * 0.05
    bias = torch.randn(Cout, device=dev)
    y1 = ops._raw_conv_fwd(plan, x, w, bias)
    y2 = ops._raw_conv_fwd(plan, x, w, bias)
    yr = F.conv2d(x.double().cpu(), w.double().cpu(), bias.double().cpu(), stride=2, padding=1)
    e = (y1.double().cpu() - yr)
    print('%s cfg %d splits %d | bitwise repeatable %s | max %.2e rms %.2e mean %.2e (|y| rms %.2e)' % (
        (B, Cin, H, W, Cout), plan.mode(CONV_FWD).info.cfg, plan.mode(CONV_FWD).info.splits[0], bool((y1 == y2).all()),
        e.abs().max(), e.pow(2).mean().sqrt(), e.mean(), yr.pow(2).mean().sqrt()))
    # where are the largest errors?
    idx = e.abs().flatten().topk(5).indices
    sh = e.shape
    print('   worst at', [tuple(int(v) for v in torch.unravel_index(i, sh)) for i in idx])
