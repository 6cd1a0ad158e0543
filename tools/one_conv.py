"""Run one conv primitive a few times (for rocprofv3 --pmc): python tools/one_conv.py <fwd|dgrad|wgrad> B Cin H W Cout k s p ups"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_pyramid_for_image_generation_amd import ops
mode = sys.argv[1]
B, Cin, H, W, Cout, k, s, p, ups = [int(v) for v in sys.argv[2:11]]
dev = torch.device('cuda:0')
plan = ops.get_plan((B, Cin, H, W, Cout, k, k, s, p, ups), dev)
x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
y = ops._raw_conv_fwd(plan, x, w, None); gy = torch.randn_like(y)
for _ in range(3):
    if mode == 'fwd': ops._raw_conv_fwd(plan, x, w, None)
    elif mode == 'dgrad': ops._raw_conv_dgrad(plan.noups(), gy, w, None)
    else: ops._raw_conv_wgrad(plan, gy, x)
torch.cuda.synchronize()
