"""time one conv primitive: python tools/diag_conv_time.py <fwd|dgrad|wgrad> B Cin H W Cout k s p ups"""
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
def f():
    if mode == 'fwd': ops._raw_conv_fwd(plan, x, w, None)
    elif mode == 'dgrad': ops._raw_conv_dgrad(plan.noups(), gy, w, None)
    else: ops._raw_conv_wgrad(plan, gy, x)
for _ in range(3): f()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print(f"{' '.join(sys.argv[1:])} dbg={os.environ.get('SEMPYR_DBG','0')}: {e0.elapsed_time(e1)/20*1000:.1f} us")
