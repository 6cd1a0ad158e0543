"""Host-side cost per operator call (tiny tensors, so the GPU is never the limit)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from semantic_pyramid_for_image_generation_amd import ops
dev = torch.device('cuda:0')
x = torch.randn(2, 16, 8, 8, device=dev); w = torch.randn(16, 16, 3, 3, device=dev); b = torch.randn(16, device=dev)
gam = torch.ones(16, device=dev); bet = torch.zeros(16, device=dev)
def bench(name, fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    host = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    print('%-34s %6.1f us/call (host issue)' % (name, host * 1e6))
with torch.no_grad():
    bench('conv2d fwd (no grad)', lambda: ops.conv2d(x, w, b, 1, 1))
    bench('norm_act instance (no grad)', lambda: ops.norm_act(x, gam, bet, ops.NORM_INSTANCE, ops.ACT_LRELU, 0.2))
    bench('add_act', lambda: ops.add_act(x, x, ops.ACT_RELU))
    bench('torch.empty_like', lambda: torch.empty_like(x))
    bench('torch add (reference)', lambda: x + x)
xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
gr = gam.clone().requires_grad_(True); ber = bet.clone().requires_grad_(True)
def fb():
    y = ops.norm_act(ops.conv2d(xr, wr, br, 1, 1), gr, ber, ops.NORM_INSTANCE, ops.ACT_LRELU, 0.2)
    y.sum().backward()
bench('conv+norm fwd+bwd (autograd)', fb, 500)
def fbt():
    y = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(torch.nn.functional.conv2d(xr, wr, br, 1, 1), weight=gr, bias=ber), 0.2)
    y.sum().backward()
bench('same with torch ops (reference)', fbt, 500)
