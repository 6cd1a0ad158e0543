import sys, os, cProfile, pstats, io
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from semantic_pyramid_for_image_generation_amd import ops
dev = torch.device('cuda:0')
x = torch.randn(2, 16, 8, 8, device=dev); w = torch.randn(16, 16, 3, 3, device=dev); b = torch.randn(16, device=dev)
with torch.no_grad():
    for _ in range(100): ops.conv2d(x, w, b, 1, 1)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3000): ops.conv2d(x, w, b, 1, 1)
    pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:3500])
