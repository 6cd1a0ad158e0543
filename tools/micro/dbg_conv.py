import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (B, Cin, H, W, Cout, k, s, p) in [(2, 3, 32, 32, 64, 3, 1, 1), (2, 64, 32, 32, 64, 3, 1, 1), (1, 16, 8, 8, 32, 1, 1, 0)]:
    x = torch.randn(B, Cin, H, W); w = torch.randn(Cout, Cin, k, k) * 0.1
    yr = F.conv2d(x, w, None, s, p)
    y = ops.conv2d(x.to(dev), w.to(dev), None, s, p).cpu()
    e = (y - yr).abs()
    print((B, Cin, H, W, Cout, k), 'max err', float(e.max()), 'interior err', float(e[:, :, 1:-1, 1:-1].max()) if k > 1 else '-')
    print(' err by batch', [float(e[b].max()) for b in range(B)], 'by row', [round(float(e[:, :, r].max()), 3) for r in range(min(H, 8))])
    print(' err by out channel (first 8)', [round(float(e[:, c].max()), 3) for c in range(8)])
