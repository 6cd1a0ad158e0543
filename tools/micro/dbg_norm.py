import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from semantic_pyramid_for_image_generation_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for shape in [(2, 512, 4, 4), (4, 512, 4, 4), (4, 64, 4, 4), (4, 512, 8, 8), (6, 16, 4, 4)]:
    x = torch.randn(shape, dtype=torch.float64) + 0.3; gam = torch.randn(shape[1], dtype=torch.float64) + 1; bet = torch.randn(shape[1], dtype=torch.float64)
    gy = torch.randn(shape, dtype=torch.float64) + 2.0
    t = [v.clone().requires_grad_(True) for v in (x, gam, bet)]
    y = F.leaky_relu(F.instance_norm(t[0], None, None, t[1], t[2], True, 0.1, 1e-5), 0.2); y.backward(gy)
    d = [v.float().to(dev).requires_grad_(True) for v in (x, gam, bet)]
    yd = ops.norm_act(d[0], d[1], d[2], ops.NORM_INSTANCE, ops.ACT_LRELU, 0.2); yd.backward(gy.float().to(dev))
    rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max() / b.abs().max())
    print(shape, 'y', rel(yd, y.detach()), 'gx', rel(d[0].grad, t[0].grad), 'ggamma', rel(d[1].grad, t[1].grad), 'gbeta', rel(d[2].grad, t[2].grad))
