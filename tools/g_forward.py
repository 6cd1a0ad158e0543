import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from semantic_pyramid_for_image_generation_amd.generator import Generator
dev = torch.device('cuda:0'); B = 16
torch.manual_seed(0)
G = Generator('res'); G.init_weights(); G.to(dev); G.train()
shapes = [(64, 128), (64, 64), (128, 32), (256, 16), (512, 8), (512, 1)]
feats = [torch.randn(B, c, h, h, device=dev) for c, h in shapes]
masks = [torch.ones_like(f) for f in feats]
z = torch.randn(B, 128, 1, 1, device=dev)
with torch.no_grad():
    for _ in range(4):
        G(z, feats, masks)
torch.cuda.synchronize()
