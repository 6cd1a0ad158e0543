"""Phase timing of one critic step and one generator step at batch 16 (single stream, events between phases)."""
import sys, os, random
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from argparse import Namespace
from semantic_pyramid_for_image_generation_amd import ops, train_model as tm
from semantic_pyramid_for_image_generation_amd.classifier import Classifier, resnet18
from semantic_pyramid_for_image_generation_amd.discriminator import Discriminator
from semantic_pyramid_for_image_generation_amd.generator import Generator
dev = torch.device('cuda:0'); B = 16
torch.manual_seed(0)
G = Generator('res'); G.init_weights(); D = Discriminator('instance_norm', 'res'); D.init_weights(); C = Classifier(resnet18(365))
G.to(dev); D.to(dev); C.to(dev).eval()
for p in C.parameters(): p.requires_grad_(False)
tm.args = Namespace(use_diversity_loss=False, gradient_penalty_weight=10.0)
ops.TWO_STREAM_CRITIC = ops.TWO_STREAM_GENERATOR = ops._ASYNC_ENV = False
images = torch.rand(B, 3, 256, 256, device=dev)
norm_clf = tm.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
d_loss_fn, g_loss_fn = tm.get_wgan_losses_fn()
from semantic_pyramid_for_image_generation_amd import optim
d_opt = optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999)); g_opt = optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
def T():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def d_step(rec):
    ev = [T()]
    with torch.no_grad():
        _, feats = C(norm_clf(images))
    masks = [torch.ones_like(f) for f in feats]; ev.append(T())
    real = (images - 0.5) / 0.5
    z = torch.randn(B, 128, 1, 1, device=dev)
    with torch.no_grad(): fake = G(z, feats, masks)
    ev.append(T())
    logits = D(torch.cat([real, fake])); rl, fl = d_loss_fn(logits[:B], logits[B:]); ev.append(T())
    gp = tm.gradient_penalty(D, real, fake); ev.append(T())
    loss = rl + fl + 10 * gp
    D.zero_grad(); loss.backward(); ev.append(T())
    d_opt.step(); ev.append(T())
    torch.cuda.synchronize()
    if rec is not None:
        rec.append([ev[i].elapsed_time(ev[i + 1]) for i in range(len(ev) - 1)])
    return feats, masks
def g_step(feats, masks, rec):
    ev = [T()]
    z = torch.randn(B, 128, 1, 1, device=dev)
    with tm._frozen(D, C):
        fake = G(z, feats, masks); ev.append(T())
        gl = g_loss_fn(D(fake)); ev.append(T())
        _, ff = C(norm_clf(fake)); fl = tm.getLossByTrainType(feats, masks, [2], 3, ff, tm.L1Loss()); ev.append(T())
        G.zero_grad(); (gl + fl).backward(); ev.append(T())
    g_opt.step(); ev.append(T()); torch.cuda.synchronize()
    if rec is not None:
        rec.append([ev[i].elapsed_time(ev[i + 1]) for i in range(len(ev) - 1)])
f, m = d_step(None); g_step(f, m, None)
rd, rg = [], []
for _ in range(3):
    f, m = d_step(rd); g_step(f, m, rg)
import numpy as np
print('D step  [classifier, G fwd(no grad), D(real+fake) fwd, GP fwd+inner bwd, backward, Adam] ms:', np.round(np.mean(rd, 0), 2), 'sum', round(float(np.sum(np.mean(rd, 0))), 2))
print('G step  [G fwd, D(fake) fwd, C fwd + feat loss, backward, Adam] ms:', np.round(np.mean(rg, 0), 2), 'sum', round(float(np.sum(np.mean(rg, 0))), 2))
