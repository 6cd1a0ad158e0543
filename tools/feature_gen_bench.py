"""BASELINE config 5 at full size: LevelUpFeaturesGenerator(4) + FeaturesDiscriminator(res, level 3), batch 64 (script default):
5 critic steps + 1 generator step per cycle, synthetic images, seeded random weights."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from argparse import Namespace
from semantic_pyramid_for_image_generation_amd import train_feature_creation_gen as tf, train_model as tm
from semantic_pyramid_for_image_generation_amd.classifier import Classifier, resnet18
from semantic_pyramid_for_image_generation_amd.featuresCreatorGen import LevelUpFeaturesGenerator, FeaturesDiscriminator
dev = torch.device('cuda:0'); B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
Gf = LevelUpFeaturesGenerator(4); Gf.init_weights() if hasattr(Gf, 'init_weights') else None
Df = FeaturesDiscriminator('instance_norm', 'res', 3); Df.init_weights() if hasattr(Df, 'init_weights') else None
C = Classifier(resnet18(365))
Gf.to(dev); Df.to(dev); C.to(dev).eval()
for p in C.parameters(): p.requires_grad_(False)
tf.args = tm.args = Namespace(gradient_penalty_weight=10.0, use_diversity_loss=False)
norm_clf = tm.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
d_loss_fn, g_loss_fn = tm.get_wgan_losses_fn()
from semantic_pyramid_for_image_generation_amd import optim
d_opt = optim.Adam(Df.parameters(), lr=2e-4, betas=(0.5, 0.999)); g_opt = optim.Adam(Gf.parameters(), lr=2e-4, betas=(0.5, 0.999))
crit = tm.L1Loss()
images = torch.rand(B, 3, 256, 256, device=dev)
def cycle():
    for it in range(5):
        with torch.no_grad():
            _, feats = C(norm_clf(images))
        out = tf.train_discriminator(Gf, Df, d_loss_fn, d_opt, feats, 4)
        if it == 1:
            tf.train_generator(Gf, Df, C, g_opt, feats, 4, g_loss_fn, crit, crit)
    return out
cycle(); cycle(); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 3
for _ in range(n): out = cycle()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('feature-gen (f4->f3) batch %d: %.2f ms / cycle of 5 iterations = %.1f iters/s ; last d_loss %.4f gp %.4f' % (B, dt * 1e3, 5 / dt, float(out['d_loss'].detach()), float(out['gp'].detach())))
