"""Which LeakyReLU pre-activations of the critic have a different sign in the HIP run and in the fp64 oracle on the inputs
of tests/test_steps_gpu.py::test_d_step_gradients_vs_oracle[case2] (res / layer_norm, seed 71)?  (diagnostic)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, torch.nn.functional as F
import test_steps_gpu as T
from oracle import nets
dev = torch.device('cuda:0')
case = ('res', 'layer_norm', False)
tm, G, D, C, tr, images, feats, masks, z, z2, alpha = T._setup(dev, *case, 71)
real32 = ((images - 0.5) / 0.5)
real = real32.double()
# the x-hat and the fake batch the HIP step itself sees
cap = {}
orig_sl = tm.sample_line
def sl(r, f, alpha=None):
    out = orig_sl(r, f, alpha); cap['xhat'] = out.detach().clone(); cap['fake'] = f.detach().clone(); return out
tm.sample_line = sl
d_loss_fn, _ = tm.get_wgan_losses_fn()
d_opt = T._Adam(D.parameters(), lr=0.0, betas=(0.5, 0.999))
tm.train_discriminator(G, D, d_loss_fn, d_opt, real32.to(dev), [f.to(dev) for f in feats], [m.to(dev) for m in masks], z=z.to(dev), alpha=alpha.to(dev))
torch.cuda.synchronize()
with torch.no_grad():
    fake = cap['fake'].double().cpu()
    xhat = cap['xhat'].double().cpu()
    for name, x in (('x-hat (penalty branch)', xhat), ('real+fake (logit branch)', torch.cat([real, fake]))):
        # fp64 oracle, block by block
        sd = tr.d
        sd = {k: v.detach() for k, v in sd.items()}
        h = F.avg_pool2d(F.leaky_relu(F.conv2d(x, sd['first_conv.weight'], sd['first_conv.bias'], 1, 1), 0.2), 2)
        hh = D.first_conv
        from semantic_pyramid_for_image_generation_amd import ops
        from semantic_pyramid_for_image_generation_amd.blocks import run_conv
        g = run_conv(D.first_conv, x.float().to(dev), act_kind=ops.ACT_LRELU, slope=0.2, pool=True)
        print(name)
        for i, blk in enumerate((D.conv_block1, D.conv_block2, D.conv_block3, D.conv_block4, D.conv_block5), 1):
            p = 'conv_block%d.conv_block1' % i
            pre = nets._d_norm(sd, p + '.block.1', F.conv2d(h, sd[p + '.block.0.weight'], sd[p + '.block.0.bias'], 2, 1), 'layer_norm', True)
            h = F.leaky_relu(pre, 0.2)
            g = blk(g)
            gc = g.double().cpu()
            flip = (torch.sign(gc) != torch.sign(h)) & (h != 0)
            n = int(flip.sum())
            msg = '  block %d: %d of %d signs differ' % (i, n, h.numel())
            if n:
                msg += '; fp64 |pre-activation| of those: ' + ', '.join('%.2e' % v for v in pre[flip].abs().flatten()[:6].tolist()) + \
                       '  (tensor max %.2e); HIP values: ' % pre.abs().max() + ', '.join('%.2e' % v for v in gc[flip].flatten()[:6].tolist())
            print(msg, flush=True)
