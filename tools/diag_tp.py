"""Every 4x4/s2 forward-conv call of one critic step (layer_norm, B=2) against torch CPU fp64 (diagnostic)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, torch.nn.functional as F
import test_steps_gpu as T
from semantic_pyramid_for_image_generation_amd import ops
from semantic_pyramid_for_image_generation_amd._lib import CONV_FWD
dev = torch.device('cuda:0')
orig = ops._raw_conv_fwd
def patched(plan, x, w, bias, act_kind=0, slope=0.0, pooled=None, scale_ptr=0, res=None, res_half=False, bits=None, mask_bits=None, want_y=True):
    y = orig(plan, x, w, bias, act_kind, slope, pooled, scale_ptr, res, res_half, bits, mask_bits, want_y)
    if plan.kh == 4 and plan.stride == 2 and y is not None:
        torch.cuda.synchronize()
        b = bias.double().cpu() if isinstance(bias, torch.Tensor) else None
        yr = F.conv2d(x.double().cpu(), w.double().cpu(), b, stride=2, padding=1)
        e = (y.double().cpu() - yr).abs()
        print('conv fwd B=%d Cin=%d H=%d Cout=%d cfg=%d splits=%d bias=%s act=%d res=%s scale=%s contiguous=%s | max err %.2e / max|y| %.2e = %.2e | stream %x' % (
            plan.B, plan.Cin, plan.Hin, plan.Cout, plan.mode(CONV_FWD).info.cfg, plan.mode(CONV_FWD).info.splits[0], type(bias).__name__, act_kind,
            res is not None, scale_ptr != 0, x.is_contiguous(), e.max(), yr.abs().max(), e.max() / yr.abs().max(), torch.cuda.current_stream().cuda_stream), flush=True)
    return y
ops._raw_conv_fwd = patched
case = ('res', 'layer_norm', False)
tm, G, D, C, tr, images, feats, masks, z, z2, alpha = T._setup(dev, *case, 71)
d_loss_fn, _ = tm.get_wgan_losses_fn()
d_opt = T._Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
got = tm.train_discriminator(G, D, d_loss_fn, d_opt, ((images - 0.5) / 0.5).to(dev), [f.to(dev) for f in feats],
                             [m.to(dev) for m in masks], z=z.to(dev), alpha=alpha.to(dev))
torch.cuda.synchronize()
print({k: float(v) for k, v in got.items()})
