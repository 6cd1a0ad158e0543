import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import test_steps_gpu as T
import semantic_pyramid_for_image_generation_amd.train_model as _tm0
_tm0._samplewise_orig = _tm0._samplewise
dev = torch.device('cuda:0')
case = ('res', 'instance_norm', True)
res = []
for rep in range(2):
    # rep 0: reference-style separate real / fake critic passes; rep 1: the batched pass
    import semantic_pyramid_for_image_generation_amd.train_model as _tm
    _tm._samplewise = (lambda m: False) if rep == 0 else _tm.__dict__.get('_samplewise_orig', _tm._samplewise)
    tm, G, D, C, tr, images, feats, masks, z, z2, alpha = T._setup(dev, *case, 71)
    d_loss_fn, _ = tm.get_wgan_losses_fn()
    d_opt = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    if rep == 0:
        want = tr.d_step(T._d((images - 0.5) / 0.5), T._d(feats), T._d(masks), T._d(z), T._d(alpha))
        ref = {k: tr.d[k].grad.clone() for k in tr.d_params}
    got = tm.train_discriminator(G, D, d_loss_fn, d_opt, ((images - 0.5) / 0.5).to(dev), [f.to(dev) for f in feats],
                                 [m.to(dev) for m in masks], z=z.to(dev), alpha=alpha.to(dev))
    res.append({k: p.grad.detach().cpu().double().clone() for k, p in D.named_parameters()})
    print(rep, {k: float(v) for k, v in got.items()}, {k: float(v) for k, v in want.items()})
for k in ref:
    s = float(ref[k].abs().max())
    e0 = float((res[0][k] - ref[k]).abs().max()); e1 = float((res[1][k] - ref[k]).abs().max())
    d01 = float((res[0][k] - res[1][k]).abs().max())
    print('%-45s scale %.3e  err0/s %.2e err1/s %.2e  run0-run1 %.2e' % (k, s, e0 / max(s, 1e-30), e1 / max(s, 1e-30), d01))
k = 'conv_block5.conv_block1.block.1.bias'
e = (res[1][k] - ref[k]).abs(); s = float(ref[k].abs().max())
print('batched run, %s: %d of %d channels off by > 1e-3*scale; their errors / scale: %s' % (k, int((e > 1e-3 * s).sum()), e.numel(), [round(float(v) / s, 4) for v in e[e > 1e-3 * s]]))
