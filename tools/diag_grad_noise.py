"""How far are HIP-path gradients from the fp64 oracle, next to the fp32 CPU oracle's own distance from fp64?
(basis of the tolerance in tests/test_steps_gpu.py)"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import test_steps_gpu as T
from oracle import steps
dev = torch.device('cuda:0')
f64 = lambda sd: {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
for case in T.GRAD_CASES:
    gen_type, norm, div = case
    for which in ('d', 'g'):
        tm, G, D, C, tr, images, feats, masks, z, z2, alpha = T._setup(dev, gen_type, norm, div, 71 if which == 'd' else 72)
        gsd = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
        dsd = {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}
        csd = {k: v.detach().cpu().clone() for k, v in C.state_dict().items()}
        d_loss_fn, g_loss_fn = tm.get_wgan_losses_fn()
        crit = tm.L1Loss()
        noise = {}
        for nt in (1, 8):
            torch.set_num_threads(nt)
            t32 = steps.Trainer(gsd, dsd, csd, gen_type, norm, use_diversity=div)
            if which == 'd':
                t32.d_step((images - 0.5) / 0.5, feats, masks, z, alpha); src, names = t32.d, t32.d_params
            else:
                t32.g_step(feats, masks, 3, z, z2 if div else None); src, names = t32.g, t32.g_params
            noise[nt] = {k: src[k].grad.detach().clone() for k in names if src[k].grad is not None}
        torch.set_num_threads(8)
        if which == 'd':
            tr.d_step(T._d((images - 0.5) / 0.5), T._d(feats), T._d(masks), T._d(z), T._d(alpha))
            opt = T._Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
            tm.train_discriminator(G, D, d_loss_fn, opt, ((images - 0.5) / 0.5).to(dev), [f.to(dev) for f in feats], [m.to(dev) for m in masks], z=z.to(dev), alpha=alpha.to(dev))
            ref, mod = tr.d, D
        else:
            tr.g_step(T._d(feats), T._d(masks), 3, T._d(z), T._d(z2) if div else None)
            opt = T._Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
            tm.train_generator(G, D, g_loss_fn, opt, 2, [f.to(dev) for f in feats], crit, 3, C, tm.Normalize(T.recipes.CLF_MEAN, T.recipes.CLF_STD), crit, crit, [m.to(dev) for m in masks], [2], z=z.to(dev), z2=z2.to(dev) if div else None)
            ref, mod = tr.g, G
        worst = []
        for k, p in mod.named_parameters():
            if ref[k].grad is None or p.grad is None: continue
            w = ref[k].grad.double(); scale = float(w.abs().max()) + 1e-30
            e_hip = float((p.grad.detach().cpu().double() - w).abs().max()) / scale
            e1 = float((noise[1][k].double() - w).abs().max()) / scale
            e8 = float((noise[8][k].double() - w).abs().max()) / scale
            # robust (99.5th percentile) versions
            q = lambda t: float(torch.quantile(((t.double() - w).abs() / scale).flatten()[:4000000], 0.995)) if t.numel() > 200 else float(((t.double() - w).abs() / scale).max())
            worst.append((e_hip / max(e1, e8, 1e-12), k, e_hip, e1, e8, q(p.grad.detach().cpu()), q(noise[8][k])))
        worst.sort(reverse=True)
        print(case, which, 'worst ratios (hip/max(fp32 1t,8t)), name, e_hip, e_1t, e_8t, q995_hip, q995_8t')
        for w in worst[:6]:
            print('   %.2f %s %.2e %.2e %.2e %.2e %.2e' % w)
