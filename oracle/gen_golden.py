"""Generate tests/golden/* by running the REFERENCE's own modules (imported from /root/reference, CPU).

Run in the build container only:   python oracle/gen_golden.py
The reference never travels: only the small input/output vectors written here are committed.
torchvision / tensorboardX are not installed, so empty stub modules are registered before importing
`train_model` / `classifier` (their top-level imports are unused by the functions exercised here);
the ResNet-18 the reference's `Classifier` wraps is oracle/resnet18_cpu.py (torchvision's published
architecture) because torchvision is an un-vendored dependency.
"""
import hashlib
import json
import os
import random
import sys
import types
from argparse import Namespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

for name in ['torchvision', 'torchvision.datasets', 'torchvision.transforms', 'torchvision.utils',
             'torchvision.models', 'tensorboardX']:
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
tv = sys.modules['torchvision']
tv.datasets = sys.modules['torchvision.datasets']
tv.transforms = sys.modules['torchvision.transforms']
tv.utils = sys.modules['torchvision.utils']
tv.models = sys.modules['torchvision.models']
sys.modules['torchvision.datasets'].ImageFolder = object     # base class of an unused helper in eval_model.py

import generator as ref_gen            # noqa: E402
import discriminator as ref_dis        # noqa: E402
import featuresCreatorGen as ref_fc    # noqa: E402
import train_model as ref_tm           # noqa: E402
import classifier as ref_clf           # noqa: E402
import train_feature_creation_gen as ref_tf   # noqa: E402

from oracle.resnet18_cpu import ResNet18, randomize_bn_stats   # noqa: E402
from oracle import recipes                                      # noqa: E402

torch.set_num_threads(8)


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def summary(t, nsample=2048):
    """strided sample + float64 statistics of a tensor"""
    f = t.detach().double().flatten()
    step = max(1, f.numel() // nsample)
    return {'shape': list(t.shape), 'step': step, 'sample': f[::step][:nsample].float().numpy(),
            'mean': f.mean().item(), 'absmax': f.abs().max().item(), 'l2': f.norm().item()}


def put(store, prefix, s):
    store[prefix + '.sample'] = s['sample']
    store[prefix + '.meta'] = np.array([s['step'], s['mean'], s['absmax'], s['l2']], dtype=np.float64)
    store[prefix + '.shape'] = np.array(s['shape'], dtype=np.int64)


def param_checksum(module):
    return float(sum(p.detach().double().abs().sum().item() for p in module.state_dict().values() if p.is_floating_point()))


# ----------------------------------------------------------------------------------------------
# A. masks (bit-exact)
# ----------------------------------------------------------------------------------------------
def gen_masks():
    out = {'cases': []}
    shapes = recipes.tap_shapes(2)
    for seed in range(10):
        for L in range(1, 5):
            random.seed(seed)
            tt = random.choices([1, 2], [0.6, 0.4])
            masks = [torch.full(s, 0.5) for s in shapes]
            ref_tm.setMasksPart2(masks, torch.device('cpu'), L)
            nxt = random.random()
            case = {'seed': seed, 'L': L, 'train_type': tt, 'train_type_eq_1': bool(tt == 1), 'next_random': nxt,
                    'sha256': [sha(m) for m in masks], 'sum': [float(m.sum()) for m in masks]}
            out['cases'].append(case)
    # part 1 and isolate_layer
    p1 = []
    for L in range(1, 5):
        masks = [torch.full(s, 0.5) for s in shapes]
        ref_tm.setMasksPart1(masks, torch.device('cpu'), L)
        p1.append({'L': L, 'sha256': [sha(m) for m in masks]})
    out['part1'] = p1
    g = torch.Generator().manual_seed(7)
    feats = [torch.rand(s, generator=g) for s in recipes.tap_shapes(1)]
    iso = ref_tm.isolate_layer(feats, 2, torch.device('cpu'))
    out['isolate'] = {'seed': 7, 'selected': 2, 'sha256': [sha(m) for m in iso]}
    # setCroppedMask with non-default args on an odd shape
    m = ref_tm.setCroppedMask((33, 21), torch.ones(1, 2, 37, 53), 0.3, 0.6, 7)
    out['cropped'] = {'crop': [33, 21], 'shape': [1, 2, 37, 53], 'w1': 0.3, 'w2': 0.6, 'data': 7, 'sha256': sha(m),
                      'sum': float(m.sum())}
    with open(os.path.join(OUT, 'masks.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('masks:', len(out['cases']), 'cases')


# ----------------------------------------------------------------------------------------------
# B. module forwards
# ----------------------------------------------------------------------------------------------
def gen_modules():
    store = {}
    meta = {}
    # generators
    for gen_type in ['res', 'default']:
        for mode in ['train', 'eval']:
            seed = 11 if gen_type == 'res' else 12
            torch.manual_seed(seed)
            g = ref_gen.Generator(gen_type)
            g.init_weights()
            name = f'G_{gen_type}_{mode}'
            meta[name] = {'seed': seed, 'param_checksum': param_checksum(g)}
            z, feats, masks = recipes.generator_inputs(seed, B=2)
            if mode == 'eval':
                recipes.perturb_bn_buffers(g, seed)
                g.eval()
                with torch.no_grad():
                    y = g(z, feats, masks)
            else:
                g.train()
                y = g(z, feats, masks)
                store[name + '.rm'] = g.conv_block2.block[1].running_mean.detach().numpy().copy()
                store[name + '.rv'] = g.conv_block2.block[1].running_var.detach().numpy().copy()
                meta[name]['nbt'] = int(g.conv_block2.block[1].num_batches_tracked)
            put(store, name, summary(y))
            print(name, y.shape, float(y.abs().max()))
    # discriminators
    for norm in ['instance_norm', 'batch_norm', 'layer_norm']:
        for dis_type in ['res', 'default']:
            seed = 21
            torch.manual_seed(seed)
            d = ref_dis.Discriminator(norm, dis_type)
            d.init_weights()
            name = f'D_{norm}_{dis_type}'
            meta[name] = {'seed': seed, 'param_checksum': param_checksum(d)}
            x = recipes.image_batch(seed, 2) * 2 - 1
            d.train()
            y = d(x)
            store[name] = y.detach().numpy().reshape(-1)
            print(name, y.flatten().tolist())
    # feature-creation models
    sd = torch.load(os.path.join(REF, 'pre trained generators', 'features1_to_image'), weights_only=True)
    f1 = ref_fc.Features1ToImage()
    f1.load_state_dict(sd)
    f1.eval()
    g = torch.Generator().manual_seed(31)
    x = torch.rand(2, 64, 64, 64, generator=g) * 2.0
    with torch.no_grad():
        y = f1(x)
    put(store, 'F1', summary(y))
    meta['F1'] = {'seed': 31, 'param_checksum': param_checksum(f1)}
    print('F1', y.shape, float(y.abs().max()))
    for level in [2, 3, 4]:
        seed = 40 + level
        torch.manual_seed(seed)
        lu = ref_fc.LevelUpFeaturesGenerator(level)
        lu.init_weights()
        lu.train()
        shapes = recipes.tap_shapes(2)
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(shapes[level], generator=g)
        y = lu(x)
        put(store, f'LU{level}', summary(y))
        meta[f'LU{level}'] = {'seed': seed, 'param_checksum': param_checksum(lu)}
        for norm in ['instance_norm', 'batch_norm', 'layer_norm']:
            for dis_type in (['res', 'default'] if norm == 'instance_norm' else ['res']):
                torch.manual_seed(seed + 100)
                fd = ref_fc.FeaturesDiscriminator(norm, dis_type, level - 1)
                fd.init_weights()
                fd.train()
                xx = torch.rand(shapes[level - 1], generator=g)
                yy = fd(xx)
                nm = f'FD{level - 1}_{norm}_{dis_type}'
                store[nm] = yy.detach().numpy().reshape(-1)
                meta[nm] = {'seed': seed + 100, 'param_checksum': param_checksum(fd)}
                print(nm, yy.flatten().tolist())
    np.savez_compressed(os.path.join(OUT, 'modules.npz'), **store)
    with open(os.path.join(OUT, 'modules.json'), 'w') as f:
        json.dump(meta, f, indent=1)


# ----------------------------------------------------------------------------------------------
# C. train steps (reference train_model.train_discriminator / train_generator)
# ----------------------------------------------------------------------------------------------
class Recorder:
    """records the tensors torch.randn / torch.rand hand to the reference's step functions"""

    def __init__(self):
        self.randn, self.rand = [], []
        self._randn, self._rand = torch.randn, torch.rand

    def __enter__(self):
        def randn(*a, **k):
            t = self._randn(*a, **k)
            self.randn.append(t.clone())
            return t

        def rand(*a, **k):
            t = self._rand(*a, **k)
            self.rand.append(t.clone())
            return t
        torch.randn, torch.rand = randn, rand
        return self

    def __exit__(self, *exc):
        torch.randn, torch.rand = self._randn, self._rand


def tensor_norms(sd_or_named):
    return {k: float(v.detach().double().norm()) for k, v in sd_or_named}


def gen_steps():
    store, meta = {}, {}
    configs = [('res', 'instance_norm', False), ('res', 'instance_norm', True), ('default', 'batch_norm', False),
               ('default', 'layer_norm', False), ('res', 'batch_norm', False)]
    for gen_type, norm, div in configs:
        name = f'steps_{gen_type}_{norm}_{"div" if div else "nodiv"}'
        seed = 51
        torch.manual_seed(seed)
        G = ref_gen.Generator(gen_type); G.init_weights()
        D = ref_dis.Discriminator(norm, gen_type); D.init_weights()
        model = recipes.make_resnet(seed)
        C = ref_clf.Classifier(model); C.eval()
        ref_tm.args = Namespace(use_diversity_loss=div, gradient_penalty_weight=10.0)
        d_loss_fn, g_loss_fn = ref_tm.get_wgan_losses_fn()
        crit = torch.nn.L1Loss()
        g_opt = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
        d_opt = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
        B = 2
        images = recipes.image_batch(seed, B)
        images_d = (images - 0.5) / 0.5
        images_c = recipes.normalize_clf(images)
        with torch.no_grad():
            _, features = C(images_c)
        features = [f.detach() for f in features]
        random.seed(seed)
        L = 3
        masks = [f.clone() for f in features]
        ref_tm.setMasksPart2(masks, torch.device('cpu'), L)
        normalizer = recipes.normalize_clf
        log = {}
        before = {('G', k): v.detach().clone() for k, v in G.named_parameters()}
        before.update({('D', k): v.detach().clone() for k, v in D.named_parameters()})
        torch.manual_seed(seed + 1)
        with Recorder() as rec:
            d1 = ref_tm.train_discriminator(G, D, d_loss_fn, d_opt, images_d, features, masks)
            g1 = ref_tm.train_generator(G, D, g_loss_fn, g_opt, B, features, crit, L, C, normalizer, crit, crit, masks, [2])
            d2 = ref_tm.train_discriminator(G, D, d_loss_fn, d_opt, images_d, features, masks)
        for i, t in enumerate(rec.randn):
            store[f'{name}.randn{i}'] = t.numpy()
        for i, t in enumerate(rec.rand):
            store[f'{name}.rand{i}'] = t.numpy()
        for tag, dct in (('d1', d1), ('g1', g1), ('d2', d2)):
            for k, v in dct.items():
                log[f'{tag}.{k}'] = float(v)
        # strided samples of what the three optimizer steps did to every parameter tensor (after - before): a check of the
        # update itself, which tensor norms cannot see (a step of lr = 2e-4 moves a norm by less than any usable tolerance)
        for (which, k), v0 in before.items():
            v1 = dict((G if which == 'G' else D).named_parameters())[k].detach()
            d = (v1.double() - v0.double()).flatten()
            step = max(1, d.numel() // 256)
            store[f'{name}.delta.{which}.{k}'] = d[::step][:256].float().numpy()
        meta[name] = {'seed': seed, 'L': L, 'losses': log,
                      'n_randn': len(rec.randn), 'n_rand': len(rec.rand),
                      'G_param_checksum': param_checksum(G), 'D_param_checksum': param_checksum(D),
                      'G_norms': tensor_norms(G.state_dict().items()), 'D_norms': tensor_norms(D.state_dict().items()),
                      'D_grad_norms': tensor_norms((k, p.grad) for k, p in D.named_parameters()),
                      'G_grad_norms': tensor_norms((k, p.grad) for k, p in G.named_parameters())}
        print(name, log)
    np.savez_compressed(os.path.join(OUT, 'steps.npz'), **store)
    with open(os.path.join(OUT, 'steps.json'), 'w') as f:
        json.dump(meta, f, indent=1)


def gen_feature_steps():
    """train_feature_creation_gen.py train_discriminator / train_generator for block f4 -> f3."""
    store, meta = {}, {}
    level = 4
    for norm in ['instance_norm']:
        name = f'fsteps_l{level}_{norm}'
        seed = 61
        torch.manual_seed(seed)
        FG = ref_fc.LevelUpFeaturesGenerator(level); FG.init_weights()
        FD = ref_fc.FeaturesDiscriminator(norm, 'res', level - 1); FD.init_weights()
        model = recipes.make_resnet(seed)
        C = ref_clf.Classifier(model); C.eval()
        ref_tf.args = Namespace(gradient_penalty_weight=10.0)
        d_loss_fn, g_loss_fn = ref_tm.get_wgan_losses_fn()
        crit = torch.nn.L1Loss()
        g_opt = torch.optim.Adam(FG.parameters(), lr=2e-4, betas=(0.5, 0.999))
        d_opt = torch.optim.Adam(FD.parameters(), lr=2e-4, betas=(0.5, 0.999))
        images = recipes.image_batch(seed, 2)
        with torch.no_grad():
            _, features = C(recipes.normalize_clf(images))
        features = [f.detach() for f in features]
        torch.manual_seed(seed + 1)
        with Recorder() as rec:
            d1 = ref_tf.train_discriminator(FG, FD, d_loss_fn, d_opt, features, level)
            g1 = ref_tf.train_generator(FG, FD, C, g_opt, features, level, g_loss_fn, crit, crit)
        for i, t in enumerate(rec.rand):
            store[f'{name}.rand{i}'] = t.numpy()
        log = {f'd1.{k}': float(v) for k, v in d1.items()}
        log.update({f'g1.{k}': float(v) for k, v in g1.items()})
        meta[name] = {'seed': seed, 'losses': log, 'G_norms': tensor_norms(FG.state_dict().items()),
                      'D_norms': tensor_norms(FD.state_dict().items())}
        print(name, log)
    np.savez_compressed(os.path.join(OUT, 'fsteps.npz'), **store)
    with open(os.path.join(OUT, 'fsteps.json'), 'w') as f:
        json.dump(meta, f, indent=1)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['masks', 'modules', 'steps', 'fsteps']
    if 'masks' in which:
        gen_masks()
    if 'modules' in which:
        gen_modules()
    if 'steps' in which:
        gen_steps()
    if 'fsteps' in which:
        gen_feature_steps()
