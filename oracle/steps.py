"""WGAN-GP train steps restated on torch-CPU autograd (train_model.py:35-145,212-222 and
train_feature_creation_gen.py:30-79).  Random draws (z, z2, alpha) are injected by the caller so that the
HIP path, this oracle and the reference can be compared on identical inputs."""
import torch
import torch.nn.functional as F

from . import nets

CLF_MEAN = (0.485, 0.456, 0.406)
CLF_STD = (0.229, 0.224, 0.225)


def normalize(x, mean, std):
    """transforms.Normalize — train_model.py:240-245 (out-of-place sub then true division)."""
    m = torch.tensor(mean, dtype=x.dtype).view(1, -1, 1, 1)
    s = torch.tensor(std, dtype=x.dtype).view(1, -1, 1, 1)
    return (x - m) / s


def gradient_penalty(d_fwd, real, fake, alpha):
    """sample_line + gradient_penalty + one_mean_gp — train_model.py:51-76."""
    x = (real + alpha * (fake - real)).detach()
    x.requires_grad_(True)
    pred = d_fwd(x)
    grad = torch.autograd.grad(pred, x, grad_outputs=torch.ones_like(pred), create_graph=True)[0]
    n = grad.reshape(grad.size(0), -1).norm(p=2, dim=1)
    return ((n - 1) ** 2).mean()


def features_loss(features, masks, train_type_is_1, L, fake_features):
    """getLossByTrainType — train_model.py:212-222."""
    if train_type_is_1:
        loss = F.l1_loss(features[L], fake_features[L])
    else:
        loss = F.l1_loss(features[1] * masks[1], fake_features[1] * masks[1])
        for i in range(2, L + 1):
            loss = loss + F.l1_loss(features[i] * masks[i], fake_features[i] * masks[i])
    return loss * (1 / L) * 200


class Trainer:
    """Holds generator / discriminator parameters (leaf tensors) + Adam(lr, betas=(.5,.999)) states and
    performs the reference's D and G steps (train_model.py:83-145, :273-274)."""

    def __init__(self, g_sd, d_sd, clf_sd, gen_type='res', d_norm='instance_norm', lr=2e-4, gp_weight=10.0,
                 use_diversity=False):
        self.gen_type, self.d_norm, self.gp_weight, self.use_diversity = gen_type, d_norm, gp_weight, use_diversity
        self.g = {k: v.clone() for k, v in g_sd.items()}
        self.d = {k: v.clone() for k, v in d_sd.items()}
        self.clf = None if clf_sd is None else {k: v.clone() for k, v in clf_sd.items()}
        self.g_params = [k for k, v in self.g.items() if v.is_floating_point() and 'running_' not in k]
        self.d_params = [k for k, v in self.d.items() if v.is_floating_point() and 'running_' not in k]
        for k in self.g_params:
            self.g[k].requires_grad_(True)
        for k in self.d_params:
            self.d[k].requires_grad_(True)
        self.g_opt = torch.optim.Adam([self.g[k] for k in self.g_params], lr=lr, betas=(0.5, 0.999))
        self.d_opt = torch.optim.Adam([self.d[k] for k in self.d_params], lr=lr, betas=(0.5, 0.999))

    def G(self, z, features, masks, training=True):
        return nets.generator_forward(self.g, self.gen_type, z, features, masks, training)

    def D(self, x):
        return nets.discriminator_forward(self.d, self.d_norm, self.gen_type, x, True)

    def d_step(self, real, features, masks, z, alpha):
        """train_discriminator — train_model.py:122-145."""
        fake = self.G(z, features, masks).detach()
        r, f = self.D(real), self.D(fake)
        r_loss, f_loss = -r.mean(), f.mean()
        gp = gradient_penalty(self.D, real, fake, alpha)
        loss = (r_loss + f_loss) + gp * self.gp_weight
        self.d_opt.zero_grad()
        loss.backward()
        self.d_opt.step()
        return {'d_loss': (r_loss + f_loss).detach(), 'gp': gp.detach()}

    def g_step(self, features, masks, L, z, z2=None, train_type_is_1=False, epsilon=10e-4):
        """train_generator — train_model.py:83-119."""
        fake = self.G(z, features, masks)
        g_loss = -self.D(fake).mean()
        _, fake_features = nets.classifier_forward(self.clf, normalize(fake, CLF_MEAN, CLF_STD))
        f_loss = features_loss(features, masks, train_type_is_1, L, fake_features)
        total = g_loss + f_loss
        out = {'g_loss': g_loss.detach(), 'features_loss': f_loss.detach(), 'total_loss': total.detach()}
        if self.use_diversity:
            fake2 = self.G(z2, features, masks)
            div = F.l1_loss(z, z2) / (F.l1_loss(fake, fake2) + epsilon)
            total = total + div
            out['diversity loss'] = div.detach()
            out['total loss'] = total.detach()
        self.g_opt.zero_grad()
        for k in self.d_params:                 # the reference leaves stale D grads behind; they are zeroed
            self.d[k].grad = None               # by the next D step and never used
        total.backward()
        self.g_opt.step()
        return out


class FeatureTrainer:
    """train_feature_creation_gen.py:30-79 for one LevelUp block f_L -> f_{L-1}."""

    def __init__(self, fg_sd, d_sd, clf_sd, level, d_norm='instance_norm', dis_type='res', lr=2e-4, gp_weight=10.0):
        self.level, self.d_norm, self.dis_type, self.gp_weight = level, d_norm, dis_type, gp_weight
        self.g = {k: v.clone() for k, v in fg_sd.items()}
        self.d = {k: v.clone() for k, v in d_sd.items()}
        self.clf = {k: v.clone() for k, v in clf_sd.items()}
        self.g_params = [k for k, v in self.g.items() if v.is_floating_point() and 'running_' not in k]
        self.d_params = [k for k, v in self.d.items() if v.is_floating_point() and 'running_' not in k]
        for k in self.g_params:
            self.g[k].requires_grad_(True)
        for k in self.d_params:
            self.d[k].requires_grad_(True)
        self.g_opt = torch.optim.Adam([self.g[k] for k in self.g_params], lr=lr, betas=(0.5, 0.999))
        self.d_opt = torch.optim.Adam([self.d[k] for k in self.d_params], lr=lr, betas=(0.5, 0.999))

    def G(self, f):
        return nets.level_up_forward(self.g, f, True)

    def D(self, x):
        return nets.features_discriminator_forward(self.d, self.d_norm, self.dis_type, self.level - 1, x, True)

    def d_step(self, features, alpha):
        L = self.level
        fake = self.G(features[L]).detach()
        r, f = self.D(features[L - 1]), self.D(fake)
        r_loss, f_loss = -r.mean(), f.mean()
        gp = gradient_penalty(self.D, features[L - 1], fake, alpha)
        loss = (r_loss + f_loss) + gp * self.gp_weight
        self.d_opt.zero_grad()
        loss.backward()
        self.d_opt.step()
        return {'d_loss': (r_loss + f_loss).detach(), 'gp': gp.detach()}

    def g_step(self, features):
        L = self.level
        created = self.G(features[L])
        g_loss = -self.D(created).mean()
        stride = 2
        recon = nets.resnet_stage(self.clf, 'layer%d' % L, created, stride)
        f_loss = F.l1_loss(features[L], recon)
        nl_loss = F.l1_loss(created, features[L - 1])
        total = g_loss + f_loss + 0.2 * nl_loss
        self.g_opt.zero_grad()
        total.backward()
        self.g_opt.step()
        return {'generator_loss': g_loss.detach(), 'features_loss': f_loss.detach(), 'next_level_loss': nl_loss.detach()}
