"""Bisect gradient error sources: swap op families for torch's own GPU ops (diagnostic only)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, torch.nn.functional as F
import test_steps_gpu as T
from semantic_pyramid_for_image_generation_amd import ops
dev = torch.device('cuda:0')
orig = {k: getattr(ops, k) for k in ['norm_act', 'conv2d', 'act', 'avg_pool2', 'gp_loss', 'mean']}

CNT = {'norm': 0, 'conv': 0}
def t_norm_act(x, gamma, beta, kind, act_kind=0, slope=0.0, eps=1e-5, bn_state=None):
    CNT['norm'] += 1
    if kind == ops.NORM_INSTANCE: y = F.instance_norm(x, None, None, gamma, beta, True, 0.1, eps)
    elif kind == ops.NORM_LAYER: y = F.group_norm(x, 1, gamma, beta, eps)
    else: y = F.batch_norm(x, None, None, gamma, beta, True, 0.1, eps)
    if act_kind == ops.ACT_LRELU: y = F.leaky_relu(y, slope)
    elif act_kind == ops.ACT_RELU: y = F.relu(y)
    return y
def t_conv2d(x, w, bias=None, stride=1, padding=0, upsample=False):
    CNT['conv'] += 1
    if upsample: x = F.interpolate(x, scale_factor=2, mode='nearest')
    return F.conv2d(x, w, bias, stride, padding)
def t_act(x, k, slope=0.0):
    return F.leaky_relu(x, slope) if k == ops.ACT_LRELU else (F.relu(x) if k == ops.ACT_RELU else torch.tanh(x))
def t_gp(grad):
    return ((grad.reshape(grad.size(0), -1).norm(p=2, dim=1) - 1) ** 2).mean()
variants = {'hip': {}, 'torch_norm': {'norm_act': t_norm_act}, 'torch_conv': {'conv2d': t_conv2d},
            'torch_act_pool_gp': {'act': t_act, 'avg_pool2': lambda x: F.avg_pool2d(x, 2), 'gp_loss': t_gp,
                                  'mean': lambda x, s=1.0: x.mean() * s},
            'torch_all': {'norm_act': t_norm_act, 'conv2d': t_conv2d, 'act': t_act, 'avg_pool2': lambda x: F.avg_pool2d(x, 2), 'gp_loss': t_gp, 'mean': lambda x, s=1.0: x.mean() * s}}
torch.backends.cudnn.allow_tf32 = False
case = ('res', 'instance_norm', True)
ref = None
keys = ['first_conv.weight', 'conv_block1.conv_block1.block.0.weight', 'conv_block2.conv_block1.block.0.weight',
        'conv_block2.conv_block1.block.1.bias', 'conv_block3.conv_block1.block.0.weight', 'conv_block4.conv_block1.block.0.weight']
for name, patch in variants.items():
    for k, v in orig.items(): setattr(ops, k, v)
    for k, v in patch.items(): setattr(ops, k, v)
    tm, G, D, C, tr, images, feats, masks, z, z2, alpha = T._setup(dev, *case, 71)
    d_loss_fn, _ = tm.get_wgan_losses_fn()
    d_opt = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    if ref is None:
        tr.d_step(T._d((images - 0.5) / 0.5), T._d(feats), T._d(masks), T._d(z), T._d(alpha))
        ref = {k: tr.d[k].grad.clone() for k in tr.d_params}
    # the generator always runs on the HIP ops (restore for its forward)
    for k, v in orig.items(): setattr(ops, k, v)
    with torch.no_grad():
        fake = G(z.to(dev), [f.to(dev) for f in feats], [m.to(dev) for m in masks])
    for k, v in patch.items(): setattr(ops, k, v)
    real = ((images - 0.5) / 0.5).to(dev)
    r, f = D(real), D(fake)
    rl, fl = d_loss_fn(r, f)
    gp = tm.gradient_penalty(D, real, fake, alpha.to(dev))
    loss = (rl + fl) + gp * 10.0
    D.zero_grad(); loss.backward()
    print(dict(CNT), float(gp), float(rl + fl)); CNT['norm'] = CNT['conv'] = 0
    print('%-18s' % name, ' '.join('%.1e' % (float((D.get_parameter(k).grad.cpu().double() - ref[k]).abs().max()) / float(ref[k].abs().max())) for k in keys), flush=True)
