"""optim.Adam (one HIP launch per step) against torch.optim.Adam: same update within fp32 rounding, same state layout.
Tolerance: 2e-7 absolute on parameters of magnitude <= 1 after 6 steps (both sides compute in fp32 with the same operation
order; only FMA contraction may differ)."""
import copy

import pytest
import torch


def _tensors(gen, dev):
    # odd sizes, an unaligned view start, > 64 tensors (two launches), one multi-chunk tensor with a ragged tail
    shapes = [(7,), (64, 3, 3, 3), (1,), (4097,), (128, 64, 4, 4), (5, 5)] + [(33,)] * 64 + [(3 * 4096 + 5,)]
    out = []
    for sh in shapes:
        out.append(torch.randn(*sh, generator=gen) * 0.05)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("betas", [(0.5, 0.999), (0.9, 0.99)])
def test_adam_matches_torch(dev, betas):
    from semantic_pyramid_for_image_generation_amd import optim
    gen = torch.Generator().manual_seed(3)
    init = _tensors(gen, dev)
    pa = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    pb = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    # one parameter lives at an address that is not 16-byte aligned
    base = torch.zeros(1001, device=dev)
    with torch.no_grad():
        base[1:].copy_(torch.randn(1000, generator=gen) * 0.05)
    pa.append(torch.nn.Parameter(base[1:].clone())); pb.append(torch.nn.Parameter(base.clone()[1:]))
    assert pb[-1].data_ptr() % 16 != 0
    a = optim.Adam(pa, lr=2e-4, betas=betas)
    b = torch.optim.Adam(pb, lr=2e-4, betas=betas)
    for step in range(6):
        for i, (x, y) in enumerate(zip(pa, pb)):
            if step == 2 and i == 3:
                x.grad = None; y.grad = None          # a parameter that skips a step keeps its own step count
                continue
            g = torch.randn(x.shape, generator=gen) * (0.0 if (i == 2 and step < 2) else 1e-2)
            x.grad = g.to(dev); y.grad = g.to(dev).clone()
        v0 = pa[0]._version
        a.step(); b.step()
        assert pa[0]._version > v0                     # caches keyed on tensor versions see the update
    for i, (x, y) in enumerate(zip(pa, pb)):
        assert float((x - y).detach().abs().max()) <= 2e-7, i
        sa, sb = a.state[x], b.state[y]
        assert float(sa['step']) == float(sb['step'])
        assert float((sa['exp_avg'] - sb['exp_avg']).abs().max()) <= 1e-9
        assert float((sa['exp_avg_sq'] - sb['exp_avg_sq']).abs().max()) <= 1e-11
    # state interoperates with torch.optim.Adam in both directions
    # (deep copies: load_state_dict keeps the very `step` tensor objects it is given)
    b.load_state_dict(copy.deepcopy(a.state_dict()))
    a.load_state_dict(copy.deepcopy(b.state_dict()))
    for x, y in zip(pa, pb):
        g = torch.full(x.shape, 1e-3, device=dev)
        x.grad = g; y.grad = g.clone()
    a.step(); b.step()
    for x, y in zip(pa, pb):
        assert float((x - y).detach().abs().max()) <= 2e-7


@pytest.mark.gpu
def test_adam_grad_scale_and_zeroing(dev):
    from semantic_pyramid_for_image_generation_amd import optim
    gen = torch.Generator().manual_seed(4)
    w = torch.randn(5000, generator=gen) * 0.05
    g = torch.randn(5000, generator=gen)
    x = torch.nn.Parameter(w.clone().to(dev)); y = torch.nn.Parameter(w.clone().to(dev))
    a = optim.Adam([x], lr=1e-3, betas=(0.5, 0.999), grad_scale=0.25, zero_grads=True)
    b = torch.optim.Adam([y], lr=1e-3, betas=(0.5, 0.999))
    x.grad = g.to(dev); y.grad = g.to(dev) * 0.25
    a.step(); b.step()
    assert float((x - y).detach().abs().max()) <= 2e-7
    assert float(x.grad.abs().max()) == 0.0


def test_adam_interface_matches_torch():
    """constructor checks, param_group keys and the refusal of options without a kernel (no GPU needed)"""
    from semantic_pyramid_for_image_generation_amd import optim
    p = [torch.nn.Parameter(torch.zeros(3))]
    a = optim.Adam(p, lr=2e-4, betas=(0.5, 0.999))
    b = torch.optim.Adam(p, lr=2e-4, betas=(0.5, 0.999))
    ka, kb = set(a.param_groups[0]), set(b.param_groups[0])
    assert {'lr', 'betas', 'eps', 'weight_decay', 'amsgrad', 'params'} <= ka
    assert kb - ka <= {'decoupled_weight_decay'}
    with pytest.raises(NotImplementedError):
        optim.Adam(p, weight_decay=0.1)
    with pytest.raises(NotImplementedError):
        optim.Adam(p, amsgrad=True)
    with pytest.raises(ValueError):
        optim.Adam(p, lr=-1.0)
    # CPU parameters are refused at step time: there is no CPU fallback
    p[0].grad = torch.ones(3)
    from semantic_pyramid_for_image_generation_amd._lib import SempyrError
    with pytest.raises(SempyrError):
        a.step()


@pytest.mark.gpu
def test_adam_step_refreshes_packed_weights(dev):
    """The optimizer step re-packs every cached k-major weight copy in one launch: the next convolution finds a valid cache
    entry (no pack launch of its own) and computes with the UPDATED weights."""
    import torch.nn.functional as F
    from semantic_pyramid_for_image_generation_amd import ops, optim
    gen = torch.Generator().manual_seed(7)
    w1 = torch.nn.Parameter((torch.randn(16, 8, 3, 3, generator=gen) * 0.1).to(dev))
    w2 = torch.nn.Parameter((torch.randn(8, 16, 4, 4, generator=gen) * 0.1).to(dev))
    x = torch.randn(2, 8, 12, 12, generator=gen).to(dev)
    opt = optim.Adam([w1, w2], lr=1e-2, betas=(0.5, 0.999))
    for _ in range(2):
        y = ops.conv2d(ops.conv2d(x, w1, None, 1, 1), w2, None, 2, 1)
        opt.zero_grad()
        y.square().mean().backward()          # forward + input-gradient layouts of both weights are now cached
        opt.step()
        for w in (w1, w2):
            assert len(w._sempyr_pack) >= 1
            for hit in w._sempyr_pack.values():
                assert hit[0] == w._version
    with torch.no_grad():
        got = ops.conv2d(ops.conv2d(x, w1, None, 1, 1), w2, None, 2, 1)
        want = F.conv2d(F.conv2d(x.cpu(), w1.detach().cpu(), None, 1, 1), w2.detach().cpu(), None, 2, 1)
    assert float((got.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
