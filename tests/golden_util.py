"""Helpers shared by the golden-fixture tests."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def check_summary(store, prefix, t, atol):
    """compare tensor `t` with a fixture written by gen_golden.put()"""
    step, mean, absmax, l2 = store[prefix + '.meta']
    assert list(t.shape) == list(store[prefix + '.shape'])
    f = t.detach().double().cpu().flatten()
    sample = f[::int(step)][:len(store[prefix + '.sample'])]
    err = float((sample - torch.from_numpy(store[prefix + '.sample']).double()).abs().max())
    assert err <= atol, f'{prefix}: sample max-abs err {err}'
    assert abs(f.mean().item() - mean) <= atol, f'{prefix}: mean'
    assert abs(f.norm().item() - l2) <= atol * max(1.0, l2), f'{prefix}: l2'
    return err


def param_checksum(module_or_sd):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, 'state_dict') else module_or_sd
    return float(sum(p.detach().double().abs().sum().item() for p in sd.values() if p.is_floating_point()))


def cpu_state(module):
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def check_deltas(store, name, which, before, after, grad_norms, lr=2e-4, med_tol=0.1, frac_big=0.2):
    """Compare what the optimizer steps did to every parameter tensor (after - before, strided samples) with the
    reference's own deltas (gen_golden.gen_steps).  Adam's update is ~ lr * sign(g) per step, so an element whose gradient
    is at round-off distance from zero may move the other way (a difference of up to 2 lr per step) in ANY implementation,
    and the second critic step sits behind two updates whose trajectories have diverged at the percent level.  Per tensor:
    the MEDIAN |delta - reference delta| must be below med_tol * lr, and at most `frac_big` of the sampled elements may
    differ by more than 0.5 lr.  (With the optimizer switched off, a wrong lr or wrong betas the median difference is of
    order lr.)  Tensors whose reference gradient is structurally zero (a conv bias in front of a norm layer: pure
    round-off) are skipped.  Returns (worst median / lr, worst fraction)."""
    gmax = max(grad_norms.values())
    worst_med = worst_frac = 0.0
    checked = 0
    for k, gn in grad_norms.items():
        if gn <= 1e-4 * gmax:
            continue
        ref = torch.from_numpy(store[f'{name}.delta.{which}.{k}']).double()
        d = (after[k].detach().double().cpu() - before[k].detach().double().cpu()).flatten()
        step = max(1, d.numel() // 256)
        d = d[::step][:256]
        assert d.numel() == ref.numel(), (which, k)
        assert float(ref.abs().max()) > 0.5 * lr, (which, k, 'reference delta is empty?')
        diff = (d - ref).abs() / lr
        med, frac = float(diff.median()), float((diff > 0.5).double().mean())
        assert med <= med_tol, (which, k, 'median', med)
        assert frac <= max(frac_big, 1.0 / d.numel()), (which, k, 'fraction', frac)
        worst_med, worst_frac = max(worst_med, med), max(worst_frac, frac)
        checked += 1
    assert checked >= 10, checked
    return worst_med, worst_frac
