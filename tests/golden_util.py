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
