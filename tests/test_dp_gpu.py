"""The data-parallel leg through the PRODUCT: two real processes (tests/dp_worker.py under torch.distributed.run) drive
`TrainLoop` iterations -- classifier taps, masks, critic step, generator step, with broadcast_module / flatten_grads /
sync_gradients / optim.Adam(grad_scale, zero_grads) exactly as `train_model._main` and `bench.py` use them.

  * test_dp_nccl_two_gpus: backend nccl (= RCCL over xGMI), one GPU per rank; skipped on a box with one GPU.
  * test_dp_two_ranks_one_gpu_gloo: the same worker with both ranks on the one card of the test box and gloo moving the flat
    gradient buffer through the host (RCCL refuses two ranks on one device): everything but the transport is the product code.

Both assert: parameters and BatchNorm buffers bit-identical across ranks after every iteration, flat buffers
zero and still attached, and the parameters after the first iteration (critic step + generator step) equal to the oracle
= two independent fp64 CPU replicas on the ranks' data whose gradients are averaged before Adam."""
import os
import random
import socket
import subprocess
import sys

import pytest
import torch

from oracle import masks as omasks
from oracle import nets, recipes, steps

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 41          # dp_worker.SEED


def _launch(tmp_path, env_extra, iters=2, B=2, world=2):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env['OMP_NUM_THREADS'] = '4'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(HERE, 'dp_worker.py'), str(tmp_path), str(iters), str(B)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    return [torch.load(os.path.join(tmp_path, f'rank{k}.pt'), weights_only=True) for k in range(world)]


def _replica_oracle(ranks, gen_type='res', norm='instance_norm'):
    """Two fp64 CPU replicas (oracle/steps.py) on the ranks' recorded data: critic step then generator step, every replica
    applying the gradient averaged over the replicas.  Returns the parameters after the two steps."""
    csd = {k: v.detach().clone() for k, v in recipes.make_resnet(SEED).state_dict().items()}
    f64 = lambda sd: {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    d64 = lambda t: [x.double() for x in t] if isinstance(t, list) else t.double()
    rng = random.Random(0)          # every rank seeds Python's `random` with 0: one layer / crop draw per global step
    is1, L = omasks.draw_iteration(rng, 0.6)
    crop = (rng.randint(20, 40), rng.randint(20, 40))
    data, reps = [], []
    for rk in ranks:
        images = rk['images']
        assert torch.equal(images, recipes.image_batch(SEED + 10 * rk['rank'], images.shape[0]))
        with torch.no_grad():
            _, feats = nets.classifier_forward(csd, recipes.normalize_clf(images))
        ms, _ = omasks.masks_part2([tuple(f.shape) for f in feats], L, _Fixed(crop))
        d0 = rk['draws'][0]
        assert d0['L'] == L and d0['key'][0] is True
        data.append(dict(images_d=(images - 0.5) / 0.5, feats=feats, masks=[torch.from_numpy(m) for m in ms],
                         z=d0['z_d'], alpha=d0['alpha'], zg=d0['z_g']))
        reps.append(steps.Trainer(f64(rk['init']['G']), f64(rk['init']['D']), f64(csd), gen_type, norm))
    for which in ('d', 'g'):
        grads = []
        for r, dt in zip(reps, data):
            if which == 'd':
                fake = r.G(d64(dt['z']), d64(dt['feats']), d64(dt['masks'])).detach()
                rl, fl = -r.D(d64(dt['images_d'])).mean(), r.D(fake).mean()
                loss = (rl + fl) + steps.gradient_penalty(r.D, d64(dt['images_d']), fake, d64(dt['alpha'])) * 10.0
                params = [r.d[k] for k in r.d_params]
            else:
                fake = r.G(d64(dt['zg']), d64(dt['feats']), d64(dt['masks']))
                _, ff = nets.classifier_forward(r.clf, steps.normalize(fake, steps.CLF_MEAN, steps.CLF_STD))
                loss = -r.D(fake).mean() + steps.features_loss(d64(dt['feats']), d64(dt['masks']), False, L, ff)
                params = [r.g[k] for k in r.g_params]
            grads.append(torch.autograd.grad(loss, params, allow_unused=True))
        for r in reps:
            params = [r.d[k] for k in r.d_params] if which == 'd' else [r.g[k] for k in r.g_params]
            for i, p in enumerate(params):
                p.grad = None if grads[0][i] is None else sum(g[i] for g in grads) / len(grads)
            (r.d_opt if which == 'd' else r.g_opt).step()
    return ({k: reps[0].d[k].detach() for k in reps[0].d_params}, {k: reps[0].g[k].detach() for k in reps[0].g_params})


class _Fixed:
    """stands for the rng of masks_part2: hands out a crop that was already drawn"""

    def __init__(self, vals):
        self.vals = list(vals)

    def randint(self, a, b):
        return self.vals.pop(0)


def _check(ranks):
    r0, r1 = ranks
    assert r0['world'] == 2 and {r0['rank'], r1['rank']} == {0, 1}
    assert not torch.equal(r0['images'], r1['images']) and not torch.equal(r0['draws'][0]['z_d'], r1['draws'][0]['z_d'])
    for rk in ranks:
        assert rk['flat_zero'] and rk['views_ok']
        assert len(rk['logged']) == len(rk['snaps'])
    # same layer / crop on every rank, every iteration
    for a, b in zip(r0['draws'], r1['draws']):
        assert a['key'] == b['key'] and a['L'] == b['L'] and a['specs'] == b['specs']
    # replicas stay bit-identical: parameters after every iteration (BatchNorm running statistics are per replica)
    for it, (sa, sb) in enumerate(zip(r0['snaps'], r1['snaps'])):
        for m in ('G', 'D'):
            for k, v in sa[m].items():
                if 'running_' in k or 'num_batches' in k:
                    continue
                assert torch.equal(v, sb[m][k]), (it, m, k)
    # the broadcast repaired rank 1: the update from the recorded initial state is an Adam step, not the 0.25 it was moved by
    lr = 2e-4
    for m in ('G', 'D'):
        for k, v in r1['snaps'][0][m].items():
            if v.is_floating_point() and 'running_' not in k:
                assert float((v - r1['init'][m][k]).abs().max()) <= 2.5 * lr, (m, k)
    # against the averaged-gradient oracle (the update is ~ lr * sign(g): compare updates, allow sign flips of round-off gradients)
    want_d, want_g = _replica_oracle(ranks)
    for name, want in (('D', want_d), ('G', want_g)):
        checked = 0
        for k, w in want.items():
            init = r0['init'][name][k].double()
            ref_delta = w.double() - init
            if float(ref_delta.abs().max()) < 0.5 * lr:
                continue
            delta = r0['snaps'][0][name][k].double() - init
            diff = (delta - ref_delta).abs() / lr
            assert float(diff.median()) <= 0.05, (name, k, float(diff.median()))
            assert float((diff > 0.5).double().mean()) <= 0.1, (name, k, float((diff > 0.5).double().mean()))
            checked += 1
        assert checked >= 10


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_dp_nccl_two_gpus(tmp_path):
    ranks = _launch(tmp_path, {'SEMPYR_DIST_BACKEND': 'nccl'})
    assert all(r['backend'] == 'nccl' for r in ranks) and ranks[0]['device'] != ranks[1]['device']
    _check(ranks)


def _check_buckets(ranks, plain):
    """The critic's gradient buckets (parallel.FlatGrads): two of them at the default 8 MB, fired from inside the backward pass
    from the third critic step on (two steps to learn the contribution counts); the result is bit-identical to the single
    collective after the backward pass."""
    for rk in ranks:
        st = rk['bucket_stats']
        assert rk['n_buckets'] == 2 and st['steps'] == len(rk['snaps'])
        assert st['early'] == 2 * (st['steps'] - 2) and st['late'] == 2 * 2, st
    for it, (sa, sb) in enumerate(zip(ranks[0]['snaps'], plain[0]['snaps'])):
        for m in ('G', 'D'):
            for k, v in sa[m].items():
                assert torch.equal(v, sb[m][k]), (it, m, k)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_dp_nccl_buckets_overlap_backward(tmp_path):
    a = tmp_path / 'a'; b = tmp_path / 'b'
    a.mkdir(); b.mkdir()
    ranks = _launch(a, {'SEMPYR_DIST_BACKEND': 'nccl'}, iters=4)
    plain = _launch(b, {'SEMPYR_DIST_BACKEND': 'nccl', 'DP_BUCKET_BYTES': '0'}, iters=4)
    _check_buckets(ranks, plain)


def test_dp_two_ranks_one_gpu_gloo(tmp_path, dev):
    a = tmp_path / 'a'; b = tmp_path / 'b'
    a.mkdir(); b.mkdir()
    ranks = _launch(a, {'SEMPYR_DIST_BACKEND': 'gloo', 'SEMPYR_SHARE_GPU': '1'}, iters=4)
    assert all(r['backend'] == 'gloo' for r in ranks)
    _check(ranks)
    plain = _launch(b, {'SEMPYR_DIST_BACKEND': 'gloo', 'SEMPYR_SHARE_GPU': '1', 'DP_BUCKET_BYTES': '0'}, iters=4)
    assert plain[0]['n_buckets'] == 0
    _check_buckets(ranks, plain)
