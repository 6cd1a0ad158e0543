"""Data-parallel path on CPU: world_size 2 over gloo.  Checks the flat-buffer gradient all-reduce (mean over ranks,
missing grads treated as zeros) and that parameters stay identical on both ranks after an Adam step —
the N>1 oracle of SURVEY.md §8e: N independent replicas with averaged gradients."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from semantic_pyramid_for_image_generation_amd import parallel
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2), torch.nn.BatchNorm1d(3))
    if rank == 1:                                    # diverge, then broadcast must repair it
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
            net[3].running_mean.add_(2.0); net[3].num_batches_tracked.add_(7)      # buffers (float and int64) too
    parallel.broadcast_module(net)
    assert float(net[3].running_mean.abs().max()) == 0.0 and int(net[3].num_batches_tracked) == 0
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, betas=(0.5, 0.999))
    g = torch.Generator().manual_seed(10 + rank)     # rank-distinct data
    x = torch.randn(4, 6, generator=g)
    h = net[1](net[0](x))
    loss = h.pow(2).mean()                           # net[2] gets no gradient on any rank -> zero-filled
    net.zero_grad()
    loss.backward()
    local = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in net.parameters()]
    parallel.sync_gradients(net)
    synced = [p.grad.clone() for p in net.parameters()]
    opt.step()
    q.put((rank, [t.numpy() for t in local], [t.numpy() for t in synced], [p.detach().numpy() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, l0, s0, p0), (_, l1, s1, p1) = res
    import numpy as np
    for a, b, sa, sb in zip(l0, l1, s0, s1):
        assert np.allclose(sa, (a + b) / 2, atol=1e-7) and np.array_equal(sa, sb)
    assert any(np.abs(a - b).max() > 1e-4 for a, b in zip(l0, l1))       # the ranks really had different grads
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b)                                      # replicas stay bit-identical


def _worker_flat(rank, world, port, q):
    """the flat-gradient route: .grad are views of one persistent buffer, the collective sums it in place"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from semantic_pyramid_for_image_generation_amd import parallel
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    parallel.broadcast_module(net)
    fg = parallel.flatten_grads(net)
    assert parallel.flatten_grads(net) is fg and all(p.grad.data_ptr() % 16 == 0 for p in net.parameters())
    out = []
    for step in range(2):                            # second step: the views must still be the gradients
        g = torch.Generator().manual_seed(10 + rank + 7 * step)
        x = torch.randn(4, 6, generator=g)
        parallel.zero_gradients(net)                 # keeps the views (the optimizer leaves zeros behind)
        net[1](net[0](x)).pow(2).mean().backward()   # net[2] gets no gradient: its slice stays zero
        local = fg.flat.clone()
        parallel.sync_gradients(net)
        out.append((local.numpy(), fg.flat.clone().numpy()))
        assert all(p.grad is v for p, v in zip(fg.params, fg.views))
        fg.flat.zero_()                              # what optim.Adam(zero_grads=True) does
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_flat, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import numpy as np
    for (l0, s0), (l1, s1) in zip(res[0][1], res[1][1]):
        assert np.allclose(s0, l0 + l1, atol=1e-7) and np.array_equal(s0, s1)      # the SUM; Adam's grad_scale = 1/N makes it the mean
        assert np.abs(l0 - l1).max() > 1e-4


def _worker_product(rank, world, port, q):
    """the PRODUCT critic's flat gradient buffer and its bucketed exchange: three gradient contributions per weight and step, in
    the order a backward pass issues them (last layer first), announced through the hook the weight-gradient path calls"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from semantic_pyramid_for_image_generation_amd import parallel
    from semantic_pyramid_for_image_generation_amd.discriminator import Discriminator
    torch.manual_seed(0)
    D = Discriminator('instance_norm', 'res'); D.init_weights()
    parallel.broadcast_module(D)
    fg = parallel.flatten_grads(D, bucket_bytes=4 << 20)
    assert len(fg.buckets) >= 2 and fg.flat.numel() >= 7026753
    weights = [p for p in D.parameters() if p.dim() >= 2][::-1]
    small = [p for p in D.parameters() if p.dim() < 2]
    out = []
    shadow = torch.zeros_like(fg.flat)               # this rank's own gradient (a bucket may be summed in place before the pass ends)
    where = {id(p): (p.grad.data_ptr() - fg.flat.data_ptr()) // 4 for p in D.parameters()}
    for step in range(4):
        g = torch.Generator().manual_seed(100 * rank + step)
        parallel.zero_gradients(D)
        shadow.zero_()

        def contribute(p):
            r = torch.randn(p.shape, generator=g)
            p.grad.add_(r)
            shadow[where[id(p)]:where[id(p)] + p.numel()].add_(r.reshape(-1))
        for rep in range(3):                         # logit pass + both chains of the penalty's double backward
            for p in weights:
                contribute(p)
                parallel._on_grad_issued(p)
        for p in small:
            contribute(p)
        local = shadow.clone()
        parallel.sync_gradients(D)
        out.append((float(local.double().sum()), float(local.abs().max()), fg.flat[::4099].clone().numpy(), local[::4099].numpy()))
        fg.flat.zero_()
    q.put((rank, out, dict(fg.stats)))
    dist.barrier()
    dist.destroy_process_group()


def test_product_critic_bucketed_exchange_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_product, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    import numpy as np
    (_, o0, st0), (_, o1, st1) = res
    for (s0, m0, f0, l0), (s1, m1, f1, l1) in zip(o0, o1):
        assert np.allclose(f0, l0 + l1, atol=1e-5) and np.array_equal(f0, f1)       # the SUM on both ranks, bit-identical
        assert np.abs(l0 - l1).max() > 1e-2
    # the contribution counts are learned from two identical steps; from then on the buckets go out from inside the pass
    assert st0 == st1 and st0["early"] >= len(res) and st0["late"] > 0 and st0["steps"] == 4
