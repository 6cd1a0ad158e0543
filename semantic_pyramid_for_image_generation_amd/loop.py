"""The loop body of the reference `train_model.py:300-349` as one object, replayable as captured HIP graphs.

One loop iteration = Python-RNG draws (train type, layer, crop) -> classifier taps -> masks -> critic step
(`train_discriminator`) -> every `discriminator_steps`-th iteration a generator step (`train_generator`).  Issued from
Python this is ~500 kernel launches per iteration and the host, not the GPU, sets the pace.
`TrainLoop` therefore captures the body ONCE per variant into a HIP graph (torch.cuda.CUDAGraph is the plumbing: capture
stream, private memory pool) and replays it:

  * the variants are keyed by what changes the kernel sequence: whether the iteration ends with a generator step, the
    train type and the selected layer L (the number of feature-loss terms; which masks are all-zero);
  * everything that changes from iteration to iteration WITHOUT changing the kernel sequence is data in device memory that
    is written before the replay: the image batch, the critic's z (CPU generator, as the reference), alpha and the
    generator step's z / z2 (device generator, drawn in the reference's order), the mask windows (six int32 words per
    mask, read by `sempyr_build_masks`), the two step-dependent Adam scalars (`optim.Adam(capturable=True)`);
  * the step functions that are captured are the very same `train_model.train_discriminator / train_generator` that
    run eagerly — a graph is a recording of them, not a second implementation; replay is bit-identical to eager execution
    (tests/test_graph_loop_gpu.py).

The first time a variant is met it runs eagerly (first use of a shape builds launch plans and uploads tables, which a
capture cannot contain); `prepare()` does that ahead of time for every variant and restores all state afterwards.
"""
import ctypes
import os
import random
from argparse import Namespace

import torch

from . import _lib, ops
from . import train_model as tm

_USE_GRAPHS_ENV = os.environ.get("SEMPYR_GRAPHS", "1") != "0"
# who replays a captured body: "own" = sempyr_replay_launch (plain stream launches from native code, branches on several
# streams); "hip" = hipGraphLaunch through torch (kept for A/B: in ROCm 7.2 it runs the branches one after the other)
_GRAPH_EXEC_ENV = os.environ.get("SEMPYR_GRAPH_EXEC", "own")
_REPLAY_STREAMS = int(os.environ.get("SEMPYR_REPLAY_STREAMS", "4"))


class _Captured:
    """One captured variant of the body: the torch graph object (owner of the private memory pool and of the hipGraph_t),
    the body's outputs (tensors in that pool) and, for the native executor, the replay plan."""

    def __init__(self, graph, out, executor):
        self.graph, self.out, self.plan, self.stats = graph, out, None, None
        if executor == "own":
            plan = ctypes.c_void_p()
            stats = (ctypes.c_int32 * 8)()
            _lib.check(_lib.lib.sempyr_replay_create(graph.raw_cuda_graph(), _REPLAY_STREAMS, ctypes.byref(plan), stats),
                       "sempyr_replay_create")
            self.plan, self.stats = plan, list(stats)

    def replay(self, device):
        if self.plan is not None:
            _lib.check(_lib.lib.sempyr_replay_launch(self.plan, torch._C._cuda_getCurrentRawStream(device.index)),
                       "sempyr_replay_launch")
        else:
            self.graph.replay()

    def __del__(self):
        if self.plan is not None:
            _lib.lib.sempyr_replay_destroy(self.plan)
            self.plan = None


class TrainLoop:
    LOG_ROWS = 64
    LOG_KEYS = ('D/d_loss', 'D/gp', 'G/g_loss', 'G/features_loss', 'G/total_loss', 'G/diversity loss', 'G/total loss')

    def __init__(self, generator, discriminator, classifier, generator_optimizer, discriminator_optimizer, batch_size, device,
                 discriminator_steps=5, train1_prob=0.6, fixed_layer=None, use_graphs=None, executor=None, image_size=256,
                 normalizer_clf=None, normalizer_discriminator=None, criterion_features=None, criterion_diversity_n=None,
                 criterion_diversity_d=None):
        self.G, self.D, self.C = generator, discriminator, classifier
        self.g_opt, self.d_opt = generator_optimizer, discriminator_optimizer
        self.B, self.device = batch_size, torch.device(device)
        self.d_steps, self.train1_prob, self.fixed_layer = discriminator_steps, train1_prob, fixed_layer
        self.use_graphs = _USE_GRAPHS_ENV if use_graphs is None else bool(use_graphs)
        self.executor = executor or _GRAPH_EXEC_ENV
        if self.executor not in ("own", "hip"):
            raise ValueError("TrainLoop: executor is 'own' or 'hip'")
        if self.use_graphs and not (getattr(generator_optimizer, 'capturable', False)
                                    and getattr(discriminator_optimizer, 'capturable', False)):
            raise ValueError("TrainLoop(use_graphs=True) needs optim.Adam(..., capturable=True) optimizers")
        self.normalizer_clf = normalizer_clf or tm.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
        self.normalizer_d = normalizer_discriminator or tm.Normalize([0.5, 0.5, 0.5], [0.5, 0.5, 0.5])
        self.d_loss_fn, self.g_loss_fn = tm.get_wgan_losses_fn()
        self.crit_f = criterion_features or tm.L1Loss()
        self.crit_dn = criterion_diversity_n or tm.L1Loss()
        self.crit_dd = criterion_diversity_d or tm.L1Loss()
        self.iterations = 0
        dev, B = self.device, batch_size
        # device-resident inputs of the body
        self.images = torch.zeros(B, 3, image_size, image_size, device=dev)
        self.z_d = torch.zeros(B, 128, 1, 1, device=dev)
        self.alpha = torch.zeros(B, 1, 1, 1, device=dev)
        self.z_g = torch.zeros(B, 128, 1, 1, device=dev)
        self.z2 = torch.zeros(B, 128, 1, 1, device=dev)
        self.spec_dev = torch.zeros(36, dtype=torch.int32, device=dev)
        self.tap_shapes = None          # known after the first classifier pass
        self.masks = None
        self.graphs = {}
        self.seen = set()
        self.features_to_train = None
        self.train_type = None
        self.last = None                # what the last iteration's body returned (features, masks, ... for callers that log)
        # loss scalars of the last LOG_ROWS iterations, gathered by one launch per iteration (a replayed body overwrites its
        # outputs, and reading them right away would put a host sync into every iteration: train_model.py:333,342 do)
        self.loss_log = torch.zeros(self.LOG_ROWS, len(self.LOG_KEYS), device=dev)
        self._logged = []               # (row, iteration, had generator step, had diversity loss)

    # ------------------------------------------------------------------------------------------------
    def _tap_shapes(self):
        if self.tap_shapes is None:
            s = self.images.shape[2]
            B = self.B
            self.tap_shapes = [(B, 64, s // 2, s // 2), (B, 64, s // 4, s // 4), (B, 128, s // 8, s // 8), (B, 256, s // 16, s // 16),
                               (B, 512, s // 32, s // 32), (B, 512, 1, 1)]
            # only the masks that anything reads are kept as tensors: levels 1..4 (generator.py:126-129, train_model.py:212-222)
            self.masks = [None] + [torch.zeros(self.tap_shapes[i], device=self.device) for i in range(1, 5)] + [None]
        return self.tap_shapes

    def draw(self):
        """The host-side draws of one iteration, in the reference's order (train_model.py:301 train type, :327 layer, :201
        crop, all from Python's `random`).  Returns the variant key (generator step?, train type 1?, layer)."""
        self.iterations += 1
        it = self.iterations
        self.train_type = random.choices([1, 2], [self.train1_prob, 1 - self.train1_prob])
        L = random.randint(1, 4) if self.fixed_layer is None else self.fixed_layer
        self.features_to_train = L
        is1 = (self.train_type == 1)          # a list never equals 1 (quirk Q1): always the type-2 masks
        crop = None if is1 else (random.randint(20, 40), random.randint(20, 40))
        self.specs = tm.mask_specs(self._tap_shapes(), is1, L, crop)
        do_g = (it % self.d_steps == 1)          # train_model.py:336 verbatim (never true for discriminator_steps == 1)
        return (do_g, bool(is1), L)

    def _draw_device(self, key, images, static):
        """The device-side inputs of the body, drawn in the reference's order (train_model.py:128 z on the CPU generator; :53
        alpha, :90 z, :107 z2 on the device generator).  static: written into the buffers a captured body reads; otherwise
        fresh tensors (an eagerly issued body overlaps the previous iteration, which may still be reading its own)."""
        do_g = key[0]
        dev, B = self.device, self.B
        inp = Namespace()
        if static:
            if images is not self.images:
                self.images.copy_(images, non_blocking=True)
            inp.images, inp.spec_dev, inp.masks = self.images, self.spec_dev, self.masks
            inp.z_d, inp.alpha, inp.z_g, inp.z2 = self.z_d, self.alpha, self.z_g, self.z2
        else:
            inp.images = images
            inp.spec_dev = torch.empty(36, dtype=torch.int32, device=dev)
            inp.masks = [None] + [torch.empty(self.tap_shapes[i], device=dev) for i in range(1, 5)] + [None]
            inp.z_d = torch.empty(B, 128, 1, 1, device=dev)
            inp.alpha = torch.empty(B, 1, 1, 1, device=dev)
            inp.z_g = torch.empty(B, 128, 1, 1, device=dev) if do_g else None
            inp.z2 = torch.empty(B, 128, 1, 1, device=dev) if do_g and tm.args.use_diversity_loss else None
        ops.store_words(inp.spec_dev, tm.spec_words(self.specs))
        tm._cpu_randn_to(dev, B, 128, 1, 1, out=inp.z_d)
        torch.rand(inp.alpha.shape, out=inp.alpha)
        if do_g:
            torch.randn(inp.z_g.shape, out=inp.z_g)
            if tm.args.use_diversity_loss:
                torch.randn(inp.z2.shape, out=inp.z2)
        return inp

    def _static_inputs(self):
        inp = Namespace(images=self.images, spec_dev=self.spec_dev, masks=self.masks, z_d=self.z_d, alpha=self.alpha,
                        z_g=self.z_g, z2=self.z2)
        return inp

    def _body(self, key, inp=None, images=None):
        """classifier taps -> masks -> critic step [-> generator step].  `inp`: the device-side inputs (a captured body reads
        the static ones); None: draw them here, inside the prefetch block, so that they are produced on the stream that
        consumes them first."""
        do_g, is1, L = key
        B, dev = self.B, self.device
        train_type = 1 if is1 else [2]
        with tm.prefetch(self.G, dev) as pf:            # (runs in place while a graph is being captured)
            if inp is None:
                inp = self._draw_device(key, images, static=False)
                pf.hand_over(inp.alpha, inp.z_g, inp.z2)
            images_d = self.normalizer_d(inp.images)
            ops.build_masks(inp.masks[1:5], inp.spec_dev[6:30])
            masks = [None] + [tm.tag_mask(inp.masks[i], self.specs[i]) for i in range(1, 5)] + [None]
            upto = tm.live_taps(masks)          # taps behind it are multiplied by all-zero masks wherever they are used
            with torch.no_grad():
                if upto is None:
                    _, features = self.C(self.normalizer_clf(inp.images))
                else:
                    _, features = self.C(self.normalizer_clf(inp.images), taps_upto=upto) if upto else (None, [None] * 6)
            fake = tm.fake_for_discriminator(self.G, B, features, masks, z=inp.z_d)
            pf.hand_over(images_d, features, masks[1:5], fake)
            fake_g = None
            if do_g and pf.on:
                fake_g = tm.fake_for_generator(self.G, features, masks, inp.z_g)
                pf.hand_over(fake_g)
        d_losses = tm.train_discriminator(self.G, self.D, self.d_loss_fn, self.d_opt, images_d, features, masks,
                                          alpha=inp.alpha, fake_images=fake)
        g_losses = None
        if do_g:
            g_losses = tm.train_generator(self.G, self.D, self.g_loss_fn, self.g_opt, B, features, self.crit_f, L, self.C,
                                          self.normalizer_clf, self.crit_dn, self.crit_dd, masks, train_type,
                                          z=inp.z_g, z2=inp.z2, fake_images=fake_g)
        return {'d': d_losses, 'g': g_losses, 'features': features, 'masks': masks, 'images_d': images_d}

    def _capture(self, key):
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph(keep_graph=(self.executor == "own"))
        with ops.capture_epoch():
            with torch.cuda.graph(graph):
                out = self._body(key, self._static_inputs())
        torch.cuda.synchronize(self.device)
        self.graphs[key] = _Captured(graph, out, self.executor)

    def _advance_optimizers(self, do_g):
        self.d_opt.advance()
        if do_g:
            self.g_opt.advance()

    def iteration(self, images, log=False):
        """One loop iteration on `images` ((B,3,H,W) in [0,1], already on the device).  `log`: also keep the loss scalars for
        pop_logged().  Returns {'d': {...}, 'g': {...} | None, 'features', 'masks', 'images_d'}; the tensors of a replayed
        body are overwritten by the next iteration of the same variant."""
        key = self.draw()
        out = None
        if self.use_graphs:
            cap = self.graphs.get(key)
            if cap is None and key in self.seen:
                self._draw_device(key, images, static=True)
                self._capture(key)
                cap = self.graphs[key]
            elif cap is not None:
                self._draw_device(key, images, static=True)
            if cap is not None:
                self._advance_optimizers(key[0])
                cap.replay(self.device)
                # every body touches the generator (its train-mode forwards update the BatchNorm running statistics): an
                # eagerly issued body that follows must not start its generator forward on the prefetch stream before this
                # one is done, whether or not it ended with a generator step
                tm.mark_generator_updated(self.G, self.device)
                out = cap.out
            self.seen.add(key)
        if out is None:
            out = self._body(key, None, images)
        self.last = out
        if log:
            self._log(out)
        return out

    def _log(self, out):
        d, g = out['d'], out['g']
        vals = [d['d_loss'], d['gp']]
        if g is not None:
            vals += [g['g_loss'], g['features_loss'], g['total_loss']]
            if 'diversity loss' in g:
                vals += [g['diversity loss'], g['total loss']]
        if len(self._logged) >= self.LOG_ROWS:
            raise RuntimeError("TrainLoop: loss log is full; call pop_logged() at least every %d iterations" % self.LOG_ROWS)
        row = len(self._logged)
        ops.gather_scalars([v.detach() for v in vals], self.loss_log[row])
        self._logged.append((row, self.iterations, g is not None, g is not None and 'diversity loss' in g))

    def pop_logged(self):
        """[(iteration, {tag: float})] for the iterations logged since the last call (one device-to-host copy)."""
        if not self._logged:
            return []
        host = self.loss_log[:len(self._logged)].cpu()
        out = []
        for row, it, had_g, had_div in self._logged:
            n = 2 + (3 if had_g else 0) + (2 if had_div else 0)
            out.append((it, {k: float(host[row, i]) for i, k in enumerate(self.LOG_KEYS[:n])}))
        self._logged = []
        return out

    # ------------------------------------------------------------------------------------------------
    def _state_tensors(self):
        out = []
        for m in (self.G, self.D):
            out += list(m.parameters()) + list(m.buffers())
        return out

    def prepare(self, keys=None):
        """Warm and capture every variant ahead of time (first use of a shape cannot be captured) on throw-away inputs, then
        put parameters, buffers and optimizer state back bit-exactly: training is unaffected, and no iteration of the real
        loop pays for a capture.  Python's and torch's RNG streams are not touched."""
        if not self.use_graphs:
            return
        if keys is None:
            layers = [self.fixed_layer] if self.fixed_layer is not None else [1, 2, 3, 4]
            keys = [(g, False, L) for g in (False, True) for L in layers]
        shapes = self._tap_shapes()
        gen = torch.Generator(device='cpu').manual_seed(987)
        saved_inputs = [t.clone() for t in (self.images, self.z_d, self.alpha, self.z_g, self.z2)]
        self.images.copy_(torch.rand(self.images.shape, generator=gen))
        for t in (self.z_d, self.z_g, self.z2):
            t.copy_(torch.randn(t.shape, generator=gen))
        self.alpha.copy_(torch.rand(self.alpha.shape, generator=gen))
        with torch.no_grad():
            snap = [(t, t.detach().clone()) for t in self._state_tensors()]
        opt_snap = []
        for opt in (self.g_opt, self.d_opt):
            had = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state[p].items()}
                   for g in opt.param_groups for p in g['params'] if len(opt.state[p]) != 0}
            opt_snap.append((opt, had))
        for key in keys:
            if key in self.graphs:
                continue
            do_g, is1, L = key
            self.specs = tm.mask_specs(shapes, is1, L, None if is1 else (30, 30))
            ops.store_words(self.spec_dev, tm.spec_words(self.specs))
            torch.cuda.synchronize(self.device)      # (the eager body starts on the prefetch stream, which does not follow this one)
            if key not in self.seen:
                self._body(key, self._static_inputs())
                torch.cuda.synchronize(self.device)
                self.seen.add(key)
            self._capture(key)
        with torch.no_grad():
            for t, v in snap:
                t.copy_(v)
            for opt, had in opt_snap:
                for g in opt.param_groups:
                    for p in g['params']:
                        st = opt.state[p]
                        if len(st) == 0:
                            continue
                        if p in had:
                            for k, v in had[p].items():
                                if torch.is_tensor(v):
                                    st[k].copy_(v)
                                else:
                                    st[k] = v
                        else:
                            st['step'].zero_(); st['exp_avg'].zero_(); st['exp_avg_sq'].zero_()
            for opt in (self.g_opt, self.d_opt):
                for gi, g in enumerate(opt.param_groups):
                    torch.autograd.graph.increment_version(list(g['params']))
                    ops.repack_cached(g['params'], None)
            for t, v in zip((self.images, self.z_d, self.alpha, self.z_g, self.z2), saved_inputs):
                t.copy_(v)
        for m in (self.G, self.D):
            for p in m.parameters():
                p.grad = None
        torch.cuda.synchronize(self.device)
