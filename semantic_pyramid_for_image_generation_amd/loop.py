"""The loop body of the reference `train_model.py:300-349` as one object, replayable as captured HIP graphs.

One loop iteration = Python-RNG draws (train type, layer, crop) -> classifier taps -> masks -> critic step
(`train_discriminator`) -> every `discriminator_steps`-th iteration a generator step (`train_generator`).  Issued from
Python this is ~500 kernel launches per iteration and the host, not the GPU, sets the pace.
`TrainLoop` therefore captures the body ONCE per variant and replays it -- as TWO graphs, because the body has two halves
with different dependencies:

  * stage A ("prefetch": both Normalizes, masks, classifier taps, the critic step's detached generator forward and, in an
    iteration that ends with a generator step, that step's own forward with its autograd graph) depends only on the
    generator's last update;
  * stage B (critic step [+ generator step]) depends on stage A and on the previous stage B.

Replayed, stage A of the next `lookahead` iterations is launched on the prefetch stream BEFORE stage B of iteration i is
launched on the main stream (iteration(next_images=...)), so it runs under that step's backward pass and in the gaps at
the iteration boundaries (Adam, weight re-pack, the small kernels at the head of a stage) -- one captured graph per
iteration cannot do this: round 2 measured it slower than eager.  k stage-A graphs in flight need k+1 sets of static
inputs / outputs: every variant is captured once per buffer set ("parity" = iteration number modulo lookahead + 1), and
stage A of iteration i waits for the event behind stage B of iteration i - (lookahead + 1), whose buffers it overwrites.

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
_REPLAY_STREAMS = int(os.environ.get("SEMPYR_REPLAY_STREAMS", "3"))
# how many iterations ahead of the running stage B a stage A may be launched (= sets of static buffers - 1).  Measured
# on the headline configuration (tools/la_sweep.sh): 1 -> 45.3, 2 -> 44.9, 3 -> 44.9 ms per cycle, i.e. nothing beyond the
# run-to-run spread: with one stage A beside every stage B the GPU has no idle share left to fill, so the default stays at
# the smallest memory footprint (two buffer sets, 16 captured graphs).
_LOOKAHEAD_ENV = max(1, int(os.environ.get("SEMPYR_LOOKAHEAD", "1")))


class _Slot:
    """The static inputs of one pipeline parity (what the captured stages read) and the event behind the last stage B that
    read them and stage A's outputs."""

    def __init__(self, B, image_size, shapes, dev):
        self.images = torch.zeros(B, 3, image_size, image_size, device=dev)
        self.z_d = torch.zeros(B, 128, 1, 1, device=dev)
        self.alpha = torch.zeros(B, 1, 1, 1, device=dev)
        self.z_g = torch.zeros(B, 128, 1, 1, device=dev)
        self.z2 = torch.zeros(B, 128, 1, 1, device=dev)
        self.spec_dev = torch.zeros(36, dtype=torch.int32, device=dev)
        # only the masks that anything reads are kept as tensors: levels 1..4 (generator.py:126-129, train_model.py:212-222)
        self.masks = [None] + [torch.zeros(shapes[i], device=dev) for i in range(1, 5)] + [None]
        self.b_done = None

    def tensors(self):
        return (self.images, self.z_d, self.alpha, self.z_g, self.z2)


class _Captured:
    """One captured variant of the body at one parity: the two torch graph objects (owners of the hipGraph_t; the memory
    lives in the loop's per-stage, per-parity pools), the body's outputs and, for the native executor, the replay plans."""

    def __init__(self, graph_a, graph_b, out, executor, side_streams=()):
        self.graph_a, self.graph_b, self.out = graph_a, graph_b, out
        self.plan_a = self.plan_b = self.stats = None
        if executor == "own":
            stats_a, stats_b = (ctypes.c_int32 * 8)(), (ctypes.c_int32 * 8)()
            pa, pb = ctypes.c_void_p(), ctypes.c_void_p()
            # stage A is one chain: it runs on the prefetch stream alone; stage B forks onto the shared side streams
            _lib.check(_lib.lib.sempyr_replay_create(graph_a.raw_cuda_graph(), 1, ctypes.byref(pa), stats_a), "sempyr_replay_create")
            self.plan_a = pa
            # ... starting with the streams the eager schedule uses for its branches: the same stream -> hardware-queue
            # mapping as eager execution (streams created per plan landed two busy streams on one queue: measured)
            side = (ctypes.c_void_p * max(len(side_streams), 1))(*side_streams)
            _lib.check(_lib.lib.sempyr_replay_create_on(graph_b.raw_cuda_graph(), _REPLAY_STREAMS, side, len(side_streams),
                                                        ctypes.byref(pb), stats_b), "sempyr_replay_create_on")
            self.plan_b = pb
            self.stats_a, self.stats = list(stats_a), list(stats_b)

    def _replay(self, plan, graph, raw_stream):
        if plan is not None:
            _lib.check(_lib.lib.sempyr_replay_launch(plan, raw_stream), "sempyr_replay_launch")
        else:
            graph.replay()          # hipGraphLaunch on the current stream

    def replay_a(self, raw_stream):
        self._replay(self.plan_a, self.graph_a, raw_stream)

    def replay_b(self, raw_stream):
        self._replay(self.plan_b, self.graph_b, raw_stream)

    def __del__(self):
        plans = [p for p in (self.plan_a, self.plan_b) if p is not None]
        self.plan_a = self.plan_b = None
        if plans:
            try:
                torch.cuda.synchronize()          # a replay may still be running on the plan's streams / events
            except Exception:  # pragma: no cover - interpreter shutdown
                pass
            for p in plans:
                _lib.lib.sempyr_replay_destroy(p)


class TrainLoop:
    LOG_ROWS = 64
    LOG_KEYS = ('D/d_loss', 'D/gp', 'G/g_loss', 'G/features_loss', 'G/total_loss', 'G/diversity loss', 'G/total loss')

    def __init__(self, generator, discriminator, classifier, generator_optimizer, discriminator_optimizer, batch_size, device,
                 discriminator_steps=5, train1_prob=0.6, fixed_layer=None, use_graphs=None, executor=None, image_size=256,
                 lookahead=None,
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
        self.iterations = 0             # iterations drawn (one ahead of `current` while a stage A is launched ahead)
        self.current = 0                # number of the iteration that the last iteration() call ran
        dev, B = self.device, batch_size
        self.image_size = image_size
        s_ = image_size
        self.tap_shapes = [(B, 64, s_ // 2, s_ // 2), (B, 64, s_ // 4, s_ // 4), (B, 128, s_ // 8, s_ // 8), (B, 256, s_ // 16, s_ // 16),
                           (B, 512, s_ // 32, s_ // 32), (B, 512, 1, 1)]
        # device-resident inputs of the captured stages, one set per pipeline parity (created with the first capture)
        self.slots = None
        self.graphs = {}                # (variant key, parity) -> _Captured
        self._pools = {}
        self._cap_stream = None
        self.lookahead = _LOOKAHEAD_ENV if lookahead is None else max(1, int(lookahead))
        self.nslots = self.lookahead + 1
        self._parity = 0                # the buffer set ("parity": iteration number modulo nslots) of the iteration drawn last
        self._ahead = []                # [(key, parity, graphs, event, attributes)] of iterations whose stage A is already launched
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
        return self.tap_shapes

    def _slots(self):
        if self.slots is None:
            self.slots = [_Slot(self.B, self.image_size, self.tap_shapes, self.device) for _ in range(self.nslots)]
        return self.slots

    def draw(self):
        """The host-side draws of one iteration, in the reference's order (train_model.py:301 train type, :327 layer, :201
        crop, all from Python's `random`).  Returns the variant key (generator step?, train type 1?, layer)."""
        self.iterations += 1
        it = self.current = self.iterations
        self.train_type = random.choices([1, 2], [self.train1_prob, 1 - self.train1_prob])
        L = random.randint(1, 4) if self.fixed_layer is None else self.fixed_layer
        self.features_to_train = L
        is1 = (self.train_type == 1)          # a list never equals 1 (quirk Q1): always the type-2 masks
        crop = None if is1 else (random.randint(20, 40), random.randint(20, 40))
        self.specs = tm.mask_specs(self._tap_shapes(), is1, L, crop)
        do_g = (it % self.d_steps == 1)          # train_model.py:336 verbatim (never true for discriminator_steps == 1)
        return (do_g, bool(is1), L)

    def _draw_device(self, key, images, slot):
        """The device-side inputs of the body, drawn in the reference's order (train_model.py:128 z on the CPU generator; :53
        alpha, :90 z, :107 z2 on the device generator), on the current stream.  slot: written into the buffers a captured
        body reads; None: fresh tensors (an eagerly issued body overlaps the previous iteration, which may still be reading
        its own).  `images` may still be in host memory: it is uploaded here, on the stream that consumes it."""
        do_g = key[0]
        dev, B = self.device, self.B
        inp = Namespace()
        if slot is not None:
            if images is not slot.images:
                slot.images.copy_(images, non_blocking=True)
            inp.images, inp.spec_dev, inp.masks = slot.images, slot.spec_dev, slot.masks
            inp.z_d, inp.alpha, inp.z_g, inp.z2 = slot.z_d, slot.alpha, slot.z_g, slot.z2
        else:
            inp.images = images if images.is_cuda else images.to(dev, non_blocking=True)
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

    @staticmethod
    def _slot_inputs(slot):
        return Namespace(images=slot.images, spec_dev=slot.spec_dev, masks=slot.masks, z_d=slot.z_d, alpha=slot.alpha,
                         z_g=slot.z_g, z2=slot.z2)

    def _stage_a(self, key, inp, with_g_forward):
        """Both Normalizes, masks, classifier taps, the critic step's detached generator forward and (with_g_forward) the
        generator step's own forward -- everything of an iteration that does not depend on the previous critic step."""
        do_g, is1, L = key
        pre = Namespace()
        pre.images_d = self.normalizer_d(inp.images)
        ops.build_masks(inp.masks[1:5], inp.spec_dev[6:30])
        pre.masks = [None] + [tm.tag_mask(inp.masks[i], self.specs[i]) for i in range(1, 5)] + [None]
        upto = tm.live_taps(pre.masks)          # taps behind it are multiplied by all-zero masks wherever they are used
        with torch.no_grad():
            if upto is None:
                _, pre.features = self.C(self.normalizer_clf(inp.images))
            else:
                _, pre.features = self.C(self.normalizer_clf(inp.images), taps_upto=upto) if upto else (None, [None] * 6)
        pre.fake = tm.fake_for_discriminator(self.G, self.B, pre.features, pre.masks, z=inp.z_d)
        pre.fake_g = tm.fake_for_generator(self.G, pre.features, pre.masks, inp.z_g) if (do_g and with_g_forward) else None
        return pre

    def _stage_b(self, key, inp, pre):
        """critic step [-> generator step] (train_model.py:331-339)"""
        do_g, is1, L = key
        train_type = 1 if is1 else [2]
        d_losses = tm.train_discriminator(self.G, self.D, self.d_loss_fn, self.d_opt, pre.images_d, pre.features, pre.masks,
                                          alpha=inp.alpha, fake_images=pre.fake)
        g_losses = None
        if do_g:
            g_losses = tm.train_generator(self.G, self.D, self.g_loss_fn, self.g_opt, self.B, pre.features, self.crit_f, L, self.C,
                                          self.normalizer_clf, self.crit_dn, self.crit_dd, pre.masks, train_type,
                                          z=inp.z_g, z2=inp.z2, fake_images=pre.fake_g)
        return {'d': d_losses, 'g': g_losses, 'features': pre.features, 'masks': pre.masks, 'images_d': pre.images_d}

    def _body(self, key, inp=None, images=None):
        """One eagerly issued iteration: stage A inside the prefetch block (its own stream, see train_model.prefetch), then
        stage B.  `inp`: the device-side inputs; None: draw them here, inside the prefetch block, so that they are produced on
        the stream that consumes them first."""
        with tm.prefetch(self.G, self.device) as pf:
            if inp is None:
                inp = self._draw_device(key, images, None)
                pf.hand_over(inp.alpha, inp.z_g, inp.z2)
            pre = self._stage_a(key, inp, pf.on)
            pf.hand_over(pre.images_d, pre.features, pre.masks[1:5], pre.fake, pre.fake_g)
        return self._stage_b(key, inp, pre)

    def _capture(self, key, parity):
        """Record the two stages of variant `key` reading / writing the buffers of `parity`.  Both captures run on one capture
        stream (stage B's backward pass replays nodes of the autograd graph that stage A built: the engine issues them on the
        stream of their forward) and in different capture epochs (stage B must not wait on events recorded inside stage A's
        capture; stage A is complete when stage B starts, by the event the launcher records between them)."""
        dev = self.device
        torch.cuda.synchronize(dev)
        if self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream(device=dev)
        for st in ('A', 'B'):
            if (st, parity) not in self._pools:
                self._pools[(st, parity)] = torch.cuda.graph_pool_handle()
        keep = (self.executor == "own")
        ga, gb = torch.cuda.CUDAGraph(keep_graph=keep), torch.cuda.CUDAGraph(keep_graph=keep)
        inp = self._slot_inputs(self._slots()[parity])
        with ops.capture_epoch():
            with torch.cuda.graph(ga, pool=self._pools[('A', parity)], stream=self._cap_stream):
                pre = self._stage_a(key, inp, True)
        # hipGraphLaunch (kept for A/B only) mis-orders the generator step's forked critic branch when the step's own
        # forward lives in another graph: the branch reads the critic's weights of BEFORE the critic step that the same graph
        # contains (measured on ROCm 7.2: g_loss off by one Adam step; the native executor replays the very same DAG
        # bit-exactly).  For that executor the generator step is recorded unforked.
        saved_fork = ops.TWO_STREAM_GENERATOR
        if self.executor == "hip":
            ops.TWO_STREAM_GENERATOR = False
        try:
            with ops.capture_epoch():
                with torch.cuda.graph(gb, pool=self._pools[('B', parity)], stream=self._cap_stream):
                    out = self._stage_b(key, inp, pre)
        finally:
            ops.TWO_STREAM_GENERATOR = saved_fork
        del pre
        torch.cuda.synchronize(dev)
        side = [ops._side_stream(dev).cuda_stream, ops._side_stream2(dev).cuda_stream]
        self.graphs[(key, parity)] = _Captured(ga, gb, out, self.executor, side)

    def _advance_optimizers(self, do_g):
        self.d_opt.advance()
        if do_g:
            self.g_opt.advance()

    def _issue_a(self, cap, key, images, parity):
        """Launch the captured stage A of one iteration on the prefetch stream: behind the generator's last update and behind
        the stage B that last read this parity's buffers.  Returns the event behind it."""
        dev = self.device
        slot = self.slots[parity]
        main = ops._current_stream_obj(dev)
        pre = ops.prefetch_stream(dev)
        torch.cuda.set_stream(pre)
        try:
            if slot.b_done is not None:
                pre.wait_event(slot.b_done)
            tm._wait_for_generator_update(self.G, dev)
            self._draw_device(key, images, slot)
            cap.replay_a(pre.cuda_stream)
            a_done = torch.cuda.Event()
            a_done.record(pre)
        finally:
            torch.cuda.set_stream(main)
        return a_done

    def _issue_b(self, cap, key, parity, a_done):
        """Launch the captured stage B on the current stream, behind its stage A."""
        dev = self.device
        slot = self.slots[parity]
        main = ops._current_stream_obj(dev)
        main.wait_event(a_done)
        self._advance_optimizers(key[0])
        cap.replay_b(main.cuda_stream)
        slot.b_done = torch.cuda.Event()
        slot.b_done.record(main)
        if key[0]:
            # (every generator forward of a replayed iteration runs on the prefetch stream or, inside the generator step,
            # in front of this event: an eager iteration that follows is ordered behind all of them)
            tm.mark_generator_updated(self.G, dev)
        return cap.out

    def _next_variant(self):
        """The draws of the next iteration and the captured graphs of its variant (captured now if it has been seen before)."""
        key = self.draw()
        self._parity = (self._parity + 1) % self.nslots
        parity = self._parity
        cap = None
        if self.use_graphs:
            cap = self.graphs.get((key, parity))
            if cap is None and (key, parity) in self.seen:
                self._capture(key, parity)
                cap = self.graphs[(key, parity)]
            self.seen.add((key, parity))
        return key, parity, cap

    def iteration(self, images, log=False, next_images=None):
        """One loop iteration on `images` ((B,3,H,W) in [0,1]; on the device and complete, or in (pinned) host memory -- it is
        then uploaded on the prefetch stream).  `log`: also keep the loss scalars for pop_logged().  Returns {'d': {...},
        'g': {...} | None, 'features', 'masks', 'images_d'}; the tensors of a replayed body are overwritten by the replayed
        iteration after the next (whose stage A the NEXT call launches, when that call is given `next_images`).

        `next_images`: the batch of the NEXT call, or a list with the batches of the next calls in order (`lookahead` of them
        are used).  Stage A of those iterations (draws, taps, masks, generator forwards) is then launched on the prefetch
        stream BEFORE this iteration's stage B is launched on the main stream -- up to and including the first iteration that
        ends with a generator step, whose update every later stage A must see.  The host needs ~20 us per launch and a
        stage B has ~400 of them, about as long as the GPU needs to run it: issued in program order, stage A of iteration i+1
        reaches the GPU when stage B of iteration i is nearly done and the two never overlap (measured: 0.0 ms of overlap
        between the prefetch stream and the others in a kernel trace); issued first, it runs beside that stage B.  The draws
        (Python `random`, CPU and device generators) happen in the same order either way, and the results are identical.
        The next calls must pass those batches as `images`, in that order.  Callers that touch the generator between two calls (sampling,
        checkpoints) pass None for that call."""
        out = None
        if self._ahead:
            key, parity, cap, a_done, attrs = self._ahead.pop(0)
            self._set_attrs(attrs)
        else:
            key, parity, cap = self._next_variant()
            a_done = self._issue_a(cap, key, images, parity) if cap is not None else None
        if cap is not None:
            if next_images is not None:
                upcoming = list(next_images) if isinstance(next_images, (list, tuple)) else [next_images]
                attrs = self._get_attrs()
                # a stage A must see every generator update in front of it; an eagerly issued iteration draws its device
                # inputs when it is called, so nothing behind it may draw earlier
                blocked = key[0] or any(e[0][0] or e[2] is None for e in self._ahead)
                while not blocked and len(self._ahead) < min(len(upcoming), self.lookahead):
                    nkey, nparity, ncap = self._next_variant()
                    n_done = self._issue_a(ncap, nkey, upcoming[len(self._ahead)], nparity) if ncap is not None else None
                    self._ahead.append((nkey, nparity, ncap, n_done, self._get_attrs()))
                    blocked = nkey[0] or ncap is None
                self._set_attrs(attrs)
            out = self._issue_b(cap, key, parity, a_done)
        if out is None:
            out = self._body(key, None, images)
        self.last = out
        if log:
            self._log(out)
        return out

    def _get_attrs(self):
        return (self.train_type, self.features_to_train, self.specs, self.current)

    def _set_attrs(self, attrs):
        self.train_type, self.features_to_train, self.specs, self.current = attrs

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
        self._logged.append((row, self.current, g is not None, g is not None and 'diversity loss' in g))

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
        """Warm and capture every variant (at both parities) ahead of time -- first use of a shape cannot be captured -- on
        throw-away inputs, then put parameters, buffers and optimizer state back bit-exactly: training is unaffected, and
        no iteration of the real loop pays for a capture.  Python's and torch's RNG streams are not touched."""
        if not self.use_graphs:
            return
        if keys is None:
            layers = [self.fixed_layer] if self.fixed_layer is not None else [1, 2, 3, 4]
            keys = [(g, False, L) for g in (False, True) for L in layers]
        shapes = self._tap_shapes()
        gen = torch.Generator(device='cpu').manual_seed(987)
        slots = self._slots()
        for slot in slots:
            slot.images.copy_(torch.rand(slot.images.shape, generator=gen))
            for t in (slot.z_d, slot.z_g, slot.z2):
                t.copy_(torch.randn(t.shape, generator=gen))
            slot.alpha.copy_(torch.rand(slot.alpha.shape, generator=gen))
        with torch.no_grad():
            snap = [(t, t.detach().clone()) for t in self._state_tensors()]
        opt_snap = []
        for opt in (self.g_opt, self.d_opt):
            had = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state[p].items()}
                   for g in opt.param_groups for p in g['params'] if len(opt.state[p]) != 0}
            opt_snap.append((opt, had))
        warmed = set()
        for key in keys:
            do_g, is1, L = key
            for parity in range(self.nslots):
                if (key, parity) in self.graphs:
                    continue
                self.specs = tm.mask_specs(shapes, is1, L, None if is1 else (30, 30))
                ops.store_words(slots[parity].spec_dev, tm.spec_words(self.specs))
                torch.cuda.synchronize(self.device)      # (the eager body starts on the prefetch stream, which does not follow this one)
                if key not in warmed and not any((key, q) in self.seen for q in range(self.nslots)):
                    self._body(key, self._slot_inputs(slots[parity]))
                    torch.cuda.synchronize(self.device)
                warmed.add(key)
                self.seen.add((key, parity))
                self._capture(key, parity)
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
        for m in (self.G, self.D):
            for p in m.parameters():
                p.grad = None
        torch.cuda.synchronize(self.device)
