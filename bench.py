#!/usr/bin/env python3
"""bench.py — WGAN-GP training loop throughput (BASELINE.json metric) on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): reference train_model.py loop body, gen-type res, instance_norm critic,
GP weight 10, batch 16 per GPU, synthetic 256x256 images resident in HBM, seeded random-init G/D/ResNet18
(no trained weights ship).  One *step* = one cycle of the reference schedule = 5 loop iterations
(classifier taps + masks + D step each; a G step in the first one) — train_model.py:300-349.
`value` = loop iterations per second over all ranks (5 * K * N / seconds).

Extra objects on the same JSON line: `roofline` for the dominant kernel (the fp32-MFMA implicit-GEMM
convolution; per-launch HIP-event timing inside the timed region), `cpu_baseline` (the CPU oracle port timed on
the host cores, rank 0 / N=1 only) and the eval images/sec of the eval_model.py loop.
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# SURVEY.md §8(d) algorithmic FLOPs (2 FLOP/MAC, reference's wasted work excluded), per sample:
F_G, F_D, F_C = 19.053e9, 1.703e9, 4.738e9
D_STEP_FLOPS = F_C + F_G + 12 * F_D            # 44.23 GFLOP / sample
G_STEP_FLOPS = 3 * F_G + 2 * F_D + 2 * F_C     # 70.04 GFLOP / sample
G_STEP_DIV_FLOPS = 6 * F_G + 2 * F_D + 2 * F_C
EVAL_FLOPS_PER_IMAGE = F_G + F_C / 16


def build_models(device, gen_type, norm, seed=0):
    from semantic_pyramid_for_image_generation_amd.classifier import Classifier, resnet18
    from semantic_pyramid_for_image_generation_amd.discriminator import Discriminator
    from semantic_pyramid_for_image_generation_amd.generator import Generator
    torch.manual_seed(seed)
    G = Generator(gen_type); G.init_weights()
    D = Discriminator(norm, gen_type); D.init_weights()
    C = Classifier(resnet18(365))
    G.to(device); D.to(device); C.to(device).eval()
    for p in C.parameters():
        p.requires_grad_(False)
    return G, D, C


def _cpu_threads():
    """Threads for the CPU baseline: torch's default on this host, i.e. all cores unless the box pins a share of them
    (OMP_NUM_THREADS: the GPU box gives a one-GPU job 16 and asks jobs to stay within it)."""
    return torch.get_num_threads()


def cpu_baseline(batch, diversity):
    """The oracle (CPU restatement, kind 'port') on the host cores: 1 classifier pass + 1 D step + 1 G step at the bench
    batch size after a small warm-up, combined with the reference schedule (5 D + 1 G per cycle)."""
    from oracle import masks as omasks
    from oracle import nets, recipes, steps
    from semantic_pyramid_for_image_generation_amd.discriminator import Discriminator
    from semantic_pyramid_for_image_generation_amd.generator import Generator
    cores = _cpu_threads()
    torch.manual_seed(0)
    G = Generator('res'); G.init_weights()
    D = Discriminator('instance_norm', 'res'); D.init_weights()
    clf_sd = {k: v.detach() for k, v in recipes.make_resnet(0).state_dict().items()}

    def one(bs, timed):
        tr = steps.Trainer(G.state_dict(), D.state_dict(), clf_sd, 'res', 'instance_norm', use_diversity=diversity)
        images = recipes.image_batch(0, bs)
        t0 = time.time()
        with torch.no_grad():
            _, feats = nets.classifier_forward(clf_sd, recipes.normalize_clf(images))
        t_clf = time.time() - t0
        ms, _ = omasks.masks_part2([tuple(f.shape) for f in feats], 3, random.Random(0))
        masks = [torch.from_numpy(m) for m in ms]
        g = torch.Generator().manual_seed(1)
        z = torch.randn(bs, 128, 1, 1, generator=g); z2 = torch.randn(bs, 128, 1, 1, generator=g)
        alpha = torch.rand(bs, 1, 1, 1, generator=g)
        images_d = (images - 0.5) / 0.5
        t0 = time.time(); tr.d_step(images_d, feats, masks, z, alpha); t_d = time.time() - t0
        t0 = time.time(); tr.g_step(feats, masks, 3, z, z2 if diversity else None); t_g = time.time() - t0
        return t_clf, t_d, t_g
    one(2, False)                                   # warm-up: thread pools, oneDNN primitive caches, allocator
    reps = [one(batch, True) for _ in range(3)]     # three timed repetitions (~8 s each on 16 cores): the median is reported
    med = lambda i: sorted(r[i] for r in reps)[1]
    t_clf, t_d, t_g = med(0), med(1), med(2)
    cycle = 5 * (t_clf + t_d) + t_g
    return {"value": 5.0 / cycle, "unit": "iters/s", "cores": cores, "os_cpu_count": os.cpu_count(), "kind": "port",
            "sample": f"oracle (torch-CPU restatement) at batch {batch}, after a batch-2 warm-up, medians of 3 repetitions: 1 classifier "
                      f"pass {t_clf:.2f}s + 1 D step {t_d:.2f}s + 1 G step {t_g:.2f}s, combined as 5*(clf+D)+G per cycle",
            "repetitions": [{"clf_s": round(r[0], 3), "d_step_s": round(r[1], 3), "g_step_s": round(r[2], 3)} for r in reps],
            "torch": torch.__version__}


def cpu_baseline_eval():
    """eval_model.py loop on the host cores: 1 classifier pass + 2 eval-mode generator forwards at batch 8 after a warm-up
    forward, extrapolated to the loop's 16 forwards per source batch (128 images)."""
    from oracle import nets, recipes
    from semantic_pyramid_for_image_generation_amd.generator import Generator
    torch.manual_seed(0)
    G = Generator('res'); G.init_weights()
    gsd = {k: v.detach() for k, v in G.state_dict().items()}
    clf_sd = {k: v.detach() for k, v in recipes.make_resnet(0).state_dict().items()}
    images = recipes.image_batch(0, 8)
    with torch.no_grad():
        t0 = time.time(); _, feats = nets.classifier_forward(clf_sd, recipes.normalize_clf(images)); t_clf = time.time() - t0
        masks = [torch.ones_like(f) for f in feats]
        z = torch.randn(8, 128, 1, 1)
        nets.generator_forward(gsd, 'res', z[:2], [f[:2] for f in feats], [m[:2] for m in masks], False)        # warm-up
        ts = []
        for _ in range(5):
            t0 = time.time()
            nets.generator_forward(gsd, 'res', z, feats, masks, False)
            ts.append(time.time() - t0)
        t_g = sorted(ts)[2]
    total = t_clf + 16 * t_g
    return {"value": 128.0 / total, "unit": "images/s", "cores": _cpu_threads(), "os_cpu_count": os.cpu_count(), "kind": "port",
            "sample": f"oracle at batch 8: 1 classifier pass {t_clf:.2f}s + the median of 5 generator forwards ({t_g:.2f}s), "
                      "16 forwards = 128 images per source batch"}


def launch_ranks(n, argv):
    """Run this script as `n` ranks of one node under torch.distributed.run (RCCL rendezvous on 127.0.0.1, a free port)
    in a child process; returns its exit code.  stdout / stderr of the ranks pass through (rank 0 prints the line)."""
    import socket
    import subprocess
    if 'SEMPYR_SHARE_GPU' not in os.environ and torch.cuda.device_count() < n:          # (counting devices does not initialise HIP)
        print(f"bench.py: --gpus {n} but this node shows {torch.cuda.device_count()} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def hbm_report(steps):
    """Per HBM-bound kernel of the bracket that sempyr_profile_end just closed: launches, algorithmic bytes (4 B x
    elements read once + written once, counted by the library at the launch site) and summed HIP-event duration."""
    from semantic_pyramid_for_image_generation_amd import _lib
    buf = ctypes.create_string_buffer(1 << 16)
    _lib.check(_lib.lib.sempyr_profile_hbm_report(buf, len(buf)), "sempyr_profile_hbm_report")
    rows = []
    for line in buf.value.decode().strip().splitlines()[1:]:
        name, n, nbytes, ms = line.rsplit(',', 3)
        n, nbytes, ms = int(n), float(nbytes), float(ms)
        if ms <= 0:
            continue
        rows.append({"kernel": name, "launches_per_cycle": n / steps, "ms_per_cycle": ms / steps,
                     "algorithmic_MB_per_launch": nbytes / n / 1e6, "achieved_GBps": nbytes / (ms * 1e-3) / 1e9,
                     "frac_of_peak": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS})
    rows.sort(key=lambda r: -r["ms_per_cycle"])
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--gen-type', default='res', choices=['res', 'default'])
    ap.add_argument('--discriminator-norm', default='instance_norm')
    ap.add_argument('--use-diversity-loss', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-eval', action='store_true')
    ap.add_argument('--no-bf16x3', action='store_true', help='skip the separate split-bf16 figure')
    ap.add_argument('--profile-pass', action='store_true',
                    help='run ONLY the single-stream instrumented pass of --steps cycles, bracketed by sempyr_marker_kernel launches '
                         '(for rocprofv3 --kernel-trace; SEMPYR_PROFILE_DUMP=<csv> keeps the per-launch shapes, see tools/roofline_table.py)')
    ap.add_argument('--loop-mode', default=os.environ.get('SEMPYR_LOOP_MODE', 'replay'), choices=['eager', 'replay', 'hipgraph'],
                    help='eager: the body is issued from Python on four streams; replay: captured once per variant and re-issued by '
                         'sempyr_replay_launch; hipgraph: captured and replayed by hipGraphLaunch')
    a = ap.parse_args()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher around it: start the N ranks ourselves (one process per GPU) and
        # relay rank 0's JSON line and the exit code.  This parent never touches the GPU (nothing above initialises HIP) and
        # never replaces itself: the ranks are children.
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    if os.environ.get('SEMPYR_BENCH_TRACE'):          # debugging aid: where is every thread after N seconds?
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['SEMPYR_BENCH_TRACE']), exit=True)

    from semantic_pyramid_for_image_generation_amd import _lib, eval_model, loop, ops, optim, parallel
    from semantic_pyramid_for_image_generation_amd import train_model as tm
    device = parallel.init_device()
    rank, world = parallel.rank(), parallel.world_size()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")

    G, D, C = build_models(device, a.gen_type, a.discriminator_norm)
    parallel.broadcast_module(G); parallel.broadcast_module(D); parallel.broadcast_module(C)
    tm.args = Namespace(use_diversity_loss=a.use_diversity_loss, gradient_penalty_weight=10.0)
    dp = {}
    if world > 1:          # flat gradient buffers: one in-place RCCL all-reduce per optimizer step, 1/N folded into Adam
        parallel.flatten_grads(G); parallel.flatten_grads(D, bucket_bytes=parallel.BUCKET_BYTES)
        dp = dict(grad_scale=1.0 / world, zero_grads=True)
    g_opt = optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True, **dp)
    d_opt = optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True, **dp)
    normalizer_clf = tm.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])

    B = a.batch
    gen = torch.Generator().manual_seed(1234 + rank)
    pool = [torch.rand(B, 3, 256, 256, generator=gen).to(device) for _ in range(5)]    # resident in HBM
    torch.manual_seed(100 + rank)          # rank-distinct z / alpha
    random.seed(0)                         # identical layer / crop draws on every rank
    # the loop body of train_model.py:300-349 (draws, classifier taps, masks, critic step, generator step every 5th
    # iteration) -- the object train_model._main drives; `--loop-mode` picks how the recorded body is issued
    def make_loop():
        return loop.TrainLoop(G, D, C, g_opt, d_opt, B, device, use_graphs=(a.loop_mode != 'eager' and world == 1),
                              executor='hip' if a.loop_mode == 'hipgraph' else 'own')
    lp = make_loop()
    it = [0]

    layers = []

    def cycle(lp_=None, lookahead=True):
        # the loop is told the next batch, so that it can launch the next iteration's stage A ahead of this one's stage B
        # (loop.TrainLoop.iteration); lookahead=False on the last cycle before the loop's mode is switched
        lp_ = lp_ or lp
        for j in range(5):
            it[0] += 1
            ahead = lp_.lookahead if lookahead else min(lp_.lookahead, 4 - j)
            nxt = [pool[(it[0] + 1 + a) % len(pool)] for a in range(ahead)]
            lp_.iteration(pool[it[0] % len(pool)], next_images=nxt or None)
            layers.append(lp_.features_to_train)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # first use of every shape (launch plans, gather tables, packed weights, allocator pools for all four feature levels and
    # both mask types) and the captures happen here, before the W warmup steps the contract asks for
    lp.prepare()
    for _ in range(2):
        cycle()
    for _ in range(a.warmup):
        cycle()
    timing = not a.no_kernel_timing
    if a.profile_pass:
        cycle(lookahead=False)          # (no stage A of a captured iteration left in flight)
        ops.TWO_STREAM_CRITIC = ops.TWO_STREAM_GENERATOR = ops.PREFETCH = False
        ops.SIDE_ON_MAIN = True         # (the product's weight-gradient kernels, issued on the one stream)
        lp.use_graphs = False
        cycle()
        barrier()
        kms, kl, kf = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0)
        raw = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib.sempyr_profile_begin(4096 * max(a.steps, 1)), "sempyr_profile_begin")
        _lib.check(_lib.lib.sempyr_profile_marker(1, raw), "sempyr_profile_marker")
        t1 = time.perf_counter()
        for _ in range(a.steps):
            cycle()
        _lib.check(_lib.lib.sempyr_profile_marker(2, raw), "sempyr_profile_marker")
        barrier()
        el = time.perf_counter() - t1
        _lib.check(_lib.lib.sempyr_profile_end(ctypes.byref(kms), ctypes.byref(kl), ctypes.byref(kf)), "sempyr_profile_end")
        hbm_path = os.environ.get("SEMPYR_HBM_DUMP")
        if hbm_path:
            with open(hbm_path, "w") as f:
                json.dump({"steps": a.steps, "kernels": hbm_report(a.steps)}, f, indent=1)
        print(json.dumps({"profile_pass": True, "steps": a.steps, "ms_per_step_single_stream": 1e3 * el / a.steps,
                          "conv_launches": int(kl.value), "conv_ms_hip_events": kms.value, "conv_gflop_2MNK": kf.value * 1e-9}), flush=True)
        return
    barrier()
    mark = os.environ.get('SEMPYR_MARK_TIMED', '0') == '1'        # tools/busy_union.py: brackets the timed region in a kernel trace
    if mark:
        _lib.check(_lib.lib.sempyr_profile_marker(1, torch.cuda.current_stream().cuda_stream), "sempyr_profile_marker")
    del layers[:]
    c0 = time.process_time()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        cycle()
    host_issue = time.perf_counter() - t0          # all launches issued (GPU may still be running)
    if mark:
        _lib.check(_lib.lib.sempyr_profile_marker(2, torch.cuda.current_stream().cuda_stream), "sempyr_profile_marker")
    barrier()
    elapsed = time.perf_counter() - t0
    cpu_time = time.process_time() - c0
    timed_layers = list(layers)
    # The same K steps without the two exact shortcuts that rest on the reference's all-zero masks (quirk Q2: no convolution of
    # a zero-masked tap, no classifier layer behind the deepest tap with a non-zero mask).  Results are bit-identical either
    # way (tests/test_graph_loop_gpu.py::test_exact_shortcuts_change_nothing); reported so that either number can be quoted.
    from semantic_pyramid_for_image_generation_amd import generator as gmod
    plain = None
    if (gmod.ZERO_SKIP or tm.DEAD_TAPS) and not a.profile_pass:
        saved_sc = (gmod.ZERO_SKIP, tm.DEAD_TAPS)
        gmod.ZERO_SKIP, tm.DEAD_TAPS = False, False
        rs = random.getstate()
        lp_plain = make_loop()          # (a captured body has the shortcuts recorded in it: this loop records its own, without)
        lp_plain.prepare()
        cycle(lp_plain)
        barrier()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            cycle(lp_plain)
        barrier()
        plain = time.perf_counter() - t1
        del lp_plain
        gmod.ZERO_SKIP, tm.DEAD_TAPS = saved_sc
        random.setstate(rs)
        if world > 1:
            t = torch.tensor([plain], dtype=torch.float64, device='cpu' if torch.distributed.get_backend() == 'gloo' else device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            plain = float(t.item())
    # The same K steps with the opt-in split-bf16 convolution arithmetic (SEMPYR_MATH_BF16X3: 3x3 stride-1 convolutions and their
    # input gradients on the bf16 matrix pipe, own tolerances in tests/test_bf16x3_gpu.py).  A separate figure: it never
    # replaces `value`, which is exact fp32.
    bf16x3 = None
    if not a.profile_pass and not a.no_bf16x3 and world == 1:
        saved_math = ops.CONV_MATH
        ops.CONV_MATH = _lib.MATH_BF16X3
        rs = random.getstate()
        try:
            lp16 = make_loop()
            lp16.prepare()
            for _ in range(2):
                cycle(lp16)
            barrier()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                cycle(lp16)
            barrier()
            bf16x3 = time.perf_counter() - t1
            del lp16
        finally:
            ops.CONV_MATH = saved_math
            random.setstate(rs)
    # Kernel durations for the roofline: the SAME K steps run once more with a HIP event pair around every conv launch
    # (recorded on the launch's own stream).  They are not recorded during the headline pass because an event pair per
    # launch keeps consecutive kernels from overlapping (measured: +4 % on the cycle time); `instrumented_ms_per_step`
    # reports the cost.
    kms, kl, kf = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0)
    elapsed_instr = None
    hbm_rows = []
    if timing:
        cycle(lookahead=False)          # (untimed: no stage A of a captured iteration left in flight when the mode changes)
        barrier()
        # single-stream for this pass: a kernel's event-to-event time is only its own duration when nothing else shares the GPU
        saved = (ops.TWO_STREAM_CRITIC, ops.TWO_STREAM_GENERATOR, ops.SIDE_ON_MAIN, ops.PREFETCH)
        ops.TWO_STREAM_CRITIC = ops.TWO_STREAM_GENERATOR = ops.PREFETCH = False
        ops.SIDE_ON_MAIN = True         # (the product's weight-gradient kernels, issued on the one stream)
        saved_graphs, lp.use_graphs = lp.use_graphs, False          # event pairs are recorded by the launch functions: issue eagerly
        _lib.check(_lib.lib.sempyr_profile_begin(4096 * max(a.steps, 1)), "sempyr_profile_begin")
        t1 = time.perf_counter()
        for _ in range(a.steps):
            cycle()
        barrier()
        elapsed_instr = time.perf_counter() - t1
        _lib.check(_lib.lib.sempyr_profile_end(ctypes.byref(kms), ctypes.byref(kl), ctypes.byref(kf)), "sempyr_profile_end")
        hbm_rows = hbm_report(a.steps)
        ops.TWO_STREAM_CRITIC, ops.TWO_STREAM_GENERATOR, ops.SIDE_ON_MAIN, ops.PREFETCH = saved
        lp.use_graphs = saved_graphs
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if torch.distributed.get_backend() == 'gloo' else device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # eval_model.py loop (BASELINE configs[0] shape: batch 8, 4 levels x 4 noises) — images/sec
    eval_ips = None
    if not a.no_eval:
        G.eval()
        src = pool[0][:8]
        eval_model.synthesize_batch(G, C, src, normalizer_clf)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            eval_model.synthesize_batch(G, C, src, normalizer_clf)
        torch.cuda.synchronize()
        eval_ips = reps * 128 / (time.perf_counter() - t1)
        # kernel durations of the same loop for its roofline (second pass, HIP events around every conv launch)
        ekms, ekl, ekf = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0)
        if timing:
            _lib.check(_lib.lib.sempyr_profile_begin(4096), "sempyr_profile_begin")
            for _ in range(2):
                eval_model.synthesize_batch(G, C, src, normalizer_clf)
            torch.cuda.synchronize()
            _lib.check(_lib.lib.sempyr_profile_end(ctypes.byref(ekms), ctypes.byref(ekl), ctypes.byref(ekf)), "sempyr_profile_end")
        if world > 1:
            t = torch.tensor([eval_ips], dtype=torch.float64, device='cpu' if torch.distributed.get_backend() == 'gloo' else device)
            torch.distributed.all_reduce(t)
            eval_ips = float(t.item())

    if rank == 0:
        iters = 5 * a.steps
        # SURVEY 8d per-sample forward GFLOP: res F_G 19.053 / F_D 1.703, default F_G 2.921 / F_D 1.577, classifier 4.738
        f_g, f_d = (F_G, F_D) if a.gen_type == 'res' else (2.921e9, 1.577e9)
        d_step = F_C + f_g + 12 * f_d
        g_step = (6 if a.use_diversity_loss else 3) * f_g + 2 * f_d + 2 * F_C
        cycle_flops = B * (5 * d_step + g_step)
        out = {
            "metric": "WGAN-GP train loop iterations/sec (train_model.py loop body; 1 D step per iteration, 1 G step every 5)",
            "value": iters * world / elapsed, "unit": "iters/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1000.0 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "host": {"issue_ms_per_step": 1000.0 * host_issue / a.steps, "cpu_time_per_step_ms": 1000.0 * cpu_time / a.steps},
            "config": {"workload": f"train_model.py cycle = 5 loop iterations (5 D steps + 1 G step), gen-type {a.gen_type}, "
                                   f"{a.discriminator_norm}, GP weight 10, batch {B}/GPU, 256x256"
                                   + (", diversity loss" if a.use_diversity_loss else ""),
                       "iters_per_step": 5, "batch_per_gpu": B, "parallelism": f"dp{world}", "loop_mode": a.loop_mode,
                       "weights": "seeded random init (trained blobs are not shipped)",
                       "layers_drawn_in_timed_steps": {str(k): timed_layers.count(k) for k in (1, 2, 3, 4)}},
        }
        if plain is not None:
            out["exact_shortcuts"] = {
                "what": "taps / feature-branch convolutions that the reference multiplies by all-zero masks (quirk Q2, "
                        "train_model.py:197-209) are not computed: generator._ZeroConv, train_model.live_taps; outputs, losses and "
                        "parameter updates are bit-identical with and without (tests/test_graph_loop_gpu.py)",
                "value_without": iters * world / plain, "ms_per_step_without": 1000.0 * plain / a.steps}
        if bf16x3 is not None:
            out["bf16x3"] = {
                "what": "opt-in SEMPYR_CONV_MATH=bf16x3: 3x3 stride-1 convolutions and their input gradients as split-bf16 products "
                        "(hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate); everything else fp32.  Images / losses "
                        "within 1e-3, gradients within 3x the fp32 path's bound (tests/test_bf16x3_gpu.py).  Not the headline.",
                "value": iters * world / bf16x3, "ms_per_step": 1000.0 * bf16x3 / a.steps, "unit": "iters/s"}
        if eval_ips is not None:
            out["eval_images_per_sec"] = eval_ips
            out["eval_config"] = "eval_model.py loop, gen-type res, batch 8, 4 levels x 4 noises, classifier pass included, 10 repetitions"
            if timing and ekl.value > 0:
                ach = ekf.value / (ekms.value * 1e-3) / 1e12
                out["eval_roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": None, "launches": int(ekl.value),
                                        "kernel_ms_total": ekms.value, "flops_counted_2MNK": ekf.value,
                                        "note": "conv kernels of 2 repetitions of the loop, HIP events around every launch"}
            if world == 1 and not a.no_cpu_baseline:
                out["eval_cpu_baseline"] = cpu_baseline_eval()
        if timing and kl.value > 0:
            # algorithmic FLOPs of the launches that ran (2*M*N*K on unpadded dims, summed by the library).  SURVEY 8(d)'s
            # per-cycle figure is quoted beside it; the executed count is ~18 % lower (3817 vs 4659 GFLOP per cycle)
            # because classifier layers and feature-branch convolutions behind all-zero masks (levels at or above the selected
            # layer, quirk Q2) are not executed, and the 1x1 skip convolutions run before the upsample.
            achieved = kf.value / (kms.value * 1e-3) / 1e12
            # HBM-side bytes per launch: PMC passes cannot run inside this process; the committed per-kernel summary of the
            # dominant family (tools/pmc_top3.sh -> profiles/r03_top3_traffic.json) is quoted; None if absent.
            traffic = traffic_note = None
            tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_top3_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    top = json.load(f)["families"][0]
                traffic = top["traffic_bytes_per_launch"]
                traffic_note = f"{top['name']}: measured {traffic / 1e6:.0f} MB vs algorithmic {top['algorithmic_bytes'] / 1e6:.0f} MB per launch"
            out["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": traffic, "traffic_of": traffic_note,
                               "kernel": "implicit-GEMM conv kernels on fp32 v_mfma_f32_32x32x2_f32 (igemm_kernel fwd/dgrad/wgrad, patch3x3_kernel, tpatch_kernel, wpatch_kernel; incl. the VALU RGB-head launches)",
                               "launches": int(kl.value), "avg_launch_us": 1000.0 * kms.value / kl.value,
                               "kernel_ms_total": kms.value, "kernel_time_share": kms.value * 1e-3 / elapsed_instr,
                               "instrumented_ms_per_step": 1000.0 * elapsed_instr / a.steps,
                               "note": "durations from HIP events around every conv launch over a second pass of the same "
                                       "steps, run single-stream so that each duration is the kernel's own (the headline "
                                       "pass overlaps branches and weight-gradient kernels on extra streams and records "
                                       "no events)",
                               "flops_per_cycle_survey": cycle_flops, "flops_counted_2MNK_per_cycle": kf.value / a.steps,
                               "achieved_on_survey_flops": cycle_flops * a.steps / (kms.value * 1e-3) / 1e12,
                               "table": "profiles/r03_conv_family_table.csv (rocprofv3 kernel trace of the same single-stream pass)"}
        if hbm_rows:
            # the HBM-bound kernels of the same instrumented pass: every kernel with >= 0.2 ms per cycle, the three furthest
            # below the ~8 TB/s peak first (tensors of <= 268 MB partly live in the 256 MB Infinity Cache between producer and
            # consumer, so some rates exceed what HBM alone gives; tiny finalize kernels are latency-, not bandwidth-bound)
            big = [r for r in hbm_rows if r["ms_per_cycle"] >= 0.2]
            # a kernel that moves less than 1 MB per launch finishes in 3-8 us whatever its bytes: it is launch-latency bound and
            # its "bandwidth" says nothing; the worst three are taken among the launches that can be bandwidth bound at all
            streaming = [r for r in big if r["algorithmic_MB_per_launch"] >= 1.0]
            out["hbm"] = {"peak_GBps": HBM_PEAK_GBPS, "unit": "GB/s",
                          "bytes": "algorithmic: 4 B x (elements read once + written once) per launch, counted at the launch site",
                          "worst_three": sorted(streaming, key=lambda r: r["frac_of_peak"])[:3],
                          "latency_bound_under_1MB_per_launch": [r["kernel"] for r in big if r["algorithmic_MB_per_launch"] < 1.0],
                          "kernels_over_0.2ms_per_cycle": big,
                          "other_kernels_ms_per_cycle": sum(r["ms_per_cycle"] for r in hbm_rows),
                          "table": "profiles/r03_hbm_kernels.json (rocprofv3 durations of the same single-stream pass)"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, a.use_diversity_loss)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
