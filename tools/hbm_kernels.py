"""Achieved bandwidth of the HBM-bound kernels on the largest tensors of the bench cycle, from a rocprofv3 kernel trace
(gpurun_out/prof/bench_kernel_trace.csv).  Algorithmic bytes per launch are derived from the launch grid (threads x
elements per thread) and the kernel's reads / writes per element (fp32), i.e. SURVEY.md 8d's per-element figures.
Writes profiles/<tag>_hbm_kernels.json."""
import collections, csv, json, sys
path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof/bench_kernel_trace.csv'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r01'
# kernel substring -> (elements per thread, bytes per element, what is counted)
K = {
    'affine_act_kernel': (4, 8, 'read x + write y'),
    'affine_act_bwd_kernel': (4, 12, 'read gy, y + write gx'),
    'norm_bwd_apply_kernel': (4, 16, 'read gy, y, x + write gx'),
    'pool2_up_v2_kernel': (2, 36, 'per coarse pixel: read 1 + 4 (yref), write 4'),
    'pool2_down_v2_kernel': (2, 36, 'per coarse pixel: read 4 + 4 (yref), write 1'),
    'row_moments_kernel<256>': (1.0 / 256, None, 'read x once (one workgroup per row)'),
}
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    for k, (ept, bpe, what) in K.items():
        if k in n and bpe is not None:
            grid = int(r['Grid_Size_X'])
            agg[(k, grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = []
for (k, grid), d in sorted(agg.items(), key=lambda kv: -kv[0][1]):
    ept, bpe, what = K[k]
    nbytes = grid * ept * bpe
    if nbytes < 100e6:
        continue
    us = sum(d) / len(d)
    out.append({'kernel': k, 'grid_threads': grid, 'launches': len(d), 'algorithmic_MB': round(nbytes / 1e6, 1),
                'avg_us': round(us, 1), 'achieved_GBps': round(nbytes / us / 1e3, 0), 'bytes_counted': what})
for o in out:
    print(o)
json.dump({'source': 'rocprofv3 --kernel-trace of bench.py (same command as r01_bench_kernel_stats.csv)', 'peak_GBps': 8000,
           'note': 'tensors of 268 MB (16 samples x 64 ch x 256^2) partly live in the 256 MB Infinity Cache between producer and '
                   'consumer, so some rates exceed what HBM alone would give; 537 MB tensors (32 samples) do not',
           'kernels': out}, open(f'profiles/{tag}_hbm_kernels.json', 'w'), indent=1)
