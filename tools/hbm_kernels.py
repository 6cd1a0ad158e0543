"""Achieved bandwidth of the HBM-bound kernels of the bench cycle against the ~8 TB/s peak.

    bash tools/profile_pass.sh r03 2      # single-stream rocprofv3 kernel trace of `bench.py --profile-pass` + the library's logs
    python tools/hbm_kernels.py <trace_kernel_trace.csv> <hbm.json> profiles/r03

Algorithmic bytes per launch come from the library (HbmScope at every launch site: 4 B x elements read once + written once,
SURVEY.md 8d), durations from the ROCPROF trace of the same pass (rows between the two sempyr_marker_kernel launches); the
HIP-event durations bench.py reports in its `hbm` block are listed beside them."""
import collections, csv, json, re, sys
trace, dump, out = sys.argv[1], sys.argv[2], sys.argv[3]
lib = json.load(open(dump))
steps = lib["steps"]
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'sempyr_marker_kernel' in r['Kernel_Name']]
rows = rows[marks[0] + 1:marks[1]]
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    return n.split('(')[0]
dur = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = dur[short(r['Kernel_Name'])]
    d[0] += 1; d[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
def match(libname):
    """trace kernel names that a library scope covers"""
    if libname.startswith('wgrad_reduce'): return [k for k in dur if k.startswith('wgrad_reduce')]
    if libname.startswith('splitk_reduce'): return [k for k in dur if k.startswith('splitk_reduce')]
    if libname == 'instnorm_fwd_kernel': return [k for k in dur if k.startswith('instnorm_fwd')]
    if libname == 'instnorm_bwd_kernel': return [k for k in dur if k.startswith('instnorm_bwd')]
    return [k for k in dur if k == libname]
res = []
for k in lib["kernels"]:
    names = match(k["kernel"])
    n = sum(dur[x][0] for x in names); ms = sum(dur[x][1] for x in names)
    if not n:
        continue
    nbytes = k["algorithmic_MB_per_launch"] * 1e6 * k["launches_per_cycle"] * steps
    res.append({"kernel": k["kernel"], "launches_per_cycle": n / steps, "algorithmic_MB_per_launch": round(nbytes / n / 1e6, 3),
                "rocprof_avg_us": round(1e3 * ms / n, 2), "rocprof_ms_per_cycle": round(ms / steps, 3),
                "hip_event_ms_per_cycle": round(k["ms_per_cycle"], 3),
                "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 0), "frac_of_peak": round(nbytes / (ms * 1e-3) / 1e9 / 8000.0, 3)})
res.sort(key=lambda r: -r["rocprof_ms_per_cycle"])
covered = set(x for k in lib["kernels"] for x in match(k["kernel"]))
conv = ('igemm_kernel', 'patch3x3', 'tpatch_kernel', 'wpatch_kernel', 'rgb3x3', 'gather_small_m', 'small_m_cmajor', 'rgb_wgrad3x3', 'sempyr_marker')
rest = sorted(((k, v) for k, v in dur.items() if k not in covered and not any(c in k for c in conv)), key=lambda kv: -kv[1][1])
for r in res:
    print('%-36s n/cycle %6.1f  %9.3f MB  %8.2f us  %6.3f ms/cycle  %6.0f GB/s  %.3f' % (
        r["kernel"], r["launches_per_cycle"], r["algorithmic_MB_per_launch"], r["rocprof_avg_us"], r["rocprof_ms_per_cycle"], r["achieved_GBps"], r["frac_of_peak"]))
json.dump({"source": "rocprofv3 --kernel-trace of `bench.py --profile-pass` (single stream), same trace as the conv family table",
           "peak_GBps": 8000, "steps": steps,
           "bytes": "algorithmic, counted by the library at the launch site: 4 B x (elements read once + elements written once)",
           "note": "tensors of <= 268 MB (16 samples x 64 ch x 256^2) partly live in the 256 MB Infinity Cache between producer and "
                   "consumer, so some rates exceed what HBM alone gives; the finalize / fold kernels move kilobytes and are "
                   "latency-bound (their rate says nothing about bandwidth); MFMA-bound non-igemm kernels (stem3x3_*) are in "
                   "`not_hbm_bound_ms_per_cycle`",
           "kernels": res,
           "not_covered_ms_per_cycle": {k: round(v[1] / steps, 3) for k, v in rest[:20]}}, open(out + '_hbm_kernels.json', 'w'), indent=1)
