"""Per-kernel-family roofline table of the conv kernels from a rocprofv3 kernel trace of `bench.py --profile-pass`.

    SEMPYR_PROFILE_DUMP=gpurun_out/r02_launches.csv rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pass -- \
        python3 bench.py --profile-pass --steps 2 --no-cpu-baseline --no-eval
    python tools/roofline_table.py <kernel_trace.csv> gpurun_out/r02_launches.csv profiles/r02

The pass is single-stream, so the trace rows of the conv kernels between the two `sempyr_marker_kernel` launches are in
issue order: row i is launch i of the library's own per-launch log (shape M, N, K and 2MNK FLOPs).  Output: a table per
family (kernel template, M, N, K) with the ROCPROF average duration, TFLOP/s and fraction of the 157.3 TFLOP/s fp32 MFMA
peak, plus totals (which is what bench.py's roofline.frac must reproduce) and a per-kernel-name summary of everything else
in the timed region (HBM-bound kernels)."""
import collections, csv, json, re, sys

trace, launches, out = sys.argv[1], sys.argv[2], sys.argv[3]
PEAK = 157.3
CONV = ('igemm_kernel', 'patch3x3_kernel', 'tpatch_kernel', 'wpatch_kernel', 'rgb3x3_kernel', 'gather_small_m_kernel', 'small_m_cmajor_kernel', 'rgb_wgrad3x3_kernel')
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'sempyr_marker_kernel' in r['Kernel_Name']]
assert len(marks) >= 2, 'marker kernels not found'
rows = rows[marks[0] + 1:marks[1]]
span_ms = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) * 1e-6
conv = [r for r in rows if any(k in r['Kernel_Name'] for k in CONV)]
recs = list(csv.DictReader(open(launches)))
assert len(conv) == len(recs), (len(conv), len(recs))


def short(name):
    m = re.search(r'(igemm_kernel|patch3x3_kernel|tpatch_kernel|wpatch_kernel|rgb3x3_kernel|gather_small_m_kernel|small_m_cmajor_kernel|rgb_wgrad3x3_kernel)(<[^>]*>)?', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:60]


fam = collections.OrderedDict()
for r, q in zip(conv, recs):
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
    key = (short(r['Kernel_Name']), int(q['M']), int(q['N']), int(q['K']), int(q['splits']), int(q['nphases']))
    f = fam.setdefault(key, [0, 0.0, 0.0, 0.0])
    f[0] += 1; f[1] += dur; f[2] += float(q['gflop']); f[3] += float(q['ms'])
tot_ms = sum(f[1] for f in fam.values()); tot_gf = sum(f[2] for f in fam.values()); tot_ev = sum(f[3] for f in fam.values())
with open(out + '_conv_family_table.csv', 'w') as f:
    f.write('kernel,M,N,K,splits,phases,launches,gflop_per_launch,rocprof_avg_us,hip_event_avg_us,tflops,frac_of_peak,share_of_conv_time\n')
    for key, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        f.write('"%s",%d,%d,%d,%d,%d,%d,%.4f,%.2f,%.2f,%.1f,%.3f,%.4f\n' % (key + (v[0], v[2] / v[0], 1e3 * v[1] / v[0], 1e3 * v[3] / v[0],
                                                                        v[2] / v[1], v[2] / v[1] / PEAK, v[1] / tot_ms)))
other = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if not any(k in r['Kernel_Name'] for k in CONV):
        nm = re.sub(r'^void ', '', r['Kernel_Name']).replace('(anonymous namespace)::', '')
        o = other[re.sub(r'\(.*', '', nm)[:100]]
        o[0] += 1; o[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
summary = {'peak_tflops': PEAK, 'conv_launches': len(conv), 'conv_ms_rocprof': tot_ms, 'conv_ms_hip_events': tot_ev, 'conv_gflop_2MNK': tot_gf,
           'tflops_rocprof': tot_gf / tot_ms, 'frac_rocprof': tot_gf / tot_ms / PEAK, 'timed_span_ms': span_ms,
           'other_kernels_ms': sum(o[1] for o in other.values()), 'other_kernels': {k: {'launches': v[0], 'ms': v[1]} for k, v in sorted(other.items(), key=lambda kv: -kv[1][1])}}
json.dump(summary, open(out + '_conv_roofline_summary.json', 'w'), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != 'other_kernels'}))
for k, v in list(summary['other_kernels'].items())[:25]:
    print('%8.3f ms %5d  %s' % (v['ms'], v['launches'], k))
