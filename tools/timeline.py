"""Where a cycle's wall time goes, from a multi-stream rocprofv3 kernel trace of `SEMPYR_MARK_TIMED=1 python bench.py ...`:
conv-active time, time with only other kernels running (per kernel: what is NOT under a convolution), idle gaps.
python tools/timeline.py <kernel_trace.csv> <steps>"""
import bisect, collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'marker' in r['Kernel_Name']]
a, b = marks[0], marks[-1]
t0, t1 = int(rows[a]['End_Timestamp']), int(rows[b]['Start_Timestamp'])
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows[a + 1:b]]
isconv = lambda n: any(k in n for k in ('igemm_kernel', 'patch3x3', 'tpatch_kernel', 'wpatch_kernel', 'rgb3x3', 'gather_small', 'small_m_cmajor', 'rgb_wgrad3x3', 'stem3x3'))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n)
    m = re.match(r'at::native::\w+<[^,]*, at::native::(\w+)', n)
    return ('torch:' + m.group(1)) if m else n.split('(')[0][:56]
def union(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
conv = union([(s, e) for s, e, n in ev if isconv(n)])
anyb = union([(s, e) for s, e, n in ev])
starts = [x[0] for x in conv]
def covered(s, e):
    i = max(bisect.bisect_right(starts, s) - 1, 0); c = 0
    while i < len(conv) and conv[i][0] < e:
        lo, hi = max(s, conv[i][0]), min(e, conv[i][1])
        if hi > lo: c += hi - lo
        i += 1
    return c
ms = lambda v: v / 1e6 / steps
tot_conv, tot_any = sum(e - s for s, e in conv), sum(e - s for s, e in anyb)
print('per cycle: wall %.2f ms | some kernel running %.2f (idle %.2f) | a conv kernel running %.2f | only other kernels %.2f | serial sum conv %.2f other %.2f | launches %d' % (
    ms(t1 - t0), ms(tot_any), ms(t1 - t0 - tot_any), ms(tot_conv), ms(tot_any - tot_conv),
    ms(sum(e - s for s, e, n in ev if isconv(n))), ms(sum(e - s for s, e, n in ev if not isconv(n))), len(ev) // steps))
exp = collections.defaultdict(lambda: [0, 0, 0])
for s, e, n in ev:
    if isconv(n): continue
    k = short(n); exp[k][0] += 1; exp[k][1] += e - s; exp[k][2] += e - s - covered(s, e)
print('kernels that run while no conv kernel is active (per cycle):')
for k, v in sorted(exp.items(), key=lambda kv: -kv[1][2])[:28]:
    print('  %-56s n=%5.0f total %6.2f ms  not under a conv %6.2f ms' % (k, v[0] / steps, ms(v[1]), ms(v[2])))
gaps = [(anyb[i + 1][0] - anyb[i][1]) for i in range(len(anyb) - 1)]
print('idle gaps per cycle: <2us %.2f ms (%d), 2-10us %.2f ms (%d), >10us %.2f ms (%d)' % (
    ms(sum(g for g in gaps if g < 2000)), sum(1 for g in gaps if g < 2000) // steps, ms(sum(g for g in gaps if 2000 <= g < 10000)),
    sum(1 for g in gaps if 2000 <= g < 10000) // steps, ms(sum(g for g in gaps if g >= 10000)), sum(1 for g in gaps if g >= 10000) // steps))
