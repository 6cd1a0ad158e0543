"""Which kernels run while no convolution kernel is active?  Reads a rocprofv3 kernel trace of bench.py and splits the cycle
between the 6th and 12th optimizer step into conv-active time and the rest (per kernel name: total and 'exposed' time)."""
import bisect, collections, csv, re, sys
path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof/bench_kernel_trace.csv'
rows = list(csv.DictReader(open(path)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Stream_Id']) for r in rows)
marks = [e[1] for e in ev if 'adam_kernel' in e[2]]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a, b = marks[6 * first - 1], marks[6 * first + 5]
isconv = lambda n: any(k in n for k in ('igemm_kernel', 'patch3x3', 'rgb3x3', 'gather_small'))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    m = re.match(r'at::native::\w+<[^,]*, at::native::(\w+)', n)
    if m: return 'torch:' + m.group(1)
    return n.split('(')[0][:48]
sub = [e for e in ev if e[1] > a and e[0] < b]
m = []
for s, e, n, _ in sorted(x for x in sub if isconv(x[2])):
    if m and s <= m[-1][1]: m[-1][1] = max(m[-1][1], e)
    else: m.append([s, e])
starts = [x[0] for x in m]
def covered(s, e):
    i = max(bisect.bisect_right(starts, s) - 1, 0); c = 0
    while i < len(m) and m[i][0] < e:
        lo, hi = max(s, m[i][0]), min(e, m[i][1])
        if hi > lo: c += hi - lo
        i += 1
    return c
anyb = []
for s, e, n, _ in sub:
    if anyb and s <= anyb[-1][1]: anyb[-1][1] = max(anyb[-1][1], e)
    else: anyb.append([s, e])
exp = collections.defaultdict(lambda: [0, 0, 0])
for s, e, n, st in sub:
    if isconv(n): continue
    s, e = max(s, a), min(e, b)
    k = short(n)
    exp[k][0] += 1; exp[k][1] += e - s; exp[k][2] += e - s - covered(s, e)
print('cycle wall %.1f ms, any kernel active %.1f, conv active %.1f' % ((b - a) / 1e6, sum(y - x for x, y in anyb) / 1e6, sum(y - x for x, y in m) / 1e6))
for k, v in sorted(exp.items(), key=lambda kv: -kv[1][2])[:30]:
    print('%-50s n=%4d total %6.2f ms  not under a conv %6.2f ms' % (k, v[0], v[1] / 1e6, v[2] / 1e6))
