"""GPU busy time (union of kernel intervals over all streams) inside the timed region of a multi-stream rocprofv3 kernel
trace of `SEMPYR_MARK_TIMED=1 python bench.py ...`, the idle gaps, and what ran right after the largest gaps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'marker' in r['Kernel_Name']]
a, b = marks[0], marks[-1]
t0, t1 = int(rows[a]['End_Timestamp']), int(rows[b]['Start_Timestamp'])
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60]) for r in rows[a + 1:b]]
busy = 0; cur_s = cur_e = None; gaps = []
for s, e, n in iv:
    if cur_e is None: cur_s, cur_e = s, e; continue
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n, cur_e - t0)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
ser = sum(e - s for s, e, n in iv)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print(f"span {(t1-t0)/1e6/steps:.2f} ms/cycle  busy(union) {busy/1e6/steps:.2f}  idle {(t1-t0-busy)/1e6/steps:.2f}  serial sum {ser/1e6/steps:.2f}  kernels {len(iv)//steps}")
gaps.sort(reverse=True)
print("gaps > 20us:", sum(1 for g in gaps if g[0] > 20000) // steps, "per cycle, total", sum(g[0] for g in gaps if g[0] > 20000) / 1e6 / steps, "ms/cycle")
print("gaps 5-20us:", sum(1 for g in gaps if 5000 < g[0] <= 20000) // steps, "per cycle, total", sum(g[0] for g in gaps if 5000 < g[0] <= 20000) / 1e6 / steps, "ms/cycle")
print("gaps < 5us total", sum(g[0] for g in gaps if g[0] <= 5000) / 1e6 / steps, "ms/cycle, count", sum(1 for g in gaps if g[0] <= 5000) // steps)
for g in gaps[:25]:
    print(f"  {g[0]/1e3:8.1f} us before {g[1]}  at {g[2]/1e6:.2f} ms")
