"""Summarise SEMPYR_TP_STAMPS timelines (100 MHz ticks): per launch, median/max of each phase over workgroups (tuning aid)."""
import sys, statistics
cur = None; rows = []
def flush():
    if not rows: return
    n = len(rows)
    st = [[r[k] for r in rows] for k in range(5)]
    med = lambda v: statistics.median(v) / 100.0
    print(cur.strip())
    print('  wgs %d | start: med %.1f max %.1f us | prologue %.1f | first load+store %.1f | main %.1f | epilogue %.1f | end: med %.1f max %.1f' % (
        n, med(st[0]), max(st[0]) / 100.0, med([r[1] - r[0] for r in rows]), med([r[2] - r[1] for r in rows]),
        med([r[3] - r[2] for r in rows]), med([r[4] - r[3] for r in rows]), med(st[4]), max(st[4]) / 100.0))
for line in open(sys.argv[1]):
    if line.startswith('#'):
        flush(); cur = line; rows = []
    else:
        rows.append([int(v) for v in line.split()[1:]])
flush()
