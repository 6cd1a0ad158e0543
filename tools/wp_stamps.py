"""Summary of SEMPYR_WP_STAMPS dumps (100 MHz s_memrealtime ticks): python tools/wp_stamps.py <file>"""
import sys, statistics as st
launch = None; rows = []
def flush():
    if not rows: return
    med = lambda k: st.median([r[k] for r in rows if r[k] >= 0]) / 100.0
    end = max(r[7] for r in rows) / 100.0
    print(launch.strip())
    print('   start %.1f | prologue done %.1f | step1 %.1f step2 %.1f step3 %.1f step4 %.1f | loop done %.1f | end med %.1f max %.1f us  (medians over %d workgroups)' % (
        med(0), med(1), med(2) - med(1), med(3) - med(2), med(4) - med(3), med(5) - med(4), med(6), med(7), end, len(rows)))
for line in open(sys.argv[1]):
    if line.startswith('#'):
        flush(); launch = line; rows = []
    else:
        v = [int(x) for x in line.split()]; rows.append(v[1:])
flush()
