#!/bin/bash
# rocprofv3 kernel durations of tools/tp_check.py under a set of env variants (tuning aid): bash tools/tp_prof.sh "<args>" VAR=.. -- VAR=.. 
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/tp_prof; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="$1"; shift
i=0
IFS='|' read -ra VARIANTS <<< "$*"
for v in "${VARIANTS[@]}"; do
  i=$((i+1))
  echo "== variant $i: $v" >> $O/summary.txt
  for kv in $v; do export "$kv"; done
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O -o v$i -- python3 $R/tools/tp_check.py nocheck $ARGS > $O/log$i.txt 2>&1 || { echo "variant $i failed"; tail -5 $O/log$i.txt; exit 1; }
  for kv in $v; do unset "${kv%%=*}"; done
  T=$(find $O -name "v${i}_kernel_trace.csv" | head -1)
  python3 - "$T" >> $O/summary.txt <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
runs = []
for r in rows:
    n = r['Kernel_Name']
    if not any(k in n for k in ('tpatch_kernel', 'igemm_kernel', 'splitk_reduce', 'wgrad_reduce', 'wpatch')): continue
    m = re.search(r'(tpatch_kernel|igemm_kernel|splitk_reduce\w*|wgrad_reduce\w*|wpatch\w*)(<[^>]*>)?', n)
    key = (m.group(1) + (m.group(2) or ''), r['Grid_Size_X'])
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    # a timing loop alternates (conv, reduce): group by key within a window of consecutive launches
    for run in runs[-2:]:
        if run[0] == key: run[1].append(d); break
    else:
        runs.append((key, [d]))
for (k, g), d in runs:
    if len(d) < 5: continue
    d = sorted(d)
    print('%-52s grid %8s n=%3d median %8.2f us min %8.2f' % (k, g, len(d), d[len(d) // 2], d[0]))
PY
done
cat $O/summary.txt
