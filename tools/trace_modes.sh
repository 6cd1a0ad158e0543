#!/bin/bash
# multi-stream rocprofv3 kernel trace of the timed cycles in eager and replay mode -> busy-union / gap statistics
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/trace_modes; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for m in ${MODES:-eager replay}; do
  SEMPYR_MARK_TIMED=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$m -o t -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-eval --no-kernel-timing --loop-mode $m > $O/$m.log 2>&1 || { echo "$m failed"; tail -5 $O/$m.log; exit 1; }
  T=$(find $O/$m -name 't_kernel_trace.csv' | head -1)
  echo "== $m"; python3 $R/tools/busy_union.py $T 4 | head -8
  python3 - $T <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
marks=[i for i,r in enumerate(rows) if 'marker' in r['Kernel_Name']]
a,b=marks[0],marks[-1]
q=collections.Counter(); qt=collections.Counter()
for r in rows[a+1:b]:
    k=(r.get('Queue_Id'),r.get('Stream_Id'))
    q[k]+=1; qt[k]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k in q: print('  queue/stream',k,'kernels',q[k]//4,'busy ms/cycle %.2f'%(qt[k]/4e6))
PY
  gzip -f $T
done
