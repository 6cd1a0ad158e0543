#!/bin/bash
# eager vs replay vs hipgraph on the bench workload (ms per cycle, host issue time)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
for m in ${MODES:-eager replay}; do
  timeout -k 10 300 python -u bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-eval --no-kernel-timing --loop-mode $m > $O/ab_$m.log 2> $O/ab_$m.err || { echo "$m failed"; tail -5 $O/ab_$m.err; exit 1; }
  python3 - "$O/ab_$m.log" "$m" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'ms/cycle %.2f'%d['ms_per_step'], 'iters/s %.1f'%d['value'], 'host issue %.1f'%d['host']['issue_ms_per_step'], 'without shortcuts %.1f'%d.get('exact_shortcuts',{}).get('ms_per_step_without',0))
PY
done
