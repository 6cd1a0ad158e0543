#!/bin/bash
# kernel-only durations of the weight-gradient kernels: bash tools/wp_prof.sh <tag> [args of wp_check.py]
TAG=${1:-wp}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/wpprof_$TAG
rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $R/tools/wp_check.py nocheck "$@" > $O/log.txt 2>&1 || { echo failed; tail -5 $O/log.txt; exit 1; }
S=$(find $O -name 't_kernel_stats.csv' | head -1)
python3 - "$S" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name']).replace('void ', '').split('(')[0]
    if any(k in n for k in ('wpatch', 'igemm', 'wgrad_reduce')):
        print('%-60s calls %5s avg us %9.1f min %9.1f max %9.1f' % (n[:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
