#!/bin/bash
# multi-stream kernel trace of the headline pass -> tools/timeline.py summary: bash tools/trace_bench.sh <tag> [steps]
TAG=${1:-t}; STEPS=${2:-4}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/trace_$TAG
rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export SEMPYR_MARK_TIMED=1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O -o trace -- python3 $R/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-eval --no-kernel-timing --no-bf16x3 > $O/log.txt 2>&1 || { echo "trace failed"; tail -5 $O/log.txt; exit 1; }
T=$(find $O -name 'trace_kernel_trace.csv' | head -1)
cd $R && python3 tools/timeline.py $T $STEPS | tee gpurun_out/timeline_$TAG.txt
