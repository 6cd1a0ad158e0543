#!/bin/bash
# single-stream rocprofv3 kernel trace of the timed cycles + the library's per-launch log -> profiles/<tag>_* tables
TAG=${1:-r02}; STEPS=${2:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_pass_$TAG
rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export SEMPYR_PROFILE_DUMP=$O/launches.csv
export SEMPYR_HBM_DUMP=$O/hbm.json
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O -o trace -- python3 $R/bench.py --profile-pass --steps $STEPS --no-cpu-baseline --no-eval > $O/log.txt 2>&1 || { echo "profile pass failed"; tail -5 $O/log.txt; exit 1; }
tail -1 $O/log.txt
T=$(find $O -name 'trace_kernel_trace.csv' | head -1)
cd $R && python3 tools/roofline_table.py $T $O/launches.csv gpurun_out/$TAG && cp $O/log.txt gpurun_out/${TAG}_profile_pass_log.txt
python3 tools/hbm_kernels.py $T $O/hbm.json gpurun_out/$TAG
