#!/bin/bash
# HBM traffic of the implicit-GEMM kernels over one bench cycle: two PMC passes (FETCH_SIZE, WRITE_SIZE need separate
# runs), summarised per launch by tools/pmc_traffic.py into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_traffic
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 600 rocprofv3 --pmc $c --output-format csv -d $O -o $c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-eval --no-kernel-timing > $O/log_$c.txt 2>&1 || { echo "pass $c failed"; tail -3 $O/log_$c.txt; exit 1; }
done
ls $O
