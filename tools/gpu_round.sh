#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel stats. Every GPU step has its own time limit and
# nothing runs after a fault / abort / timeout.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -v --timeout 300 --timeout-method thread -p no:cacheprovider > $O/tests.log 2>&1
rc=$?; echo "tests_exit=$rc" | tee -a $O/tests.log
grep -E "passed|failed" $O/tests.log | tail -3
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
rc=$?; echo "smoke_exit=$rc"; tail -2 $O/smoke.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
SEMPYR_PROFILE_DUMP=$O/igemm_launches.csv timeout -k 10 900 python -u bench.py --steps ${BENCH_STEPS:-3} --warmup 1 > $O/bench.log 2> $O/bench.err
rc=$?; echo "bench_exit=$rc"; tail -1 $O/bench.log; tail -3 $O/bench.err
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-eval > $O/rocprof.log 2>&1
rc=$?; echo "rocprof_exit=$rc"; tail -2 $O/rocprof.log
ls -R $O/prof | head -20
