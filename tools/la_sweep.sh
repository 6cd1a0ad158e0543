set -e
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -u -m pytest tests/test_graph_loop_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/la_test.log 2>&1
for L in 1 2 3; do
SEMPYR_LOOKAHEAD=$L timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-eval --no-kernel-timing > gpurun_out/la_bench_$L.json 2> gpurun_out/la_bench_$L.err
done
