for rep in 1 2; do
for cfg in "1 1 1" "1 1 0" "1 0 0" "0 0 0"; do set -- $cfg
echo "ASYNC_WGRAD=$1 TWO_STREAM=$2 TWO_STREAM_G=$3: $(SEMPYR_ASYNC_WGRAD=$1 SEMPYR_TWO_STREAM=$2 SEMPYR_TWO_STREAM_G=$3 timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-eval --no-kernel-timing 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")"
done; done
