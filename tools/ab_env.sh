#!/bin/bash
# A/B of one environment switch on the headline pass: bash tools/ab_env.sh VAR valueA valueB [steps]
V=$1; A=$2; B=$3; STEPS=${4:-6}
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R
for X in $A $B $A $B; do
env $V=$X timeout -k 10 300 python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-eval --no-kernel-timing --no-bf16x3 > gpurun_out/ab_$X.json 2> gpurun_out/ab_$X.err || exit 1
python - <<PY
import json; d=json.loads(open('gpurun_out/ab_$X.json').read().strip().splitlines()[-1]); print('$V=$X', round(d['value'],2), round(d['ms_per_step'],3), d['exact_shortcuts'].get('value_without'))
PY
done
