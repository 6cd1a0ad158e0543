# other BASELINE configs at full size: diversity loss, default generator, batch_norm / layer_norm critics
run() { echo "== $*"; timeout -k 10 280 python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-eval --no-bf16x3 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2),'iters/s', round(d['ms_per_step'],2),'ms/cycle  frac', round(d['roofline']['frac'],3), ' without shortcuts', round(d['exact_shortcuts']['value_without'],2), ' host issue', round(d['host']['issue_ms_per_step'],1))"; }
run --use-diversity-loss
run --gen-type default --discriminator-norm batch_norm
run --gen-type default --discriminator-norm layer_norm
run --gen-type res --discriminator-norm layer_norm
