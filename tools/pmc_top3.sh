#!/bin/bash
# HBM-side traffic PER KERNEL for the three conv families with the largest share of the cycle (two PMC passes each:
# FETCH_SIZE and WRITE_SIZE do not fit one pass), summarised by tools/pmc_top3.py
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_top3
rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
while read -r name args; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O -o f${i}_$c -- python3 $R/tools/one_conv.py $args > $O/log_${i}_$c.txt 2>&1 || { echo "pass $name $c failed"; tail -3 $O/log_${i}_$c.txt; exit 1; }
  done
  echo "$i $name $args" >> $O/families.txt
  i=$((i+1))
done <<'LIST'
G.b7.conv2_64to64_3x3_at256_fwd fwd 16 64 256 256 64 3 1 1 0
G.b7.conv1_64to64_3x3_ups128to256_fwd fwd 16 64 128 128 64 3 1 1 1
G.b7.conv2_wgrad_64x576xK1M wgrad 16 64 256 256 64 3 1 1 0
G.b7.conv1_wgrad_ups_64x576xK1M wgrad 16 64 128 128 64 3 1 1 1
D.b1_64to64_4x4s2_at128_fwd fwd 16 64 128 128 64 4 2 1 0
D.b3_128to256_4x4s2_at32_dgrad dgrad 16 128 32 32 256 4 2 1 0
LIST
cd $R && python3 tools/pmc_top3.py
