#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R
for D in 0; do
rm -f gpurun_out/wp_stamps_$D.txt
SEMPYR_WP_DBG=$D SEMPYR_WP_STAMPS=gpurun_out/wp_stamps_$D.txt timeout -k 10 120 python tools/wp_check.py nocheck one > gpurun_out/wp_ab_$D.txt 2>&1 || exit 1
echo "dbg=$D"; python tools/wp_stamps.py gpurun_out/wp_stamps_$D.txt | tail -4
done
