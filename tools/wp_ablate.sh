#!/bin/bash
# per-workgroup timeline of the row-ring weight-gradient kernel (SEMPYR_WP_STAMPS) on two shapes -> tools/wp_stamps.py summary
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out; cd $R
rm -f gpurun_out/wp_stamps.txt
SEMPYR_WPATCH=7 SEMPYR_WP_STAMPS=gpurun_out/wp_stamps.txt timeout -k 10 120 python tools/wp_check.py nocheck one > gpurun_out/wp_ab.txt 2>&1 || exit 1
python tools/wp_stamps.py gpurun_out/wp_stamps.txt | tail -4
