#!/bin/bash
# main-loop ablation of the 4x4 patch kernels via SEMPYR_TP_DBG (results of dbg != 0 are garbage; timelines only)
for d in 0 1 2 3; do
  rm -f gpurun_out/stamps_$d.txt
  SEMPYR_TP_DBG=$d SEMPYR_TP_STAMPS=gpurun_out/stamps_$d.txt timeout -k 10 200 python tools/tp_check.py nocheck > /dev/null 2>&1 || exit 1
  echo "=== dbg $d"
  python tools/tp_stamps.py gpurun_out/stamps_$d.txt | awk '/^# launch/{k=$0; c[k]++; getline; if (c[k]==3) {print k; print $0}}'
done
