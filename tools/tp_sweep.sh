#!/bin/bash
# variants of the 4x4 patch kernels on the critic's shapes (tuning aid)
O=gpurun_out/tp_sweep.log; : > $O
run() { echo "== $*" >> $O; env "$@" timeout -k 10 120 python tools/tp_check.py nocheck $ARGS 2>&1 | grep -v amdgpu.ids >> $O || exit 1; }
ARGS=""
run SEMPYR_TP_KI=2 SEMPYR_TP_BN=64
run SEMPYR_TP_KI=2 SEMPYR_TP_BN=128
run SEMPYR_TP_KI=2 SEMPYR_TP_BN=256
run SEMPYR_TP_KI=4 SEMPYR_TP_BN=64
run SEMPYR_TP_KI=4 SEMPYR_TP_BN=128
run SEMPYR_TP_KI=4 SEMPYR_TP_BN=256
ARGS="ksweep"
run SEMPYR_TP_KI=2
run SEMPYR_TPATCH=0
