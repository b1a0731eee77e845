#!/bin/bash
# time of the fast scan kernel in ablated builds (tools/build_exp.sh <sfx> -DFK_EXP_STOP=N / -DTJ_EXP_SINK=N) and whole
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for lib in ${LIBS:-libtatajuba_amd_fstop1.so libtatajuba_amd_fstop2.so libtatajuba_amd_fstop3.so libtatajuba_amd.so}; do
  TJ_DIAG_LIB=$lib timeout -k 5 90 python3 $R/tools/exp_scan_only.py 2>&1 | tail -1
done
