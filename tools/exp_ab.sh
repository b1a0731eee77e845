#!/bin/bash
# Run on the GPU box: scan time of several library builds, alternating (clock drift and box-to-box differences are a few
# per cent: builds are compared within one call, interleaved).  LIBS="libtatajuba_amd.so libtatajuba_amd_x.so" ROUNDS=3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
: > $O/ab.log
for r in $(seq 1 ${ROUNDS:-3}); do
  for lib in ${LIBS:-libtatajuba_amd.so}; do
    TJ_DIAG_LIB=$lib TJ_REPS=${TJ_REPS:-6} TJ_NORAW=1 timeout -k 5 120 python tools/exp_scan_only.py 2>&1 | tail -1 >> $O/ab.log
  done
done
cat $O/ab.log
python - <<PY
import re, collections
best = collections.defaultdict(list)
for ln in open("$O/ab.log"):
    m = re.match(r"(\S+) fast \S+ scan ms (.*) raw", ln)
    if m: best[m.group(1)].append(min(float(x) for x in m.group(2).split()[1:]))
for k, v in best.items(): print(k, "min ms per round:", " ".join("%.3f" % x for x in v), " best %.3f" % min(v))
PY
