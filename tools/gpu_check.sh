#!/bin/bash
# Run on the GPU box (gpurun): one round of "is it still right, is it faster" for a scan / finalise change.
#   gpurun -- 'bash tools/gpu_check.sh <tag> [quick|modes|full]'  -> gpurun_out/<tag>/*
#   quick: parity tests in the default scan mode, scan-only timing, bench line
#   modes: the parity tests in all three TATAJUBA_AMD_FAST modes as well
#   full : modes + the instruction counters of the scan kernels (tools/exp_valu.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-chk}; MODE=${2:-quick}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?
tail -2 $O/tests.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED (default mode)"; exit 1; fi
TJ_REPS=8 timeout -k 10 120 python tools/exp_scan_only.py > $O/scan_only.log 2>&1 && tail -1 $O/scan_only.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-io-stages --no-cpu-baseline > $O/bench.json 2> $O/bench.err && python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("bench: ms/step %.4f scan %.4f (frac %.4f) finalise %.4f" % (d["ms_per_step"], d["stages"]["scan"]["ms"], d["roofline"]["frac"], d["stages"]["finalise"]["ms"]))
PY
if [ "$MODE" != quick ]; then
  for f in 0 2; do
    TATAJUBA_AMD_FAST=$f timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/tests_fast$f.log 2>&1; rc=$?
    echo "FAST=$f: $(tail -1 $O/tests_fast$f.log)"
    if [ $rc -ne 0 ]; then echo "TESTS FAILED (FAST=$f)"; exit 1; fi
  done
fi
if [ "$MODE" != quick ]; then      # the in-kernel partition of rounds 1-3 (k <= 12), still the path of k > 12
  TATAJUBA_AMD_SINK=fused timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/tests_fused.log 2>&1; rc=$?
  echo "SINK=fused: $(tail -1 $O/tests_fused.log)"
  if [ $rc -ne 0 ]; then echo "TESTS FAILED (SINK=fused)"; exit 1; fi
fi
if [ "$MODE" != quick ]; then      # two-word records (k 13 ... 28) through the record log as well (default: k <= 12 only)
  TATAJUBA_AMD_SINK=log timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/tests_log2.log 2>&1; rc=$?
  echo "SINK=log: $(tail -1 $O/tests_log2.log)"
  if [ $rc -ne 0 ]; then echo "TESTS FAILED (SINK=log)"; exit 1; fi
fi
if [ "$MODE" = full ]; then
  bash tools/exp_valu.sh > $O/valu.log 2>&1; tail -4 $O/valu.log
fi
