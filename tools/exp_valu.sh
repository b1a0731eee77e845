#!/bin/bash
# deterministic cost metric for scan-kernel micro-optimisations: VALU / SALU wave-instructions per launch (rocprofv3 --pmc)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/valu
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TJ_DIAG_LIB=${TJ_DIAG_LIB:-libtatajuba_amd.so}   # (inherited by the profiled program: no env/bash hop after --)
timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/run -o p -- python3 $R/tools/exp_scan_only.py > $O/run.out 2> $O/run.err
python3 - <<PY
import csv, collections, glob
f = glob.glob("$O/run/*counter_collection.csv")
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "scan_bins" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v)/len(v)/1e6, 2) for k, v in agg.items()})
PY
