#!/bin/bash
# per-phase instruction counts of the scan kernel: ablated builds (tools/build_exp.sh stopN -DTJ_EXP_STOP_AFTER=N) under --pmc
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/phase
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in libtatajuba_amd_stop1.so libtatajuba_amd_stop2.so libtatajuba_amd_stop3.so libtatajuba_amd.so; do
  TJ_DIAG_LIB=$lib timeout -k 5 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $O/$lib -o p -- python3 $R/tools/exp_scan_only.py > $O/$lib.out 2> $O/$lib.err
  echo $lib >> $O/progress.txt
done
python3 - <<PY
import csv, collections, glob
for d in sorted(glob.glob("$O/*.so")):
    f = glob.glob(d + "/*counter_collection.csv")
    if not f: print(d, "no csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "scan_bins" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(d.split("/")[-1], {k: round(sum(v)/len(v)/1e6, 1) for k, v in agg.items()})
PY
