#!/bin/bash
# SQ counters of aggregate1_kernel per launch on the headline workload (rocprofv3 --pmc, one pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/agg1pmc
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $O -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-io-stages --no-cpu-baseline > /dev/null 2> $O/err.txt
python3 - <<PY
import csv, collections, glob
f = glob.glob("$O/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    kn = r["Kernel_Name"].split("(")[0]
    if "aggregate" in kn or "bin_sort" in kn or "partition_log" in kn or "scan_fast" in kn: agg[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn, a in agg.items():
    print(kn[:30], {k: round(sum(v) / len(v) / 1e6, 2) for k, v in a.items()})
PY
