#!/bin/bash
# SQ counters of the scan kernels per launch (rocprofv3 --pmc): instruction counts are the deterministic metric for
# micro-optimisations, busy / wait cycles show what the waves spend their time on.  LIBS="a.so b.so" compares builds.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/valu
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TJ_REPS=3
for lib in ${LIBS:-libtatajuba_amd.so}; do
  export TJ_DIAG_LIB=$lib       # (inherited by the profiled program: no env/bash hop after --)
  rm -rf $O/$lib.a $O/$lib.b
  timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/$lib.a -o p -- python3 $R/tools/exp_scan_only.py > $O/$lib.a.out 2> $O/$lib.a.err
  timeout -k 5 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/$lib.b -o p -- python3 $R/tools/exp_scan_only.py > $O/$lib.b.out 2> $O/$lib.b.err
  if [ -n "$TRAFFIC" ]; then
    rm -rf $O/$lib.c $O/$lib.d
    timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/$lib.c -o p -- python3 $R/tools/exp_scan_only.py > $O/$lib.c.out 2> $O/$lib.c.err
    timeout -k 5 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/$lib.d -o p -- python3 $R/tools/exp_scan_only.py > $O/$lib.d.out 2> $O/$lib.d.err
  fi
  echo $lib >> $O/progress.txt
done
python3 - <<PY
import csv, collections, glob
for d in sorted(glob.glob("$O/*.so.[abcd]")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        kn = r["Kernel_Name"].split("(")[0]
        if "scan_" in kn or "partition_" in kn: agg[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, a in agg.items():
        print(d.split("/")[-1], kn[:40], {k: round(sum(v) / len(v) / 1e6, 2) for k, v in a.items()})
PY
