#!/bin/bash
# Run on the GPU box: kernel-trace stats of tools/exp_merge.py (cross-sample merge of N samples' histograms), printed.  [N]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ktm; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tools/exp_merge.py ${1:-8} > $O/out.txt 2> $O/kt.err
tail -1 $O/out.txt
python3 - <<PY
import csv, glob
f = glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "merge" in r["Name"] or "bin_" in r["Name"] or "fill" in r["Name"] or "copy" in r["Name"].lower():
        print(r["Name"][:48].ljust(48), r["Calls"].rjust(4), "%9.1f us" % (float(r["AverageNs"]) / 1e3))
PY
