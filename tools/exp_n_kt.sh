#!/bin/bash
# kernel trace of tools/exp_n_reads.py for one fraction of reads with an N (which kernel the extra time goes to); [fraction]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/nkt; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/tools/exp_n_reads.py ${1:-0.005} > $O/out.txt 2> $O/err.txt
tail -1 $O/out.txt
python3 - <<PY
import csv, glob
f = glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print(r["Name"][:48].ljust(48), r["Calls"].rjust(4), "%9.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
