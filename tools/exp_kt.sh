#!/bin/bash
# Run on the GPU box: kernel-trace stats of a short bench run (per-kernel averages), printed.  [bench args...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/kt; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-io-stages --no-cpu-baseline "$@" > $O/bench.json 2> $O/kt.err
python3 - <<PY
import csv, glob
f = glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:48].ljust(48), r["Calls"].rjust(4), "%9.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
