#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 evidence for profiles/.  Timings and counters in separate passes.
#   gpurun -- 'bash tools/collect_profiles.sh [tag] [bench args...]'   -> gpurun_out/<tag>/*      (tag defaults to r02)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}; shift || true
# one rank only: with --gpus N > 1 bench.py becomes a launcher that starts its ranks as child processes, and under
# rocprofv3 (whose preloaded library has initialised the GPU in that parent) such a hop is forbidden on this pool
for a in "$@"; do case "$prev$a" in --gpus[2-9]*|--gpus=[2-9]*|--gpus1[0-9]*) echo "collect_profiles.sh: profile one rank (no --gpus N > 1)"; exit 2;; esac; prev=$a; done
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-io-stages "$@" > $O/bench_under_rocprof.json 2> $O/kt.err
echo "kernel trace done" > $O/progress.txt
if [ -n "$KT_ONLY" ]; then      # kernel trace + plain bench only (the configurations other than the headline)
  python3 $R/bench.py --steps 20 --warmup 5 --no-io-stages "$@" > $O/bench_plain.json 2> $O/bench_plain.err
  exit 0
fi
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-io-stages "$@" > $O/fetch.out 2> $O/fetch.err
echo "fetch done" >> $O/progress.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-io-stages "$@" > $O/write.out 2> $O/write.err
echo "write done" >> $O/progress.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq1 -o sq1 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-io-stages "$@" > $O/sq1.out 2> $O/sq1.err
echo "sq1 done" >> $O/progress.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq2 -o sq2 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-io-stages "$@" > $O/sq2.out 2> $O/sq2.err
echo "sq2 done" >> $O/progress.txt
python3 $R/bench.py --steps 20 --warmup 5 "$@" > $O/bench_plain.json 2> $O/bench_plain.err
find $O -name "*.csv" | head -20
