cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "c3s:--reads 20000000 --genome 50000000 --kmer 15 --min-tract 4" "c5:--config 5"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/aggpmc_$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-io-stages --no-cpu-baseline $args > /dev/null 2>&1
done
python3 - <<PY
import csv, collections, glob
for d in sorted(glob.glob("$R/gpurun_out/aggpmc_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        kn = r["Kernel_Name"].split("(")[0]
        if "aggregate" in kn: agg[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, a in agg.items():
        print(d.split("/")[-1], kn[:30], {k: round(sum(v) / len(v) / 1e6, 2) for k, v in a.items()})
PY
