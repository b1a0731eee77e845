"""Copy the rocprofv3 results of tools/collect_profiles.sh (gpurun_out/<tag>/) into profiles/<tag>_* and print the summary.
usage: install_profiles.py [tag]      (tag defaults to r02)"""
import collections, csv, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
G, P = os.path.join(R, "gpurun_out", TAG), os.path.join(R, "profiles")
HOT = ("scan_fast", "scan_bins", "partition_log", "aggregate", "bin_sort_index")
shutil.copy(os.path.join(G, "kt", "kt_kernel_stats.csv"), os.path.join(P, TAG + "_kernel_stats.csv"))
bench = None
for src, dst in (("bench_under_rocprof.json", TAG + "_bench_under_rocprof.json"), ("bench_plain.json", TAG + "_bench.json")):
    d = json.loads(open(os.path.join(G, src)).read().strip().split("\n")[-1])
    bench = d
    json.dump(d, open(os.path.join(P, dst), "w"), indent=1)
    print(dst, "%.4g reads/s, %.3f ms/step, scan %.3f ms, finalise %.3f ms" % (d["value"], d["ms_per_step"], d["stages"]["scan"]["ms"], d["stages"]["finalise"]["ms"]),
          d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("multi_sample", {}).get("value"))
shutil.copy(os.path.join(G, "fetch", "fetch_counter_collection.csv"), os.path.join(P, TAG + "_pmc_fetch_size.csv"))
shutil.copy(os.path.join(G, "write", "write_counter_collection.csv"), os.path.join(P, TAG + "_pmc_write_size.csv"))
subprocess.check_output([sys.executable, os.path.join(R, "tools", "pmc_traffic.py"), os.path.join(P, TAG + "_pmc_fetch_size.csv"),
                         os.path.join(P, TAG + "_pmc_write_size.csv"), os.path.join(P, TAG + "_hbm_traffic.json")])
t = json.load(open(os.path.join(P, TAG + "_hbm_traffic.json")))
cfg = bench["config"]
import re
mm = re.search(r"x (\d+) synthetic (\d+) bp.*k=(\d+) min_tract=(\d+)", cfg["workload"])
t["_workload"] = {"reads": int(mm.group(1)), "read_len": int(mm.group(2)), "kmer": int(mm.group(3)), "min_tract": int(mm.group(4)), "text": cfg["workload"]}
json.dump(t, open(os.path.join(P, TAG + "_hbm_traffic.json"), "w"), indent=1)
out = {}
for f in ("sq1/sq1_counter_collection.csv", "sq2/sq2_counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(G, f))):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if any(x in k for x in HOT):
            out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
json.dump(out, open(os.path.join(P, TAG + "_pmc_sq_counters.json"), "w"), indent=1)
print({k: {c: round(x / 1e6, 1) for c, x in v.items()} for k, v in out.items()})
print({k: (round(v["hbm_read_bytes"] / 1e6), round(v["hbm_write_bytes"] / 1e6)) for k, v in t.items() if isinstance(v, dict) and "hbm_bytes" in v})
for r in list(csv.DictReader(open(os.path.join(P, TAG + "_kernel_stats.csv"))))[:8]:
    print(r["Name"][:44], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["MinNs"], r["MaxNs"])
