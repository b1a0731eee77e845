"""Copy the rocprofv3 results of tools/collect_profiles.sh (gpurun_out/r01/) into profiles/ and print the summary."""
import collections, csv, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out", "r01"), os.path.join(R, "profiles")
shutil.copy(os.path.join(G, "kt", "kt_kernel_stats.csv"), os.path.join(P, "r01_kernel_stats.csv"))
for src, dst in (("bench_under_rocprof.json", "r01_bench_under_rocprof.json"), ("bench_plain.json", "r01_bench.json")):
    d = json.loads(open(os.path.join(G, src)).read().strip().split("\n")[-1])
    json.dump(d, open(os.path.join(P, dst), "w"), indent=1)
    print(dst, "%.4g reads/s, %.3f ms/step, scan %.3f ms, finalise %.3f ms" % (d["value"], d["ms_per_step"], d["stages"]["scan"]["ms"], d["stages"]["finalise"]["ms"]),
          d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("multi_sample", {}).get("value"))
shutil.copy(os.path.join(G, "fetch", "fetch_counter_collection.csv"), os.path.join(P, "r01_pmc_fetch_size.csv"))
shutil.copy(os.path.join(G, "write", "write_counter_collection.csv"), os.path.join(P, "r01_pmc_write_size.csv"))
subprocess.check_output([sys.executable, os.path.join(R, "tools", "pmc_traffic.py"), os.path.join(P, "r01_pmc_fetch_size.csv"),
                         os.path.join(P, "r01_pmc_write_size.csv"), os.path.join(P, "r01_hbm_traffic.json")])
out = {}
for f in ("sq1/sq1_counter_collection.csv", "sq2/sq2_counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(G, f))):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if any(x in k for x in ("scan_bins", "aggregate1", "bin_sort_index")):
            out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
json.dump(out, open(os.path.join(P, "r01_pmc_sq_counters.json"), "w"), indent=1)
print({k: {c: round(x / 1e6, 1) for c, x in v.items()} for k, v in out.items()})
t = json.load(open(os.path.join(P, "r01_hbm_traffic.json")))
print({k: (round(v["hbm_read_bytes"] / 1e6), round(v["hbm_write_bytes"] / 1e6)) for k, v in t.items() if k != "_note"})
for r in list(csv.DictReader(open(os.path.join(P, "r01_kernel_stats.csv"))))[:7]:
    print(r["Name"][:44], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["MinNs"], r["MaxNs"])
