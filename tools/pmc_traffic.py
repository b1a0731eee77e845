"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (csv) into per-launch HBM traffic of the hot kernels.

Usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
Corrections (MI355X_MICROARCH.md, section HBM): the counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide coalesced streaming read (16 B/lane: the scan kernels' global_load_lds_dwordx4 stream), so the
scan kernels' fetch figures are doubled; WRITE_SIZE is exact for streaming stores.  Access widths other than 16 B/lane
are uncalibrated: the aggregate kernel's figures are reported raw (x1) and flagged."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_note": __doc__.split("\n\n")[1].replace("\n", " ")}
for k in sorted(set(fetch) | set(write)):
    short = k.split("(")[0].replace("void ", "")
    if not any(s in short for s in ("scan_fast", "scan_bins", "aggregate", "radix_scatter", "partition_log")):
        continue
    f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
    # (partition_log_kernel reads its records 8 B per lane, coalesced: FETCH_SIZE reports half of those bytes as well -- its 0.25 M KiB
    # for a log of 0.48 GB -- so the same correction applies; its writes, in runs of 32 records, are counted as written)
    corr = 2.0 if ("scan_bins" in short or "scan_fast" in short or "partition_log" in short) else 1.0
    out[short] = {"launches_averaged": nf.get(k, 0), "FETCH_SIZE_KiB_raw": f_kib, "WRITE_SIZE_KiB_raw": w_kib,
                  "fetch_correction": corr, "hbm_read_bytes": f_kib * 1024 * corr, "hbm_write_bytes": w_kib * 1024,
                  "hbm_bytes": f_kib * 1024 * corr + w_kib * 1024, "calibrated": corr == 2.0}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
