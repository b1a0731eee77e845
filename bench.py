#!/usr/bin/env python3
"""bench.py -- whole-job reads/s of the homopolymer-tract hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one sample's reads: scan kernel (tract detection + flank packing) followed
by the device finalise (sort / reduce / filter / index / coverage), with the read stream already resident in HBM.
Workload at N=1: BASELINE.json configs[1] -- 1 sample, 10 M synthetic 150 bp single-end reads from a 5 Mb genome,
k=10, min_tract=3, strand-bias filter on, min_coverage=5.  With N>1 GPUs every rank holds its own sample of that size
(weak scaling, samples are independent: reference src/genome_set.c:66-94) and the step includes the exchange the
path has: an all-gatherv (RCCL) of the per-sample histograms and their merge, run on a second stream under the next
sample's scan (the last one inside the timed region).

Prints ONE JSON line on rank 0.  The CPU oracle is used only for the cpu_baseline leg (never in the timed GPU path).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per sample (per GPU)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--read-len-max", type=int, default=0, help="> read-len: ragged reads, length uniform in [read-len, read-len-max]")
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--kmer", type=int, default=10)
    ap.add_argument("--min-tract", type=int, default=3)
    ap.add_argument("--min-coverage", type=int, default=5)
    ap.add_argument("--cpu-reads", type=int, default=2_000_000, help="reads of the same sample timed on the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import tatajuba_amd as tj

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available() or tj.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("TATAJUBA_BENCH_BACKEND", "nccl")    # "gloo" only to rehearse N > 1 on a one-GPU box
    if local >= ndev:
        if backend == "nccl":
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPU(s) visible")
        local = local % ndev
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    k, m, L = args.kmer, args.min_tract, args.read_len
    # one sample per rank: same genome, per-sample tract-length variants and read seeds (SURVEY 8d config 4 recipe)
    host = tj.synth_stream(args.reads, L, args.genome, seed_reads=0x7A7A1000 + rank, read_len_max=args.read_len_max,
                           variant_seed=(rank if world > 1 else 0), n_threads=max(1, 16 // max(1, min(world, 8))))
    n_bytes = host.size
    dev = torch.from_numpy(host).cuda()                 # resident in HBM before the timed region
    stream = torch.cuda.current_stream()
    # N > 1: two counters take turns, so that a sample's histogram stays in place while the next sample is scanned; its
    # exchange (all-gatherv + merge, on a stream and a counter of their own) runs under that scan.
    ctrs = [tj.Counter(k, device=local) for _ in range(2 if world > 1 else 1)]
    for cc in ctrs:
        cc.set_stream(stream.cuda_stream)
    c = ctrs[0]
    side = merger = None
    if world > 1:
        side = torch.cuda.Stream()
        merger = tj.Counter(k, device=local)
        merger.set_stream(side.cuda_stream)

    gathered = None
    pending = None                                        # counter whose histogram has not been exchanged yet
    n_step = 0

    def exchange(cnt):
        nonlocal gathered
        from tatajuba_amd.dist import all_gather_histograms, merge_histograms_device
        torch.cuda.set_device(local)                      # (device and stream are per thread)
        with torch.cuda.stream(side):
            rec, cnts = all_gather_histograms(cnt, dist)    # RCCL all-gatherv of the per-sample histograms
            gathered = merge_histograms_device(merger, rec, cnts)   # every rank holds the union (reference: genome_set.c:250-289)

    # The exchange has host synchronisations of its own (sizes, union size): it runs in a helper thread so that the main
    # thread can go on to the finalise of the sample being scanned.  Only that thread issues collectives while the loop
    # runs, one exchange at a time, in the same order on every rank.
    import threading
    worker = None
    worker_err = []

    def exchange_async(cnt):
        nonlocal worker
        def run():
            try:
                exchange(cnt)
            except BaseException as e:                    # noqa: BLE001
                worker_err.append(e)
        worker = threading.Thread(target=run)
        worker.start()

    def exchange_join():
        nonlocal worker
        if worker is not None:
            worker.join()
            worker = None
        if worker_err:
            raise worker_err[0]

    def step():
        nonlocal pending, n_step, c
        c = ctrs[n_step % len(ctrs)]
        n_step += 1
        exchange_join()                                   # (the exchange before last: long done)
        c.reset()
        c.scan_device(dev.data_ptr(), n_bytes, m)       # asynchronous: the previous sample's exchange runs under it
        if pending is not None:
            exchange_async(pending)
        st = c.finalise(1, args.min_coverage)
        if st != 0:
            raise SystemExit(f"finalise status {st}")
        if world > 1:
            pending = c
        return c.last_scan_ms(), c.last_finalise_ms()

    def drain():                                          # the last sample's exchange
        nonlocal pending
        exchange_join()
        if pending is not None:
            exchange(pending)
            pending = None

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # raw record count of the sample (the aggregation consumes the raw records, so count them once, untimed)
    c.reset()
    c.scan_device(dev.data_ptr(), n_bytes, m)
    raw = c.raw_count()
    for _ in range(args.warmup):
        step()
    drain()
    fence()
    t0 = time.perf_counter()
    scan_ms, fin_ms = [], []
    for _ in range(args.steps):
        a, b = step()
        scan_ms.append(a); fin_ms.append(b)
    drain()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    kept = c.n_kept
    scan_avg = float(np.mean(scan_ms))
    fin_avg = float(np.mean(fin_ms))
    total_reads = args.reads * world * args.steps
    value = total_reads / dt

    # roofline of the scan kernel: algorithmic bytes = the stream itself, (L + 1) bytes per read in this layout
    # (SURVEY 8d quotes L + 8 for an offsets-table layout; the sentinel layout carries boundaries in-band -- DESIGN.md)
    scan_bytes = float(n_bytes)
    scan_gbs = scan_bytes / (scan_avg * 1e-3) / 1e9
    # sort+reduce stage, one-pass bound: read every raw record once (8*W bytes in this layout), write 24 B
    # per kept record
    wbytes = 8.0 * (1 if k <= 12 else (2 if k <= 28 else 4))
    fin_bytes = wbytes * raw + 24.0 * kept
    fin_gbs = fin_bytes / (fin_avg * 1e-3) / 1e9
    dominant = "scan" if scan_avg >= fin_avg else "finalise"
    # HBM bytes per launch from the PMC passes of the same command (tools/pmc_traffic.py -> profiles/; rocprofv3 cannot
    # run inside the timed process), only when they were taken on this workload
    traffic, traffic_rw = {}, {}
    try:
        tj_prof = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
        if args.reads == 10_000_000 and L == 150 and k == 10 and m == 3:
            traffic = {kk: vv["hbm_bytes"] for kk, vv in tj_prof.items() if isinstance(vv, dict)}
            traffic_rw = {kk: (vv["hbm_read_bytes"], vv["hbm_write_bytes"]) for kk, vv in tj_prof.items() if isinstance(vv, dict)}
    except (OSError, ValueError, KeyError):
        pass
    roof = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "traffic": None}
    if dominant == "scan":
        roof.update({"kernel": "scan_bins_kernel<1>" if k <= 12 else ("scan_bins_kernel<2>" if k <= 28 else "scan_bins_kernel<4>"),
                     "achieved": scan_gbs, "frac": scan_gbs / HBM_PEAK_GBS, "ms": scan_avg, "algorithmic_bytes": scan_bytes,
                     "traffic": traffic.get("scan_bins_kernel<1>") if k <= 12 else None})
        if k <= 12 and "scan_bins_kernel<1>" in traffic_rw:   # reads = the stream (no re-reads); writes = the raw records leaving the kernel
            roof["traffic_read"], roof["traffic_write"] = traffic_rw["scan_bins_kernel<1>"]
    else:
        roof.update({"kernel": "finalise (aggregate_kernel dominates)", "achieved": fin_gbs, "frac": fin_gbs / HBM_PEAK_GBS,
                     "ms": fin_avg, "algorithmic_bytes": fin_bytes})

    out = {
        "metric": "reads/s (whole node), 150 bp synthetic FASTQ, homopolymer scan + context histogram",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8/u64", "data": "synthetic",
        "config": {"workload": f"{world} sample(s) x {args.reads} synthetic {L} bp single-end reads, genome {args.genome} bp, "
                               f"k={k} min_tract={m} remove_biased=1 min_coverage={args.min_coverage} (BASELINE.json configs[1] per GPU)",
                   "reads_per_gpu": args.reads, "raw_records_per_gpu": int(raw), "kept_records": int(kept),
                   "parallelism": f"sample-per-gpu x{world}" + (", histogram exchange (all-gatherv + merge) overlapped with the next sample's scan" if world > 1 else "")},
        "roofline": roof,
        "stages": {"scan": {"ms": scan_avg, "algorithmic_GBps": scan_gbs, "frac_of_hbm_peak": scan_gbs / HBM_PEAK_GBS,
                            "reads_per_s": args.reads / (scan_avg * 1e-3)},
                   "finalise": {"ms": fin_avg, "algorithmic_GBps": fin_gbs, "frac_of_hbm_peak": fin_gbs / HBM_PEAK_GBS}},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import orc                             # checker / baseline only
        n_cpu = min(args.cpu_reads, args.reads)
        sample = host[: n_cpu * (L + 1)] if args.read_len_max <= L else host
        o = orc.Oracle(k)
        t1 = time.perf_counter()
        o.scan_stream(sample, m)
        t2 = time.perf_counter()
        o.finalise(1, args.min_coverage)
        t3 = time.perf_counter()
        out["cpu_baseline"] = {"value": n_cpu / (t3 - t1), "unit": "reads/s", "cores": 1, "kind": "port",
                               "sample": f"first {n_cpu} reads of the same sample; scan {t2 - t1:.2f} s + sort/dedupe/filter {t3 - t2:.2f} s "
                                         f"on 1 core (the reference runs one thread per sample: src/genome_set.c:66-68); "
                                         f"parse/inflate excluded on both sides",
                               "host_cores_available": os.cpu_count()}
        # the reference's only parallel loop is over samples (OpenMP, src/genome_set.c:66-94): the same CPU path on P
        # samples at once (P threads, one slice of the stream each) is what a whole host delivers
        import threading
        P = max(1, min(16, os.cpu_count() or 1))
        if P > 1 and args.read_len_max <= L:
            per = min(n_cpu // 2, args.reads // P)
            slices = [host[i * per * (L + 1): (i + 1) * per * (L + 1)] for i in range(P)]

            def one(sl):
                oo = orc.Oracle(k)
                oo.scan_stream(sl, m)
                oo.finalise(1, args.min_coverage)
                oo.close()
            ths = [threading.Thread(target=one, args=(sl,)) for sl in slices]
            t4 = time.perf_counter()
            [t.start() for t in ths]
            [t.join() for t in ths]
            t5 = time.perf_counter()
            out["cpu_baseline"]["multi_sample"] = {"value": P * per / (t5 - t4), "unit": "reads/s", "cores": P,
                                                   "sample": f"{P} samples of {per} reads at once, one thread each"}
    if rank == 0:
        print(json.dumps(out))
    for cc in ctrs:
        cc.close()
    if merger is not None:
        merger.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
