#!/usr/bin/env python3
"""bench.py -- whole-job reads/s of the homopolymer-tract hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one sample's reads: scan kernel (tract detection + flank packing) followed
by the device finalise (sort / reduce / filter / index / coverage), with the read stream already resident in HBM.
Workload at N=1: BASELINE.json configs[1] -- 1 sample, 10 M synthetic 150 bp single-end reads from a 5 Mb genome,
k=10, min_tract=3, strand-bias filter on, min_coverage=5.  With N>1 GPUs every rank holds its own sample of that size
(weak scaling, samples are independent: reference src/genome_set.c:66-94) and the step includes the exchange the
path has: an all-gatherv (RCCL) of the per-sample histograms and their merge, run on a second stream under the next
sample's scan (the last one inside the timed region).

`--gpus N` without a launcher (the driver's command shape): the parent starts N fresh child processes, one rank per GPU,
before anything in it has touched a GPU, relays rank 0's JSON line and exits with the children's return code.
`--config {2,3,4,5}` picks one of BASELINE.json's configurations (numbered as in SURVEY.md 8(d): 2 = configs[1], the
headline, and the default); `config.workload` names what was really run.

Prints ONE JSON line on rank 0.  The CPU oracle is used only for the cpu_baseline leg (never in the timed GPU path).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)


# BASELINE.json configurations, numbered as in SURVEY.md 8(d) (config N = BASELINE.json configs[N - 1]); inputs per
# SURVEY 8(d)'s recipe.  Config 5 (5 M reads of 2-20 kb = 55 GB per sample) runs at 1/10 scale per sample, as 8(d) allows.
CONFIGS = {
    2: dict(reads=10_000_000, read_len=150, read_len_max=0, genome=5_000_000, kmer=10, min_tract=3,
            label="BASELINE.json configs[1]: 1 sample, 10 M x 150 bp single-end, k=10 min_tract=3"),
    3: dict(reads=100_000_000, read_len=150, read_len_max=0, genome=50_000_000, kmer=15, min_tract=4,
            label="BASELINE.json configs[2]: 1 sample, 50 M pairs = 100 M x 150 bp reads, k=15 min_tract=4, strand-bias filter"),
    4: dict(reads=40_000_000, read_len=150, read_len_max=0, genome=20_000_000, kmer=15, min_tract=4,
            label="BASELINE.json configs[3]: one sample of 20 M pairs = 40 M x 150 bp reads per GPU, k=15 min_tract=4, all-gatherv merge"),
    5: dict(reads=500_000, read_len=2000, read_len_max=20000, genome=100_000_000, kmer=25, min_tract=4,
            label="BASELINE.json configs[4] at 1/10 scale per sample: 0.5 M long reads of 2-20 kb per GPU, k=25 min_tract=4"),
}


def self_launch(args):
    """--gpus N > 1 without a launcher: one child process per rank (fresh processes: nothing here has touched a GPU)."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    import tempfile
    import time as _time
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    # a rank that dies must not leave the others waiting in a rendezvous or a collective: the first non-zero exit ends
    # the job (exactly the processes started here, by their handles)
    rcs = [None] * len(procs)
    failed = False
    while any(rc is None for rc in rcs) and not failed:
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = pr.poll()
                if rcs[i] not in (None, 0):
                    failed = True
        if not failed:
            _time.sleep(0.05)
    if failed:
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                pr.terminate()
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                try:
                    rcs[i] = pr.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    pr.kill()
                    rcs[i] = pr.wait()
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs) if not failed else max(1, max(abs(rc) for rc in rcs if rc not in (None,)))


def rehearse(args):
    """Launch plumbing only (no GPU work, no throughput): rendezvous, barrier, max over ranks, one JSON line from rank 0.
    What the CPU tests run; a product run never takes this path."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("TATAJUBA_BENCH_REHEARSE_FAIL_RANK") == str(rank):     # (tests: a rank that dies before the rendezvous)
        raise SystemExit(3)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    mine = {"allgather_ms": [], "merge_ms": [], "bytes": [], "collectives": [], "scan_ms": None, "rccl_ranks": None}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)             # (the same plumbing the real run reports its exchange with)
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "max_over_ranks": float(t.item()), "value": None,
                          "config": {"workload": workload_label(args), "rccl_ranks": None},
                          "stages": {"exchange": exchange_report(per_rank)}}))
    if world > 1:
        dist.destroy_process_group()


def exchange_report(per_rank):
    """What a multi-GPU line says about the histogram exchange (all-gatherv of the kept records + merge), from what
    every rank measured: per_rank[r] = {"allgather_ms": [...], "merge_ms": [...], "bytes": [...], "collectives": [...],
    "scan_ms": mean scan ms or None, "rccl_ranks": ncclCommCount or None}.  Times are device times (HIP events inside the
    library, on the exchange's stream); "max_over_ranks" is the mean of the slowest rank."""
    def stat(key):
        means = [float(np.mean(p[key])) for p in per_rank if p.get(key)]
        return {"mean": float(np.mean(means)), "max_over_ranks": float(np.max(means))} if means else None
    n_ex = [len(p.get("allgather_ms") or []) for p in per_rank]
    colls = [x for p in per_rank for x in (p.get("collectives") or [])]
    nbytes = [x for p in per_rank for x in (p.get("bytes") or [])]
    scans = [p["scan_ms"] for p in per_rank if p.get("scan_ms") is not None]
    ranks = sorted({p["rccl_ranks"] for p in per_rank if p.get("rccl_ranks") is not None})
    return {"measured": bool(colls), "exchanges_per_rank": int(min(n_ex)) if n_ex else 0,
            "allgather_ms": stat("allgather_ms"), "merge_ms": stat("merge_ms"),
            "bytes_gathered_per_rank_and_exchange": float(np.mean(nbytes)) if nbytes else None,
            "collectives_per_exchange": float(np.mean(colls)) if colls else None,
            "rccl_ranks_reported": ranks if ranks else None,
            "scan_ms_per_rank": {"min": float(np.min(scans)), "max": float(np.max(scans))} if scans else None,
            "note": "tjamd_allgather_histograms (one ncclAllGather of max-padded blocks per settled exchange, pack + unpack kernels) and tjamd_merge_samples "
                    "on the side stream, under the next sample's scan; device times from HIP events inside the library; never run on more than one GPU "
                    "before the driver's 8-GPU node (the pool's boxes have one)"}


def workload_label(args):
    """names what the arguments really are: a BASELINE.json configuration, or 'custom'"""
    for n, c in CONFIGS.items():
        if all(getattr(args, kk) == c[kk] for kk in ("reads", "read_len", "read_len_max", "genome", "kmer", "min_tract")):
            return c["label"] + f" (SURVEY 8d config {n})"
    return "custom workload (no BASELINE.json configuration)"


def io_stages(tj, host, args, k, m, L):
    """The two throughputs around the HBM-resident one, on a bounded part of the same sample, never `value`:
    a pinned host-resident stream through tjamd_scan_host (PCIe-inclusive) + finalise, and a plain FASTQ file through
    new_or_append_hopo_counter_from_file + finalise_hopo_counter (parse + copy + scan + finalise)."""
    import ctypes as C
    import tempfile
    from tatajuba_amd import capi
    res = {}
    n_io = min(args.io_reads, args.reads)
    if args.read_len_max > L:                               # ragged reads: cut after the n_io-th delimiter
        ends = np.flatnonzero(host[: min(host.size, n_io * (args.read_len_max + 1))] == 10)
        n_io = min(n_io, ends.size)
        part = host[: ends[n_io - 1] + 1]
    else:
        part = host[: n_io * (L + 1)]
    lib = capi.lib()
    pin = lib.tjamd_host_alloc(part.size)
    C.memmove(pin, part.ctypes.data, part.size)
    c = tj.Counter(k)
    best = 1e9
    for rep in range(3):
        c.reset()
        t = time.perf_counter()
        if lib.tjamd_scan_host(c._h, C.c_void_p(pin), part.size, m):
            raise SystemExit("tjamd_scan_host failed")
        c.finalise(1, args.min_coverage)
        best = min(best, time.perf_counter() - t)
    c.close()
    lib.tjamd_host_free(C.c_void_p(pin))
    res["from_host"] = {"reads_per_s": n_io / best, "GBps": part.size / best / 1e9, "seconds": best, "reads": n_io,
                        "note": "pinned host stream -> tjamd_scan_host (host-to-device copy included) + finalise"}
    tmp = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
    fq = os.path.join(tmp, "bench.fq")
    reads = bytes(part).split(b"\n")[:-1]
    with open(fq, "wb") as f:
        for i in range(0, len(reads), 100000):
            f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (j, reads[j], b"I" * len(reads[j])) for j in range(i, min(len(reads), i + 100000))))
    opt = tj.Options.defaults(k, m, args.min_coverage, True)
    best = 1e9
    for rep in range(2):
        t = time.perf_counter()
        h = tj.HopoCounter.new_or_append_from_file(None, fq, opt)
        h.finalise()
        best = min(best, time.perf_counter() - t)
        h.delete()
    res["from_file"] = {"reads_per_s": len(reads) / best, "file_MB": os.path.getsize(fq) / 1e6, "seconds": best, "reads": len(reads),
                        "note": "plain FASTQ -> new_or_append_hopo_counter_from_file (multi-threaded feeder) + finalise_hopo_counter"}
    # the same records gzip-compressed, the way read files usually arrive: one gzip member (inflate on one thread, running
    # ahead of the parse) and BGZF (bgzip: 64 KiB members inflated side by side); qualities drawn from 8 levels so that the
    # compression ratio is that of real files rather than of a constant string
    import struct
    import zlib
    n_gz = min(len(reads), args.gz_reads)
    rng = np.random.default_rng(5)
    levels = np.frombuffer(b"#-27<AFI", np.uint8)
    chunks = []
    for i in range(0, n_gz, 100000):
        js = range(i, min(n_gz, i + 100000))
        q = levels[rng.integers(0, 8, size=sum(len(reads[j]) for j in js))].tobytes()
        off, recs = 0, []
        for j in js:
            recs.append(b"@r%d\n%s\n+\n%s\n" % (j, reads[j], q[off:off + len(reads[j])]))
            off += len(reads[j])
        chunks.append(b"".join(recs))
    txt = b"".join(chunks)

    def member(data, bgzf):
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = co.compress(data) + co.flush()
        tail = struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)
        if not bgzf:
            return b"\x1f\x8b\x08\x00" + b"\0" * 4 + b"\x00\xff" + body + tail
        return b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) + body + tail

    for key, blob, how in (("from_gzip", member(txt, False), "one gzip member: entered at block starts found by trial and inflated on all feeder threads (each stretch checked by the decoder in front of it), one window ahead of the parse"),
                           ("from_bgzf", b"".join(member(txt[i:i + 0xff00], True) for i in range(0, len(txt), 0xff00)) + member(b"", True),
                            "BGZF: members inflated side by side by the feeder threads")):
        gz = os.path.join(tmp, key + ".fq.gz")
        with open(gz, "wb") as f:
            f.write(blob)
        best = 1e9
        for rep in range(2):
            t = time.perf_counter()
            h = tj.HopoCounter.new_or_append_from_file(None, gz, opt)
            h.finalise()
            best = min(best, time.perf_counter() - t)
            h.delete()
        res[key] = {"reads_per_s": n_gz / best, "file_MB": len(blob) / 1e6, "inflated_MB": len(txt) / 1e6, "seconds": best, "reads": n_gz,
                    "note": how + " -> new_or_append_hopo_counter_from_file + finalise_hopo_counter"}
        os.remove(gz)
    os.remove(fq)
    os.rmdir(tmp)
    return res


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
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=None,
                    help="a BASELINE.json configuration (SURVEY 8d numbering; 2 = configs[1] = the headline = the default sizes)")
    ap.add_argument("--order-stream", action="store_true", help="the ordering step of a sample's finalise on a stream of its own (tjamd_counter_set_order_stream) instead of the scan's; measured: 0.832 against 0.824 ms per step, so not the default")
    ap.add_argument("--no-io-stages", action="store_true", help="skip the stages measured after the timed region other than the drop-in's (stages.fused_sink, from_host, from_file, from_gzip, from_bgzf)")
    ap.add_argument("--io-reads", type=int, default=2_000_000, help="reads of the sample used for stages.from_host / from_file")
    ap.add_argument("--gz-reads", type=int, default=500_000, help="reads of the sample used for stages.from_gzip / from_bgzf")
    ap.add_argument("--rehearse", action="store_true", help="launch plumbing only, no GPU work (CPU tests)")
    args = ap.parse_args()
    if args.config is not None:
        for kk, vv in CONFIGS[args.config].items():
            if kk != "label":
                setattr(args, kk, vv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # (under a profiler the preloaded library has already initialised the GPU in this process: starting the ranks from
        # here would be a launcher hop the pool forbids -- profile one rank with RANK / WORLD_SIZE / LOCAL_RANK set instead)
        if any(v in os.environ for v in ("ROCPROFILER_REGISTER_FORCE_LOAD", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH")) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
            raise SystemExit("bench.py --gpus N > 1 does not start its own ranks under rocprofv3: profile one rank (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT set)")
        raise SystemExit(self_launch(args))
    if args.rehearse:
        return rehearse(args)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (the host driver supports dmabuf IPC only: RCCL between processes needs it)
    import torch
    import tatajuba_amd as tj

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
        # The job's control plane (rendezvous, barrier, the max over ranks of one number) goes over gloo: the data path -- the
        # all-gatherv of the histograms -- is the C library's own RCCL communicator (tjamd_comm_*), and a second NCCL
        # communicator inside torch for two barriers would only take device memory and channels away from it.
        dist.init_process_group("gloo", rank=rank, world_size=world)

    k, m, L = args.kmer, args.min_tract, args.read_len
    # one sample per rank: same genome, per-sample tract-length variants and read seeds (SURVEY 8d config 4 recipe)
    host = tj.synth_stream(args.reads, L, args.genome, seed_reads=0x7A7A1000 + rank, read_len_max=args.read_len_max,
                           variant_seed=(rank if world > 1 else 0), n_threads=max(1, 16 // max(1, min(world, 8))))
    n_bytes = host.size
    dev = torch.from_numpy(host).cuda()                 # resident in HBM before the timed region
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()                          # one stream for both counters (the default stream's handle is null, which
                                                          # tjamd_counter_set_stream reads as "the counter's own"): samples run one after the other
    # Two counters take turns on one stream: while the host waits for a sample's counts (tjamd_finalise_end) and, with
    # N > 1, exchanges its histogram (all-gatherv + merge, on a stream and a counter of their own), the next sample's scan
    # is already queued -- a step is still one whole sample, scanned and finalised, and K samples begin and end inside the
    # timed region.
    ctrs = [tj.Counter(k, device=local) for _ in range(2)]
    # (--order-stream) a second stream for the ordering step of a sample's finalise (six small launches, latency-shaped): it
    # runs behind an event, under the next sample's scan, which the first stream has queued on the other counter.  Measured
    # in round 3: no gain (the scan fills every CU, the small kernels wait for its workgroups to retire and delay it by as
    # much as they save), so the default stays one stream.
    order = torch.cuda.Stream() if args.order_stream else None
    for cc in ctrs:
        cc.set_stream(stream.cuda_stream)
        if order is not None:
            cc.set_order_stream(order.cuda_stream)
    c = ctrs[0]
    side = merger = None
    if world > 1:
        side = torch.cuda.Stream()
        merger = tj.Counter(k, device=local)
        merger.set_stream(side.cuda_stream)

    gathered = None
    pending = None                                        # counter whose histogram has not been exchanged yet
    n_step = 0

    # The exchange is the C library's (tjamd_allgather_histograms: ncclAllGather over RCCL, on the side stream) whenever the
    # ranks have a GPU each; the gloo rehearsal on a one-GPU box keeps the torch.distributed plumbing of tatajuba_amd/dist.py.
    comm = None
    ex_stats = {"allgather_ms": [], "merge_ms": [], "bytes": [], "collectives": []}     # per exchange of this rank
    union_buf = {"keys": None, "mat": None, "cap": 0}
    if world > 1 and backend == "nccl":
        ident = [tj.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)          # (the id travels over the job's own rendezvous)
        comm = tj.Comm(merger, ident[0], rank, world)
        comm.set_stream(side.cuda_stream)

    def exchange(cnt):
        nonlocal gathered
        torch.cuda.set_device(local)                      # (device and stream are per thread)
        if comm is not None:
            ptr, cnts, tot = comm.allgather(cnt)          # all-gatherv of the per-sample histograms: one ncclAllGather per exchange
            if tot > union_buf["cap"]:                    # (output buffers of the merge: grown, never per step)
                union_buf["cap"] = int(tot * 1.5) + 1024
                with torch.cuda.stream(side):
                    union_buf["keys"] = torch.empty(union_buf["cap"] * 24, dtype=torch.uint8, device="cuda")
                    union_buf["mat"] = torch.empty((union_buf["cap"], world), dtype=torch.int32, device="cuda")
            nu = tj.lib().tjamd_merge_samples(merger._h, ptr, cnts, world, C.c_void_p(union_buf["keys"].data_ptr()),
                                              C.c_void_p(union_buf["mat"].data_ptr()), tot)   # every rank holds the union (reference: genome_set.c:250-289)
            if nu < 0:
                raise RuntimeError(tj.lib().tjamd_last_error().decode())
            gathered = (union_buf["keys"][: nu * 24], union_buf["mat"][:nu])
            le = comm.last_exchange()
            if le is not None:
                ex_stats["allgather_ms"].append(le[0]); ex_stats["bytes"].append(le[1]); ex_stats["collectives"].append(le[2])
            ex_stats["merge_ms"].append(merger.last_merge_ms())
            return
        from tatajuba_amd.dist import all_gather_histograms, merge_histograms_device
        with torch.cuda.stream(side):
            rec, cnts = all_gather_histograms(cnt, dist)
            gathered = merge_histograms_device(merger, rec, cnts)

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

    begun = None                                          # counter whose finalise has been queued but not looked at
    times = []                                            # (scan ms, finalise ms) of every sample ended

    def end_sample(cnt):
        st = cnt.finalise_end()
        if st != 0:
            raise SystemExit(f"finalise status {st}")
        times.append((cnt.last_scan_ms(), cnt.last_finalise_ms(), cnt.last_partition_ms()))

    def step():
        nonlocal pending, begun, n_step, c
        c = ctrs[n_step % len(ctrs)]
        n_step += 1
        exchange_join()                                   # (the exchange before last: long done)
        c.reset()
        c.scan_device(dev.data_ptr(), n_bytes, m)       # asynchronous
        c.finalise_begin(1, args.min_coverage)            # likewise: queued behind the scan
        if begun is not None:                             # the sample before: its counts, then (N > 1) its exchange, under this scan
            end_sample(begun)
            if world > 1:
                exchange_async(begun)
        begun = c

    def drain():                                          # the last sample's counts and exchange
        nonlocal pending, begun
        exchange_join()
        if begun is not None:
            end_sample(begun)
            if world > 1:
                exchange(begun)
            begun = None

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
    times.clear()
    for v in ex_stats.values():
        v.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    fence()
    dt = time.perf_counter() - t0
    assert len(times) == args.steps
    scan_ms, fin_ms, part_ms = [t[0] for t in times], [t[1] for t in times], [t[2] for t in times]
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    kept = c.n_kept
    scan_avg = float(np.mean(scan_ms))
    fin_avg = float(np.mean(fin_ms))
    part_avg = float(np.mean(part_ms))
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
    uses_log = c.uses_log()
    scan_kernel = "scan_fast_kernel<%d, %s>" % (1 if k <= 12 else (2 if k <= 28 else 4), "true" if uses_log else "false")
    traffic, traffic_src = None, None
    import glob
    for prof in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):     # the latest round's passes first
        try:
            tj_prof = json.load(open(prof))
            w = tj_prof.get("_workload", {})
            if (w.get("reads"), w.get("read_len"), w.get("kmer"), w.get("min_tract")) == (args.reads, L, k, m) and args.read_len_max <= L:
                traffic = {kk: vv for kk, vv in tj_prof.items() if isinstance(vv, dict) and "hbm_bytes" in vv}
                traffic_src = ("profiles/" + os.path.basename(prof) + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                               "(tools/pmc_traffic.py), per launch")
                break
        except (OSError, ValueError, KeyError):
            pass
    roof = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "traffic": None, "traffic_from_profiles": traffic_src}
    if dominant == "scan":
        roof.update({"kernel": scan_kernel, "achieved": scan_gbs, "frac": scan_gbs / HBM_PEAK_GBS, "ms": scan_avg, "algorithmic_bytes": scan_bytes})
        if traffic and scan_kernel in traffic:              # reads = the stream (no re-reads); writes = the raw records leaving the kernel
            roof["traffic"] = traffic[scan_kernel]["hbm_bytes"]
            roof["traffic_read"], roof["traffic_write"] = traffic[scan_kernel]["hbm_read_bytes"], traffic[scan_kernel]["hbm_write_bytes"]
    else:
        roof.update({"kernel": "finalise (aggregate_kernel dominates)", "achieved": fin_gbs, "frac": fin_gbs / HBM_PEAK_GBS,
                     "ms": fin_avg, "algorithmic_bytes": fin_bytes})

    out = {
        "metric": "reads/s (whole node), 150 bp synthetic FASTQ, homopolymer scan + context histogram",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8/u64", "data": "synthetic",
        "config": {"workload": f"{world} sample(s) x {args.reads} synthetic {L} bp single-end reads, genome {args.genome} bp, "
                               f"k={k} min_tract={m} remove_biased=1 min_coverage={args.min_coverage}"
                               + (f", read length uniform in [{L}, {args.read_len_max}]" if args.read_len_max > L else "") + " -- " + workload_label(args),
                   "reads_per_gpu": args.reads, "raw_records_per_gpu": int(raw), "kept_records": int(kept),
                   "parallelism": f"sample-per-gpu x{world}" + (", histogram exchange (all-gatherv + merge) overlapped with the next sample's scan" if world > 1 else ""),
                   "pipelining": "two counters take turns on one stream: sample i's counts are fetched (tjamd_finalise_end) after sample i + 1 has been queued; every step is a whole sample, K begin and end inside the timed region"
                                 + ("" if not args.order_stream else "; the ordering step of sample i's finalise (six small launches) runs on a second stream, under sample i + 1's scan")},
        "roofline": roof,
        "stages": {"scan": {"ms": scan_avg, "algorithmic_GBps": scan_gbs, "frac_of_hbm_peak": scan_gbs / HBM_PEAK_GBS,
                            "reads_per_s": args.reads / (scan_avg * 1e-3), "kernel": scan_kernel,
                            "note": "stream resident in HBM (the throughput `value` is quoted on)"},
                   "finalise": {"ms": fin_avg, "algorithmic_GBps": fin_gbs, "frac_of_hbm_peak": fin_gbs / HBM_PEAK_GBS}},
    }
    if world > 1:
        mine = dict(ex_stats, scan_ms=scan_avg, rccl_ranks=(comm.count if comm is not None else None))
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        out["stages"]["exchange"] = exchange_report(per_rank)
        out["config"]["rccl_ranks"] = comm.count if comm is not None else None
        out["config"]["control_plane"] = "torch.distributed over gloo (rendezvous, barrier, max over ranks); data path: the library's RCCL communicator"
    if uses_log:
        # k <= 28: the scan kernel appends its records to a linear log, partition_log_kernel distributes the log over the hash
        # buckets (its algorithmic bytes: every record read once and written once)
        part_bytes = 2.0 * wbytes * raw
        out["stages"]["partition"] = {"ms": part_avg, "kernel": "partition_log_kernel<%d>" % (1 if k <= 12 else 2), "algorithmic_bytes": part_bytes,
                                      "algorithmic_GBps": part_bytes / (part_avg * 1e-3) / 1e9, "frac_of_hbm_peak": part_bytes / (part_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "note": "record log -> hash buckets, between the scan and the finalise; TATAJUBA_AMD_SINK=fused makes the scan kernel partition "
                                              "by itself instead (rounds 1-3: a shorter step, the scan kernel at a third of the HBM peak)"}

    if rank == 0 and world == 1 and uses_log and "TATAJUBA_AMD_SINK" not in os.environ and not args.no_io_stages:
        # The other design, on the same box in the same process, outside the timed region: the scan kernel partitions its
        # records by itself (rounds 1-3) -- a shorter step, a scan kernel at a third of the HBM peak.  Same loop as the timed
        # one (two counters in turn on the stream), a few steps.
        os.environ["TATAJUBA_AMD_SINK"] = "fused"
        try:
            alt = [tj.Counter(k, device=local) for _ in range(2)]
        finally:
            del os.environ["TATAJUBA_AMD_SINK"]
        for cc in alt:
            cc.set_stream(stream.cuda_stream)
        alt_times = []

        def alt_loop(n):
            prev = None
            for i in range(n):
                cc = alt[i % 2]
                cc.reset()
                cc.scan_device(dev.data_ptr(), n_bytes, m)
                cc.finalise_begin(1, args.min_coverage)
                if prev is not None:
                    if prev.finalise_end() != 0:
                        raise SystemExit("finalise failed in the fused-sink stage")
                    alt_times.append((prev.last_scan_ms(), prev.last_finalise_ms()))
                prev = cc
            if prev.finalise_end() != 0:
                raise SystemExit("finalise failed in the fused-sink stage")
            alt_times.append((prev.last_scan_ms(), prev.last_finalise_ms()))
            torch.cuda.synchronize()
        alt_loop(max(2, min(args.warmup, 5)))
        alt_times.clear()
        n_alt = max(2, min(args.steps, 20))
        t1 = time.perf_counter()
        alt_loop(n_alt)
        dta = time.perf_counter() - t1
        if alt[0].uses_log() or alt[0].n_kept != kept:
            raise SystemExit("fused-sink stage: not the fused sink, or another histogram")
        a_scan = float(np.mean([t[0] for t in alt_times]))
        out["stages"]["fused_sink"] = {"reads_per_s": args.reads * n_alt / dta, "ms_per_step": dta / n_alt * 1e3, "steps": n_alt,
                                       "scan_ms": a_scan, "scan_frac_of_hbm_peak": scan_bytes / (a_scan * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "finalise_ms": float(np.mean([t[1] for t in alt_times])), "kernel": scan_kernel.replace("true", "false"),
                                       "note": "TATAJUBA_AMD_SINK=fused, measured after the timed region on the same stream and data: the trade "
                                               "between the two designs (DESIGN.md 3.1c); `value` and `roofline` are the default's"}
        for cc in alt:
            cc.close()
    if rank == 0 and world == 1:
        # what a C caller of the drop-in API gets per sample once its reads are in HBM: the scan, the whole of
        # finalise_hopo_counter's device work AND the copy of the histogram into hc->elem (24-byte records widened to the
        # 40-byte hopo_element on the host) + the index arrays -- the part of finalise_hopo_counter (csrc/hopo_host.c) that a
        # bench step, which leaves the histogram in HBM, does not pay.  One counter, synchronous calls, best of five.
        cd = tj.Counter(k, device=local)
        best = 1e9
        for rep in range(5):
            cd.reset()
            t1 = time.perf_counter()
            cd.scan_device(dev.data_ptr(), n_bytes, m)
            if cd.finalise(1, args.min_coverage) != 0:
                raise SystemExit("finalise failed in the drop-in stage")
            elems = cd.download_kept()
            cd.download_idx()
            best = min(best, time.perf_counter() - t1)
        out["stages"]["dropin"] = {"ms": best * 1e3, "reads_per_s": args.reads / best, "kept_records": int(len(elems)),
                                   "note": "HBM-resident stream -> scan + finalise + histogram copied into 40-byte hopo_elements on the host "
                                           "+ index arrays (what finalise_hopo_counter hands a C caller), synchronous, one counter"}
        cd.close()
    if rank == 0 and world == 1 and not args.no_io_stages:
        out["stages"].update(io_stages(tj, host, args, k, m, L))
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
                                         f"parse/inflate excluded on both sides; the subset flatters the CPU a little: its qsort is "
                                         f"n log n and its depth is {args.reads / max(n_cpu, 1):.0f}x lower than the full sample's",
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
