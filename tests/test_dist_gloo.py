"""The N > 1 path on CPU: two gloo ranks, one sample each, all-gatherv of the per-sample histograms (tatajuba_amd/dist.py)
and the context-keyed merge -- on the CPU that is the oracle's restatement of src/genome_set.c:250-289
(oracle/context_oracle.c); the product's merge is a device kernel (tjamd_merge_samples).  The histograms come from the CPU oracle here (no GPU in this test); the
GPU version of the same exchange runs in bench.py --gpus N and in tests/test_gpu_parity.py::test_merge_samples_device."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import tatajuba_amd as tj
from oracle import orc
from tatajuba_amd.dist import all_gatherv_bytes, RECORD_BYTES


def _sample_records(rank):
    s = tj.synth_stream(4000 + 500 * rank, 150, 60000, seed_reads=0x7A7A1000 + rank, variant_seed=rank, n_threads=1)
    o = orc.Oracle(10)
    o.scan_stream(s, 3)
    o.finalise(1, 0)
    e = o.elems()
    r = np.zeros(len(e), dtype=tj.RECORD_DTYPE)
    for f in ("ctx0", "ctx1", "meta"):
        r[f] = e[f]
    return r


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = _sample_records(rank)
    local = torch.from_numpy(mine.view(np.uint8).copy())
    parts, sizes = all_gatherv_bytes(local, dist)
    assert sizes[rank] == mine.nbytes and all(s % RECORD_BYTES == 0 for s in sizes)
    allrec = torch.cat(parts).numpy()
    counts = [s // RECORD_BYTES for s in sizes]
    _, _, keys, mat = orc.merge_samples(np.frombuffer(allrec.tobytes(), dtype=np.uint64).reshape(-1, 3), counts)
    # every rank computes the same union
    digest = torch.tensor([int(mat.astype(np.int64).sum()), len(keys)], dtype=torch.int64)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    if rank == 0:
        q.put((counts, len(keys), mat.sum(axis=0).tolist(), (mat > 0).sum(axis=1).max()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_and_merge():
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    counts, n_union, colsum, max_share = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    recs = [_sample_records(r) for r in range(2)]
    assert counts == [len(r) for r in recs]
    for r, tot in zip(recs, colsum):
        assert int(tj.decode_meta(r["meta"])["count"].sum()) == tot          # every depth lands in its sample's column
    allkeys = set()
    for r in recs:
        d = tj.decode_meta(r["meta"])
        allkeys |= set(zip(d["base"].tolist(), r["ctx0"].tolist(), r["ctx1"].tolist(), d["length"].tolist()))
    assert n_union == len(allkeys) and max_share == 2                         # the two samples share tracts


def test_merge_oracle_order_and_counts():
    a, b = _sample_records(0), _sample_records(1)
    rec = np.frombuffer(np.concatenate([a, b]).tobytes(), dtype=np.uint64).reshape(-1, 3)
    cs, ci, keys, mat = orc.merge_samples(rec, [len(a), len(b)])
    d = tj.decode_meta(keys[:, 2])
    tup = list(zip(d["base"].tolist(), keys[:, 0].tolist(), keys[:, 1].tolist(), d["length"].tolist()))
    assert all(tup[i] > tup[i + 1] for i in range(len(tup) - 1))              # reference's descending order, no duplicates
    da = tj.decode_meta(a["meta"])
    first = (int(da["base"][0]), int(a["ctx0"][0]), int(a["ctx1"][0]), int(da["length"][0]))
    assert mat[tup.index(first), 0] == da["count"][0]
    assert mat.shape == (len(tup), 2) and (mat >= 0).all() and (mat.sum(axis=1) > 0).all()
    # the concatenated list keeps every record once, each sample's records in their own order
    assert len(cs) == len(a) + len(b) and (ci[cs == 0] == np.arange(len(a))).all() and (ci[cs == 1] == np.arange(len(b))).all()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` as the driver calls it (no external launcher): N fresh ranks, one JSON line from rank 0,
    the children's return code -- rehearsed without a GPU (`--rehearse`: rendezvous, barrier and reduction only)."""
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--rehearse", "--config", "4"], env=env, stdout=subprocess.PIPE, timeout=300)
    assert r.returncode == 0
    line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["max_over_ranks"] == 2.0
    assert "configs[3]" in d["config"]["workload"]                          # the label follows the arguments
    # a multi-GPU line describes its exchange (what the driver's 8-GPU run will carry; nothing is measured in a rehearsal)
    ex = d["stages"]["exchange"]
    assert set(ex) >= {"measured", "exchanges_per_rank", "allgather_ms", "merge_ms", "bytes_gathered_per_rank_and_exchange",
                       "collectives_per_exchange", "rccl_ranks_reported", "scan_ms_per_rank"} and ex["measured"] is False
    assert "rccl_ranks" in d["config"]
    r = subprocess.run([sys.executable, bench, "--rehearse", "--kmer", "13"], env=env, stdout=subprocess.PIPE, timeout=300)
    assert "custom workload" in json.loads(r.stdout.decode().splitlines()[-1])["config"]["workload"]


def test_bench_launcher_ends_the_job_when_a_rank_dies():
    # rank 1 exits before the rendezvous; rank 0 would wait for it for minutes: the launcher has to end it and fail
    import subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TATAJUBA_BENCH_REHEARSE_FAIL_RANK="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    t = time.time()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse"], env=env, capture_output=True, timeout=120)
    assert p.returncode != 0 and time.time() - t < 90
