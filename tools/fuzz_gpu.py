"""One-off differential fuzzing on the GPU box: random parameters and inputs, scan + finalise through the C ABI against
the CPU oracle (raw count, kept records byte for byte, index ranges, coverage, status).  Not part of the test suite.
usage: python tools/fuzz_gpu.py [seconds] [seed]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tatajuba_amd as tj
from oracle import orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = random.Random(seed)
nrng = np.random.default_rng(seed)
t_end = time.time() + budget
it = 0
fails = 0


def random_stream():
    mode = rng.choice(["synth", "synth", "synth_ragged", "alphabet", "lowcomplex"])
    if mode == "synth":
        n = rng.choice([50, 2000, 30000, 150000, 400000])
        L = rng.choice([36, 75, 150, 250])
        g = rng.choice([2000, 50000, 1000000, 20000000])
        return tj.synth_stream(n, L, g, seed_reads=rng.randrange(1 << 30), variant_seed=rng.randrange(8)), mode
    if mode == "synth_ragged":
        n = rng.choice([200, 3000, 20000])
        return tj.synth_stream(n, 100, rng.choice([50000, 3000000]), seed_reads=rng.randrange(1 << 30),
                               read_len_max=rng.choice([400, 3000, 20000])), mode
    if mode == "alphabet":
        ab = rng.choice(["ACGT", "ACGTN", "ACGTacgtUN-", "AT", "ACGTNNNN"])
        reads = []
        for _ in range(rng.choice([10, 300, 3000])):
            L = rng.randint(0, 400)
            s = []
            while len(s) < L:
                s.extend(rng.choice(ab) * rng.choice([1, 1, 1, 2, 3, 4, 6, 12, 70]))
            reads.append("".join(s[:L]))
        return np.frombuffer(("\n".join(reads) + "\n").encode("latin-1"), np.uint8), mode
    # low complexity: few distinct contexts, huge counts, long tracts (length wrap), skewed buckets
    unit = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 9)))
    reads = []
    for _ in range(rng.choice([100, 5000, 60000])):
        t = rng.choice("ACGT") * rng.choice([3, 5, 9, 30, 600, 1100])
        reads.append((unit * 12)[: rng.randint(5, 60)] + t + (unit * 12)[: rng.randint(5, 60)])
    return np.frombuffer(("\n".join(reads) + "\n").encode(), np.uint8), mode


while time.time() < t_end:
    it += 1
    k = rng.choice([int(x) for x in os.environ["FUZZ_K"].split(",")] if os.environ.get("FUZZ_K") else [2, 3, 5, 8, 10, 12, 13, 15, 20, 25, 28, 29, 31, 32])
    m = rng.choice([1, 2, 3, 4, 6])
    rb = rng.choice([0, 1])
    mc = rng.choice([0, 1, 3, 5, 50])
    parts = [random_stream() for _ in range(rng.choice([1, 1, 2, 3]))]
    c = tj.Counter(k)
    o = orc.Oracle(k)
    for s, _ in parts:
        c.scan_host(s, m)
        o.scan_stream(s, m)
    desc = (it, k, m, rb, mc, [md for _, md in parts], [int(s.size) for s, _ in parts])
    try:
        assert c.raw_count() == o.c.n_elem, "raw count"
        if rng.random() < 0.4:                            # the same in two calls (tjamd_finalise_begin / _end)
            if rng.random() < 0.5:                        # ... with the ordering step on a second stream
                import torch
                if "_order" not in globals():
                    globals()["_order"] = torch.cuda.Stream()
                c.set_order_stream(_order.cuda_stream)
            c.finalise_begin(rb, mc); st = c.finalise_end()
        else:
            st = c.finalise(rb, mc)
        o.finalise(rb, mc)
        assert st == o.c.status, f"status {st} vs {o.c.status}"
        if st == 0:
            assert c.n_kept == o.c.n_elem, "kept count"
            assert c.download_kept().tobytes() == o.elems().tobytes(), "kept bytes"
            gi, gf = c.download_idx(); ei, ef = o.idx()
            assert c.n_idx == o.c.n_idx and (gi == ei).all() and (gf == ef).all(), "idx"
            assert c.coverage == o.c.coverage, "coverage"
        # the counter again after finalise (buffers, parity counters, clears)
        s2, _ = parts[0]
        c.scan_host(s2, m); o2 = orc.Oracle(k); o2.scan_stream(s2, m)
        assert c.raw_count() == o2.c.n_elem, "raw count (reuse)"
        st2 = c.finalise(rb, mc); o2.finalise(rb, mc)
        assert st2 == o2.c.status and (st2 != 0 or c.download_kept().tobytes() == o2.elems().tobytes()), "reuse"
    except AssertionError as e:
        fails += 1
        print("FAIL", desc, e, flush=True)
    c.close()
    if it % 10 == 0:
        print(f"[{it}] ok so far, fails={fails}", flush=True)
print(f"done: {it} cases, {fails} failures (seed {seed})")
sys.exit(1 if fails else 0)
