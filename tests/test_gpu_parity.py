"""Parity of the HIP path with the CPU oracle, through the C ABI (every test needs an MI355X: -m gpu).

Bit-exact everywhere: the path is integer/byte work.  Raw records are compared as multisets (their order is
irrelevant before the sort -- reference: src/hopo_counter.c:351), located records in emission order, and the
finalised arrays byte for byte (all 40 bytes of every hopo_element, the index ranges and the coverage)."""
import os
import random

import numpy as np
import pytest

import tatajuba_amd as tj
from oracle import orc
from tests.pyref import scan_closed_form

pytestmark = pytest.mark.gpu


def rec_sorted(a):
    a = np.ascontiguousarray(a)
    o = np.lexsort((a["meta"], a["ctx1"], a["ctx0"]))
    return a[o]


def oracle_raw(stream, k, m):
    o = orc.Oracle(k)
    o.scan_stream(stream, m)
    return o


def as_records(elems):
    r = np.zeros(len(elems), dtype=tj.RECORD_DTYPE)
    for f in ("ctx0", "ctx1", "meta"):
        r[f] = elems[f]
    return r


def check_raw_multiset(stream, k, m):
    stream = np.frombuffer(stream, np.uint8) if not isinstance(stream, np.ndarray) else stream
    c = tj.Counter(k)
    c.scan_host(stream, m)
    got = c.download_raw()
    o = oracle_raw(stream, k, m)
    exp = as_records(o.elems())
    assert len(got) == len(exp), (len(got), len(exp))
    assert (rec_sorted(got) == rec_sorted(exp)).all()
    assert c.undefined_runs() == o.c.n_undefined
    c.close()
    return len(got)


def check_finalise(stream_parts, k, m, remove_biased, min_cov):
    """stream_parts: list of uint8 arrays scanned one after the other into the same counter"""
    c = tj.Counter(k)
    o = orc.Oracle(k)
    for s in stream_parts:
        c.scan_host(s, m)
        o.scan_stream(s, m)
    assert c.raw_count() == o.c.n_elem
    st = c.finalise(remove_biased, min_cov)
    o.finalise(remove_biased, min_cov)
    assert st == o.c.status
    if st in (1, 2):
        c.close()
        return st, 0
    got, exp = c.download_kept(), o.elems() if st == 0 else None
    if st == 0:
        assert c.n_kept == o.c.n_elem and got.tobytes() == exp.tobytes()
        gi, gf = c.download_idx()
        ei, ef = o.idx()
        assert c.n_idx == o.c.n_idx and (gi == ei).all() and (gf == ef).all()
        assert c.coverage == o.c.coverage
    n = c.n_kept
    c.close()
    return st, n


# ---- known answers through the drop-in CPU entry ------------------------------------------------------------------

def test_known_answers_update_from_seq(known_answers):
    for case in known_answers["figure"] + known_answers["scan"]:      # "figure": the reference's own drawing (recipe/200322_001.png)
        h = tj.HopoCounter.new(case["k"])
        h.update_from_seq(case["seq"], case["m"])
        e = h.elems()
        d = tj.decode_meta(e["meta"])
        got = [(int(d["base"][i]), int(d["length"][i]), int(e["read_offset"][i]), int(d["canon_flag"][i]),
                int(e["ctx0"][i]), int(e["ctx1"][i])) for i in range(len(e))]
        exp = [(r[0], r[1], r[2], r[3], int(r[4], 16), int(r[5], 16)) for r in case["records"]]
        assert got == exp, case["seq"]
        assert (d["count"] == 1).all() and (d["mismatches"] == -2).all() and (e["loc_pos"] == -1).all()
        o = orc.Oracle(case["k"])
        o.scan_seq(case["seq"], case["m"])
        assert e.tobytes() == o.elems().tobytes()          # all 40 bytes of every element
        h.delete()


def test_reference_figure_through_the_bucket_scan(known_answers):
    """The three tracts of the reference's figure (recipe/200322_001.png, README.md:218-232), many times over, through the
    bucket scan (tiles of the fast kernel and of the generic one): exactly the drawn contexts, bases, lengths and strands."""
    fig = known_answers["figure"]
    reads = [c["seq"] for c in fig] * 4000
    random.Random(5).shuffle(reads)
    s = np.frombuffer(("\n".join(reads) + "\n").encode(), np.uint8)
    c = tj.Counter(3)
    c.scan_host(s, 2)
    got = c.download_raw()
    c.close()
    d = tj.decode_meta(got["meta"])
    seen = {}
    for i in range(len(got)):
        key = (int(d["base"][i]), int(d["length"][i]), int(d["canon_flag"][i]), int(got["ctx0"][i]), int(got["ctx1"][i]))
        seen[key] = seen.get(key, 0) + 1
    exp = {(r[0], r[1], r[3], int(r[4], 16), int(r[5], 16)): 4000 for c in fig for r in c["records"]}
    assert seen == exp
    assert len(got) == len(as_records(oracle_raw(s, 3, 2).elems()))


@pytest.mark.parametrize("both_strands,remove_biased", [(True, 1), (True, 0), (False, 0)])
def test_reference_second_figure_through_finalise_and_the_context_histogram(known_answers, tmp_path, both_strands, remove_biased):
    """The reference's second figure (recipe/200322_002.png, README.md:234-240: context CCG|GAT, base A, reads per tract
    length 2: 10, 3: 20, 4: 6, 5: 2, "a typical length would be 3") through the HIP path: bucket scan + finalise (count =
    multiplicity per context and length, src/hopo_counter.c:356-365), tjamd_context_histograms (length histogram, highest
    count first, src/context_histogram.c:278-286) and the drop-in new_genomic_context_list on a FASTQ file of those reads."""
    from tatajuba_amd.capi import Options
    from tests.pyref import figure2_expected, figure2_reads
    f2 = known_answers["figure2"]
    k, m = f2["k"], f2["m"]
    reads = figure2_reads(f2, both_strands)
    random.Random(2).shuffle(reads)
    elems_exp, hist_exp = figure2_expected(f2)
    s = np.frombuffer(("\n".join(reads) + "\n").encode(), np.uint8)
    c = tj.Counter(k)
    c.scan_host(s, m)
    assert c.raw_count() == len(reads)
    assert c.finalise(remove_biased, 5) == 0
    kept = c.download_kept()
    d = tj.decode_meta(kept["meta"])
    assert [(int(d["length"][i]), int(d["count"][i])) for i in range(len(kept))] == elems_exp
    assert (kept["ctx0"] == int(f2["ctx0"], 16)).all() and (kept["ctx1"] == int(f2["ctx1"], 16)).all() and (d["base"] == f2["base_code"]).all()
    assert (d["canon_flag"] == (3 if both_strands else 1)).all()
    gi, gf = c.download_idx()
    assert c.n_idx == 1 and (gi[0], gf[0]) == (0, 4) and c.coverage == 38
    got = c.context_histograms(1, 2)
    (g,) = got["groups"]
    assert (g["first"], g["n_elem"], g["n_context"], g["n_len"], g["integral"]) == (0, 4, 1, 4, 38)
    assert list(zip(got["hist"]["length"][:4].tolist(), got["hist"]["freq"][:4].tolist())) == hist_exp
    assert g["modal_len"] == f2["typical_length"] and g["modal_freq"] == 20
    c.close()
    # the same through the drop-in API: a file of those reads -> genomic_context_list with one context_histogram_t
    fq = str(tmp_path / "figure2.fq")
    with open(fq, "w") as fh:
        fh.write("".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads)))
    opt = Options.defaults(k, m, 5, bool(remove_biased))
    opt.max_distance_per_flank, opt.levenshtein_distance = 1, 2
    L = tj.lib()
    h = tj.HopoCounter.new_or_append_from_file(None, fq, opt)
    gl = L.new_genomic_context_list(h._p)
    assert gl, "sample excluded"
    gg = gl.contents
    assert gg.n_hist == 1 and gg.coverage == 38
    ch = gg.hist[0].contents
    assert ch.name.decode() == "%s.%s.%s" % (f2["left"], f2["base"], f2["right"]) and ch.base == f2["base_code"]
    assert (ch.n_context, ch.integral, ch.mode_context_count, ch.mode_context_length) == (1, 38, 20, f2["typical_length"])
    assert (ch.coverage, ch.n_tracts) == (38, 1)
    hh = ch.h.contents
    assert [(hh.i[t].idx, hh.i[t].freq) for t in range(hh.n)] == hist_exp
    L.del_genomic_context_list(gl)
    h.delete()


def test_stale_context_and_undefined(known_answers):
    c0, c1 = known_answers["stale_context"]
    h = tj.HopoCounter.new(c0["k"])
    h.update_from_seq(c0["seq"], c0["m"])
    o = orc.Oracle(c0["k"])
    o.scan_seq(c0["seq"], c0["m"])
    assert h.c.n_elem == c0["n_records"] and h.elems().tobytes() == o.elems().tobytes()
    h.delete()
    c = tj.Counter(c1["k"])
    assert len(c.scan_host_located((c1["seq"] + "\n").encode(), c1["m"])) == 0
    c.close()


def test_update_from_seq_appends_and_m2_rescan():
    # the reference rescans reference windows with m = 2 (src/genome_set.c:539)
    h = tj.HopoCounter.new(4)
    o = orc.Oracle(4)
    for seq in ("ACGTAACCGGTTACGTACGT", "TTGACCCCAGTAAGTC", "ACG", "ACGTTTTTTTTTTACGTAC"):
        h.update_from_seq(seq, 2)
        o.scan_seq(seq, 2)
    assert h.c.n_elem == o.c.n_elem > 3 and h.elems().tobytes() == o.elems().tobytes()
    h.delete()


def test_update_from_seq_all_monomers():
    # reference: update_hopo_counter_from_seq_all_monomers (src/hopo_counter.c:260-283; caller src/genome_set.c:543)
    rng = random.Random(5)
    for k in (2, 3, 10, 25, 32):
        h = tj.HopoCounter.new(k)
        o = orc.Oracle(k)
        for it in range(12):
            ab = ["ACGT", "ACGTN", "ACGTacgtUN-", "AT"][it % 4]
            L = rng.choice([0, 5, 70, 300, 5000, 20000])
            s = []
            while len(s) < L:
                s.extend(rng.choice(ab) * rng.choice([1, 1, 1, 1, 2, 3, 5]))
            seq = "".join(s[:L])
            if it == 7:
                seq = "ACGT" * 3000               # a monomer at every position: one candidate per byte of a tile
            h.update_from_seq_all_monomers(seq)
            o.scan_seq_all_monomers(seq)
            if it % 3 == 0:                       # tract and monomer rescans interleave in the reference's caller
                h.update_from_seq(seq, 2)
                o.scan_seq(seq, 2)
        assert h.c.n_elem == o.c.n_elem > 1000, k
        assert h.elems().tobytes() == o.elems().tobytes(), k
        h.delete()


# ---- random strings: located records in emission order, every alphabet quirk --------------------------------------

def test_random_reads_located_order():
    rng = random.Random(99)
    alphabets = ["ACGT", "ACGT", "AC", "ACGTN", "ACGTacgtUN-", "AT", "ACGTNNN"]
    for k, m in [(3, 3), (2, 1), (5, 2), (10, 3), (16, 4), (25, 4), (31, 3), (32, 6)]:
        reads = []
        for it in range(400):
            ab = alphabets[it % len(alphabets)]
            L = rng.randint(0, 260)
            s = []
            while len(s) < L:
                s.extend(rng.choice(ab) * rng.choice([1, 1, 1, 2, 3, 4, 5, 9, 40]))
            reads.append("".join(s[:L]))
        stream = ("\n".join(reads) + "\n").encode("latin-1")
        c = tj.Counter(k)
        loc = c.scan_host_located(stream, m)
        starts = np.cumsum([0] + [len(r) + 1 for r in reads])
        exp = []
        for r, st in zip(reads, starts):
            for (base, n, off, flag, c0, c1) in scan_closed_form(r, k, m):
                meta = base | ((n & 0x3ff) << 2) | (1 << 12) | (0xffe << 32) | (flag << 49)
                exp.append((c0, c1, meta, st + off + k))
        got = [(int(x["ctx0"]), int(x["ctx1"]), int(x["meta"]), int(x["pos"])) for x in loc]
        assert got == exp, (k, m)
        o = oracle_raw(np.frombuffer(stream, np.uint8), k, m)
        assert len(got) == o.c.n_elem
        c.close()


# ---- raw multiset parity -----------------------------------------------------------------------------------------

@pytest.mark.parametrize("k,m", [(10, 3), (15, 4), (25, 4), (2, 1), (32, 5), (7, 2)])
def test_raw_fixture_file(golden_dir, known_answers, k, m):
    s, n = tj.read_file_stream(os.path.join(golden_dir, "err1750956.fastq.gz"))
    nrec = check_raw_multiset(s, k, m)
    for case in known_answers["file"]["cases"]:
        if (case["k"], case["m"]) == (k, m):
            assert nrec == case["raw"]


@pytest.mark.parametrize("k,m,L", [(10, 3, 150), (15, 4, 150), (25, 4, 150), (10, 3, 37), (4, 2, 9)])
def test_raw_synthetic_short_reads(k, m, L):
    s = tj.synth_stream(20000, L, 300000)
    check_raw_multiset(s, k, m)


def test_raw_long_ragged_reads_and_tile_edges():
    s = tj.synth_stream(300, 2000, 400000, read_len_max=20000)
    check_raw_multiset(s, 25, 4)
    check_raw_multiset(s, 10, 3)
    # tracts longer than the right halo, longer than a tile, wrapping the 10-bit length, at every alignment
    rng = np.random.default_rng(3)
    parts = []
    for i in range(60):
        pre = "".join(rng.choice(list("ACGT"), size=int(rng.integers(30, 5000))))
        run = "ACGTN"[i % 5] * int([3, 17, 190, 193, 400, 511, 512, 600, 4095, 4096, 4097, 9000][i % 12])
        post = "".join(rng.choice(list("ACGT"), size=int(rng.integers(0, 70))))
        parts.append(pre + run + post)
    stream = ("\n".join(parts) + "\n").encode()
    for k, m in [(3, 3), (10, 3), (32, 4)]:
        check_raw_multiset(stream, k, m)


def test_raw_edge_streams():
    for stream in [b"", b"\n", b"\n\n\n", b"ACGT", b"ACGTAAAACGTA", b"ACGTAAAACGTA\n", b"A" * 5000 + b"\n",
                   b"ACG\n" * 3000, b"AAAA\nCCCC\nGGGG\n", b"ACGTTTTTACGT\n\nACGTTTTTACGT\n",
                   b"NNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN\n" * 10, b"ACGAAAACGTNNNNACGTNNNNNACGT\nACGNNNNACGT\n"]:
        a = np.frombuffer(stream, np.uint8)
        if a.size == 0:
            c = tj.Counter(3)
            c.scan_host(a, 3)
            assert c.raw_count() == 0
            c.close()
            continue
        check_raw_multiset(a, 3, 3)
        check_raw_multiset(a, 2, 2)


def test_long_stream_is_scanned_in_pieces(monkeypatch):
    # a stream longer than 1.5 pieces is cut after aligned read delimiters and scanned piece by piece with exact counts
    # in between (the piece size is 1 GiB; the environment hook makes it 200 kB here)
    torch = pytest.importorskip("torch")
    monkeypatch.setenv("TATAJUBA_AMD_SCAN_PIECE", "200000")
    for k, m, s in [(10, 3, tj.synth_stream(40000, 150, 300000)), (25, 4, tj.synth_stream(3000, 500, 300000, read_len_max=9000)),
                    (3, 2, np.frombuffer((b"ACGGGTTTACAACCCGT\n" * 70000), np.uint8)),          # 18-byte reads: delimiters on every residue
                    (4, 3, np.frombuffer((b"ACGGGTTTACAACCCG\n" * 40000) + b"AAAACCCCGGGGTTTT" * 20000 + b"\n", np.uint8))]:
        d = torch.from_numpy(np.ascontiguousarray(s)).cuda()
        c = tj.Counter(k)
        c.scan_device(d.data_ptr(), s.size, m)
        o = oracle_raw(s, k, m)
        assert c.raw_count() == o.c.n_elem > 0
        st = c.finalise(0, 0)
        o.finalise(0, 0)
        assert st == o.c.status == 0
        assert c.download_kept().tobytes() == o.elems().tobytes()
        assert c.last_scan_launches() >= (3 if s.size > 700000 else 1)
        c.close()


def test_scan_device_pointer_api_and_multiple_batches():
    torch = pytest.importorskip("torch")
    s = tj.synth_stream(30000, 150, 300000)
    t = torch.from_numpy(s.copy()).cuda()
    c = tj.Counter(10)
    c.set_stream(torch.cuda.current_stream().cuda_stream)
    half = (s.size // 2 // 151) * 151                    # whole reads, 16-byte aligned start for the 2nd half? no:
    half = (half // (151 * 16)) * (151 * 16)             # multiple of 16 bytes and of whole reads
    c.scan_device(t.data_ptr(), half, 3)
    c.scan_device(t.data_ptr() + half, s.size - half, 3)
    got = c.download_raw()
    exp = as_records(oracle_raw(s, 10, 3).elems())
    assert (rec_sorted(got) == rec_sorted(exp)).all()
    assert c.last_scan_ms() > 0
    with pytest.raises(tj.TatajubaAmdError, match="aligned"):
        c.scan_device(t.data_ptr() + 3, 100, 3)
    c.close()


# ---- finalise: byte-identical elem[], idx_initial/final, coverage --------------------------------------------------

@pytest.mark.parametrize("case", range(4))
def test_finalise_fixture_cases(golden_dir, known_answers, case):
    f = known_answers["file"]
    cs = f["cases"][case]
    s, _ = tj.read_file_stream(os.path.join(golden_dir, f["path"]))
    st, n = check_finalise([s], cs["k"], cs["m"], cs["remove_biased"], cs["min_coverage"])
    assert st == 0 and n == cs["n_elem"]
    st, n = check_finalise([s, s], cs["k"], cs["m"], cs["remove_biased"], cs["min_coverage"])   # "paired"
    assert st == 0


@pytest.mark.parametrize("k,m,rb,mc", [(10, 3, 1, 5), (10, 3, 0, 5), (15, 4, 1, 5), (25, 4, 1, 5), (25, 4, 0, 0),
                                       (2, 1, 1, 0), (32, 4, 1, 3)])
def test_finalise_synthetic(k, m, rb, mc):
    s = tj.synth_stream(20000, 150, 100000)              # depth 30: most tracts seen on both strands
    st, n = check_finalise([s], k, m, rb, mc)
    assert st == 0 and n > 0


@pytest.mark.parametrize("k,m", [(10, 3), (12, 2), (15, 4), (31, 3)])
def test_streams_with_no_calls_match_the_oracle(k, m):
    # 'N' is what real reads hold besides ACGT.  The fast kernel gives a tile with one a second classification (N let
    # through, an N plane for the flanks read backwards) and leaves only countable runs of N -- and every other byte -- to
    # the general kernel; whichever kernel takes a tile, the records are the oracle's
    s = tj.synth_stream(60000, 150, 200000).copy()        # ~550 fast-kernel tiles, depth 45
    rng = np.random.default_rng(k * 100 + m)
    reads = rng.choice(60000, 3000, replace=False)
    s[reads[:2400] * 151 + rng.integers(0, 150, 2400)] = ord("N")                 # isolated no-calls
    for r in reads[2400:2800]:                                                    # runs of N (stale-context rule)
        p, n = int(rng.integers(0, 140)), int(rng.integers(2, 9))
        s[r * 151 + p: r * 151 + min(p + n, 150)] = ord("N")
    for r in reads[2800:]:                                                        # lower case, U, IUPAC codes
        p = int(rng.integers(0, 147))
        s[r * 151 + p: r * 151 + p + 3] = np.frombuffer(rng.choice([b"acg", b"UUU", b"RYK", b"nnn", b"tTt"]), np.uint8)
    n = check_raw_multiset(s, k, m)
    assert n > 10000
    st, kept = check_finalise([s], k, m, 1, 3)
    assert st == 0 and kept > 100
    if os.environ.get("TATAJUBA_AMD_FAST", "1") == "1":
        # isolated no-calls alone: the fast kernel keeps (nearly) every tile -- they are classified a second time, not handed over
        import ctypes as C
        s2 = tj.synth_stream(60000, 150, 200000).copy()
        s2[reads[:2400] * 151 + rng.integers(0, 150, 2400)] = ord("N")
        c = tj.Counter(k)
        c.scan_host(s2, m)
        L = tj.lib()
        L.tjamd_debug_slow_tiles.restype = C.c_long; L.tjamd_debug_slow_tiles.argtypes = [C.c_void_p]
        n_tiles = (s2.size + 16287) // 16288
        assert 0 <= L.tjamd_debug_slow_tiles(c._h) <= n_tiles // 20, (L.tjamd_debug_slow_tiles(c._h), n_tiles)
        got = c.download_raw()
        assert (rec_sorted(got) == rec_sorted(as_records(oracle_raw(s2, k, m).elems()))).all()
        c.close()


@pytest.mark.parametrize("mode", ["plan", "noplan", "cap", "plan+order", "cap+order", "nofuse"])
def test_finalise_in_two_calls_with_two_counters_on_one_stream(monkeypatch, mode):
    # tjamd_finalise_begin / _end: sample i's outcome is fetched after sample i + 1 has been queued on the same stream
    # "+order": the ordering step of a begun finalise on a second stream (tjamd_counter_set_order_stream), as bench.py runs it
    order_stream = mode.endswith("+order")
    mode = mode.split("+")[0]
    if mode == "noplan":
        monkeypatch.setenv("TATAJUBA_AMD_NO_PLAN", "1")
    if mode == "cap":
        monkeypatch.setenv("TATAJUBA_AMD_PLAN_CAP", "300")
    if mode == "nofuse":                                     # (the ordering step counts its bins itself, as in round 2)
        monkeypatch.setenv("TATAJUBA_AMD_NO_FUSED_BINS", "1")
    import torch
    streams = [tj.synth_stream(15000 + 3000 * i, 150, 80000, seed_reads=77 + i) for i in range(4)] + [np.frombuffer(b"ACGT\n", np.uint8)]
    devs = [torch.from_numpy(s.copy()).cuda() for s in streams]
    ctr = [tj.Counter(10), tj.Counter(10)]
    # a stream of torch's making, as bench.py does: the current stream's handle is null, which tjamd_counter_set_stream reads
    # as "the counter's own stream" -- two counters on ONE non-null stream is the configuration the headline bench runs
    shared = torch.cuda.Stream()
    assert shared.cuda_stream != 0
    torch.cuda.synchronize()                                # (the uploads above went through the default stream)
    order = torch.cuda.Stream() if order_stream else None
    for c in ctr:
        c.set_stream(shared.cuda_stream)
        if order is not None:
            c.set_order_stream(order.cuda_stream)
    begun, got = None, []

    def end(i):
        c = ctr[i & 1]
        st = c.finalise_end()
        got.append((i, st, c.download_kept().tobytes() if st == 0 else b"", c.coverage if st == 0 else 0))

    for i, d in enumerate(devs):
        c = ctr[i & 1]
        c.reset()
        c.scan_device(d.data_ptr(), streams[i].size, 3)
        c.finalise_begin(1, 3)
        if begun is not None:
            end(begun)
        begun = i
    end(begun)
    assert [g[0] for g in got] == list(range(len(streams)))
    for i, st, kept, cov in got:
        o = orc.Oracle(10)
        o.scan_stream(streams[i], 3)
        o.finalise(1, 3)
        assert st == o.c.status, i
        if st == 0:
            assert kept == o.elems().tobytes() and cov == o.c.coverage, i
    assert got[-1][1] == 1                                  # (the stream without a tract: the reference's "empty sample")
    # the index ranges of the last full sample, through the same two-call path
    c = ctr[0]
    c.reset(); c.scan_device(devs[2].data_ptr(), streams[2].size, 3); c.finalise_begin(1, 3)
    ctr[1].reset(); ctr[1].scan_device(devs[3].data_ptr(), streams[3].size, 3); ctr[1].finalise_begin(1, 3)
    assert c.finalise_end() == 0 and ctr[1].finalise_end() == 0
    o = orc.Oracle(10); o.scan_stream(streams[2], 3); o.finalise(1, 3)
    gi, gf = c.download_idx(); ei, ef = o.idx()
    assert (gi == ei).all() and (gf == ef).all() and c.download_kept().tobytes() == o.elems().tobytes()
    with pytest.raises(Exception):
        ctr[0].finalise_end()                               # nothing begun
    for c in ctr:
        assert c.plan_mismatches() == 0                     # (the in-kernel reading of the kept count agreed with the kernel boundary's every time)
        c.close()


@pytest.mark.parametrize("plan", ["cap", "off"])
def test_finalise_when_more_is_kept_than_planned_for(monkeypatch, plan):
    # the ordering step is launched before the host knows how many records were kept, with buffers for a guess; a sample
    # that keeps more makes those kernels return at once and the step runs again with the count in hand
    if plan == "cap":
        monkeypatch.setenv("TATAJUBA_AMD_PLAN_CAP", "500")
    else:
        monkeypatch.setenv("TATAJUBA_AMD_NO_PLAN", "1")
    s = tj.synth_stream(20000, 150, 100000)
    st, n = check_finalise([s], 10, 3, 1, 5)
    assert st == 0 and n > 500
    st, n = check_finalise([s[: 151 * 2000], s[151 * 2000:]], 15, 4, 0, 0)
    assert st == 0 and n > 500


@pytest.mark.parametrize("k,m,limit", [(10, 3, 1), (25, 4, 2), (32, 3, 1), (2, 1, 1)])
def test_finalise_radix_fallback(monkeypatch, k, m, limit):
    # the kept set is ordered by a bin partition + rank sort; a bin fuller than the limit must switch the whole sort to
    # the stable radix passes (the environment hook makes every bin "too full")
    monkeypatch.setenv("TATAJUBA_AMD_BIN_MAX", str(limit))
    st, n = check_finalise([tj.synth_stream(20000, 150, 100000)], k, m, 1, 3)
    assert st == 0 and n > 0


def test_finalise_skewed_bins():
    # contexts that share their leading bases land in few bins: poly-A flanks with variation far from the tract
    rng = random.Random(3)
    reads = []
    for i in range(3000):
        far = "".join(rng.choice("ACGT") for _ in range(6))
        tract = "C" * rng.choice([3, 4, 5, 6])
        r = far + "A" * 14 + tract + "A" * 14 + far[::-1]
        reads.append(r)
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        reads.append("".join(comp[ch] for ch in reversed(r)))
    s = np.frombuffer(("\n".join(reads) + "\n").encode(), np.uint8)
    for k in (16, 20):
        st, n = check_finalise([s], k, 3, 1, 0)
        assert st == 0 and n > (1000 if k == 20 else 10)


@pytest.mark.parametrize("k,m", [(10, 3), (20, 3), (31, 4)])
def test_finalise_many_distinct_keys_leftover_rounds(k, m):
    # a large genome at low depth: several thousand distinct keys per hash bucket, more than one round of the
    # aggregation's LDS table holds (the records of the keys that do not fit go round again)
    s = tj.synth_stream(700000, 150, 40_000_000)
    st, n = check_finalise([s], k, m, 0, 0)
    assert st == 0 and n > 150000


def test_finalise_statuses_and_count_semantics():
    assert check_finalise([np.frombuffer(b"ACGT\n", np.uint8)], 3, 3, 1, 5)[0] == 1
    assert check_finalise([np.frombuffer(b"CCGAAAAGAT\n", np.uint8)], 3, 3, 1, 0)[0] == 2
    assert check_finalise([np.frombuffer(b"CCGAAAAGAT\nATCTTTTCGG\n", np.uint8)], 3, 3, 1, 5)[0] == 3
    assert check_finalise([np.frombuffer(b"CCGAAAAGAT\nATCTTTTCGG\n", np.uint8)], 3, 3, 1, 2) == (0, 1)
    assert check_finalise([np.frombuffer(b"CCGAAAAGAT\nCCGAAAAGAT\nCCGCCCCGAT\n", np.uint8)], 3, 3, 0, 0) == (0, 1)
    # signed 10-bit length order: a 600-base tract stores -424 and sorts after the short ones
    s = ("AC" + "G" * 600 + "AC\n" + "GT" + "C" * 600 + "GT\n" + "ACGGGAC\nGTCCCGT\n").encode()
    assert check_finalise([np.frombuffer(s, np.uint8)], 2, 3, 1, 0)[0] == 0
    # 20-bit count wrap: 2^19 copies of the same tract on each strand -> count 2^20 wraps to 0
    one = np.frombuffer(b"CCGAAAAGAT\nATCTTTTCGG\n", np.uint8)
    big = np.tile(one, 1 << 19)
    st, n = check_finalise([big, np.frombuffer(b"TTGCCCCAGT\nACTGGGGCAA\n", np.uint8)], 3, 3, 1, 0)
    assert st == 0 and n == 2


# ---- the drop-in file API -----------------------------------------------------------------------------------------

def test_dropin_from_file_single_and_paired(golden_dir, known_answers):
    f = known_answers["file"]
    path = os.path.join(golden_dir, f["path"])
    for cs in f["cases"]:
        opt = tj.Options.defaults(cs["k"], cs["m"], cs["min_coverage"], bool(cs["remove_biased"]))
        h = tj.HopoCounter.new_or_append_from_file(None, path, opt)
        assert h.c.n_elem == cs["raw"] and h.c.name == path.encode() and h.c.opt.kmer_size == cs["k"]
        h.finalise()
        o = orc.Oracle(cs["k"])
        o.scan_file(path, cs["m"])
        o.finalise(cs["remove_biased"], cs["min_coverage"])
        assert (h.c.n_elem, h.c.n_alloc, h.c.n_idx, h.c.coverage) == (cs["n_elem"], cs["n_elem"], cs["n_idx"], cs["coverage"])
        assert h.elems().tobytes() == o.elems().tobytes()
        gi, gf = h.idx()
        ei, ef = o.idx()
        assert (gi == ei).all() and (gf == ef).all()
        h.delete()
    # paired: R2 appended into the same counter (reference: src/genome_set.c:72-73)
    opt = tj.Options.defaults(10, 3, 5, True, paired_end=True)
    h = tj.HopoCounter.new_or_append_from_file(None, path, opt)
    tj.HopoCounter.new_or_append_from_file(h, path, opt)
    assert h.c.n_elem == f["paired_same_file_twice_k10_m3"]["raw"]
    h.finalise()
    assert h.c.n_elem == f["paired_same_file_twice_k10_m3"]["n_elem"]
    o = orc.Oracle(10)
    o.scan_file(path, 3); o.scan_file(path, 3)
    o.finalise(1, 5)
    assert h.elems().tobytes() == o.elems().tobytes() and h.c.coverage == o.c.coverage
    h.delete()


def test_dropin_mixed_host_and_device_records(tmp_path):
    p = str(tmp_path / "r.fa")
    reads = ["TTCCGAAAAGATTT", "TTATCTTTTCGGTT", "GGCCGAAAAGATGG", "ACGTACGT"]
    open(p, "w").write("".join(f">r{i}\n{r}\n" for i, r in enumerate(reads)))
    opt = tj.Options.defaults(3, 3, 0, True)
    h = tj.HopoCounter.new_or_append_from_file(None, p, opt)
    h.update_from_seq("AAATCTTTTCGGAA", 3)
    o = orc.Oracle(3)
    for r in reads + ["AAATCTTTTCGGAA"]:
        o.scan_seq(r, 3)
    assert h.c.n_elem == o.c.n_elem
    h.finalise()
    o.finalise(1, 0)
    assert h.elems().tobytes() == o.elems().tobytes() and h.c.coverage == o.c.coverage
    h.delete()


def test_dropin_one_thread_per_sample(tmp_path):
    # the reference's caller is an OpenMP loop with one sample per thread (src/genome_set.c:66-94): six counters are
    # filled, finalised and re-scanned concurrently on the one GPU (ctypes releases the GIL around every call)
    import threading
    jobs = []
    for i in range(6):
        k, m = [(10, 3), (25, 4), (32, 3)][i % 3]
        s = tj.synth_stream(30000 + 4000 * i, 120 + 10 * i, 50000 + 1000 * i)
        path = str(tmp_path / f"s{i}.fa")
        with open(path, "wb") as fh:
            fh.write(b"".join(b">r\n" + r + b"\n" for r in bytes(s).split(b"\n") if r))
        jobs.append((path, k, m, s))
    got, errs = [None] * len(jobs), []

    def work(i):
        try:
            path, k, m, _ = jobs[i]
            opt = tj.Options.defaults(k, m, 3, True, paired_end=True)
            h = tj.HopoCounter.new_or_append_from_file(None, path, opt)
            tj.HopoCounter.new_or_append_from_file(h, path, opt)
            h.finalise()
            w = tj.HopoCounter.new(k)
            w.update_from_seq("ACGTTGCAAGGCTTTTTTAGGCATCGATCGGGATCGATTTAGCTAGCAAAACTAGCTAGCTAGGGGGCTAGCATCGATCGAT" * 3, 2)
            got[i] = (h.c.n_elem, h.c.coverage, h.elems().tobytes(), w.elems().tobytes())
            w.delete()
            h.delete()
        except Exception as e:                      # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for i, (path, k, m, s) in enumerate(jobs):
        o = orc.Oracle(k)
        o.scan_stream(s, m); o.scan_stream(s, m)
        o.finalise(1, 3)
        w = orc.Oracle(k)
        w.scan_seq("ACGTTGCAAGGCTTTTTTAGGCATCGATCGGGATCGATTTAGCTAGCAAAACTAGCTAGCTAGGGGGCTAGCATCGATCGAT" * 3, 2)
        assert got[i][0] == o.c.n_elem > 0 and got[i][1] == o.c.coverage, i
        assert got[i][2] == o.elems().tobytes() and got[i][3] == w.elems().tobytes(), i


def test_dropin_large_file_batches(tmp_path):
    # > 64 MiB of sequence so that the double-buffered batching is exercised
    s = tj.synth_stream(600000, 150, 2000000)
    p = str(tmp_path / "big.fa")
    with open(p, "wb") as fh:
        m = s.reshape(-1, 151)
        hdr = np.frombuffer(b">r\n", np.uint8)
        out = np.empty((m.shape[0], 3 + 151), np.uint8)
        out[:, :3] = hdr
        out[:, 3:] = m
        fh.write(out.tobytes())
    opt = tj.Options.defaults(10, 3, 5, True)
    h = tj.HopoCounter.new_or_append_from_file(None, p, opt)
    o = orc.Oracle(10)
    o.scan_stream(s, 3)
    assert h.c.n_elem == o.c.n_elem
    h.finalise()
    o.finalise(1, 5)
    assert h.elems().tobytes() == o.elems().tobytes() and h.c.coverage == o.c.coverage
    gi, gf = h.idx()
    ei, ef = o.idx()
    assert (gi == ei).all() and (gf == ef).all()
    h.delete()


@pytest.mark.parametrize("kind", ["gzip", "bgzf"])
def test_dropin_gzip_files_through_the_feeder(tmp_path, monkeypatch, kind):
    # a .fastq.gz of more than 4 MiB: inflated a window ahead of the parse (BGZF: members side by side), same counter
    import ctypes as C
    import gzip
    from tests.test_cabi import _bgzf
    monkeypatch.setenv("TATAJUBA_AMD_FEEDER_THREADS", "6")
    s = tj.synth_stream(120000, 150, 1000000)
    reads = bytes(s).split(b"\n")[:-1]
    rng = np.random.default_rng(8)
    q = np.frombuffer(b"#-27<AFI@>+", np.uint8)[rng.integers(0, 11, size=150 * len(reads))].tobytes()
    txt = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, q[150 * i:150 * i + 150]) for i, r in enumerate(reads))
    p = str(tmp_path / (kind + ".fq.gz"))
    with open(p, "wb") as fh:
        fh.write(gzip.compress(txt, 1) if kind == "gzip" else _bgzf(txt))
    assert os.path.getsize(p) > (4 << 20)
    opt = tj.Options.defaults(10, 3, 3, True)
    h = tj.HopoCounter.new_or_append_from_file(None, p, opt)
    L = tj.lib()
    L.tjamd_debug_feeder_stats.restype = C.c_long
    L.tjamd_debug_feeder_bgzf_blocks.restype = C.c_long
    fb = C.c_long(-1)
    assert L.tjamd_debug_feeder_stats(C.byref(fb)) >= 1 and fb.value == 0
    assert (L.tjamd_debug_feeder_bgzf_blocks() > 500) == (kind == "bgzf")
    o = orc.Oracle(10)
    o.scan_stream(s, 3)
    assert h.c.n_elem == o.c.n_elem
    h.finalise()
    o.finalise(1, 3)
    assert h.elems().tobytes() == o.elems().tobytes() and h.c.coverage == o.c.coverage
    h.delete()


def test_dropin_plain_fastq_through_the_multithreaded_feeder(tmp_path, monkeypatch):
    # an uncompressed FASTQ of > 32 MiB goes through feeder.c (several readers over the mapped file); quality strings
    # that look like headers make its range-start guesses work for their living
    import ctypes as C
    monkeypatch.setenv("TATAJUBA_AMD_FEEDER_THREADS", "6")
    s = tj.synth_stream(250000, 150, 1000000)
    reads = bytes(s).split(b"\n")[:-1]
    p = str(tmp_path / "plain.fq")
    with open(p, "wb") as fh:
        fh.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, (b"@" if i % 3 else b"I") * len(r)) for i, r in enumerate(reads)))
    assert os.path.getsize(p) > (32 << 20)
    opt = tj.Options.defaults(10, 3, 3, True)
    h = tj.HopoCounter.new_or_append_from_file(None, p, opt)
    L = tj.lib()
    L.tjamd_debug_feeder_stats.restype = C.c_long
    fb = C.c_long(-1)
    assert L.tjamd_debug_feeder_stats(C.byref(fb)) >= 1 and fb.value == 0
    o = orc.Oracle(10)
    o.scan_stream(s, 3)
    assert h.c.n_elem == o.c.n_elem
    h.finalise()
    o.finalise(1, 3)
    assert h.elems().tobytes() == o.elems().tobytes() and h.c.coverage == o.c.coverage
    h.delete()


@pytest.mark.parametrize("k,m", [(2, 1), (10, 3), (25, 4)])
def test_many_small_batches_grow_the_bucket_storage(k, m):
    """one counter, hundreds of small scans: the chunk pool and the chunk tables have to grow (and keep what is there)
    again and again; k = 2 piles everything into a few buckets"""
    rng = np.random.default_rng(11)
    c = tj.Counter(k)
    o = orc.Oracle(k)
    for i in range(150):
        n = int(rng.integers(50, 3000))
        s = tj.synth_stream(n, int(rng.integers(40, 200)), 30000, seed_reads=1000 + i)
        c.scan_host(s, m)
        o.scan_stream(s, m)
        if i % 37 == 0:
            assert c.raw_count() == o.c.n_elem          # synchronise now and then (tightens the bounds)
    big = tj.synth_stream(400000, 150, 30000, seed_reads=77)   # then one batch far larger than everything before
    c.scan_host(big, m)
    o.scan_stream(big, m)
    got = c.download_raw()
    exp = as_records(o.elems())
    assert len(got) == len(exp) and (rec_sorted(got) == rec_sorted(exp)).all()
    st = c.finalise(1, 3)
    o.finalise(1, 3)
    assert st == o.c.status == 0 and c.download_kept().tobytes() == o.elems().tobytes() and c.coverage == o.c.coverage
    # the counter is reusable after finalise
    c.scan_host(big, m)
    assert c.raw_count() == len(as_records(oracle_raw(big, k, m).elems()))
    c.close()


@pytest.mark.parametrize("k,n_samples,force_radix", [(15, 3, False), (15, 3, True), (10, 8, False), (25, 5, False), (32, 2, False), (2, 4, False)])
def test_merge_samples_device(monkeypatch, k, n_samples, force_radix):
    """cross-sample merge on the GPU == the oracle's restatement of the reference's merge order (oracle/context_oracle.c;
    src/genome_set.c:250-289, ties :278-281) collapsed into a union: keys, key order, the whole meta word (canon flag of the
    first sample that has the key, count = total) and the counts per sample; the bin path and (forced through the test
    hook) the radix path"""
    torch = pytest.importorskip("torch")
    from tatajuba_amd.dist import merge_histograms_device, device_bytes_tensor
    parts, counts, counters = [], [], []
    for smp in range(n_samples):
        s = tj.synth_stream(30000, 150, 200000, seed_reads=0x7A7A1000 + smp, variant_seed=smp)
        c = tj.Counter(k)
        c.scan_host(s, 4 if k > 2 else 1)
        assert c.finalise(smp % 2, 0) == 0                 # (every other sample keeps one-strand tracts: canon flags 1 and 2 in the union)
        counters.append(c)
        counts.append(c.n_kept)
        parts.append(device_bytes_tensor(c.kept_device_ptr, c.n_kept * 24, torch.device("cuda", 0)).clone())
    rec = torch.cat(parts)
    if force_radix:
        monkeypatch.setenv("TATAJUBA_AMD_BIN_MAX", "1")
    merger = tj.Counter(k)
    keys_d, mat_d = merge_histograms_device(merger, rec, counts)
    _, _, keys_o, mat_o = orc.merge_samples(np.frombuffer(rec.cpu().numpy().tobytes(), dtype=np.uint64).reshape(-1, 3), counts)
    kd = np.frombuffer(keys_d.cpu().numpy().tobytes(), dtype=np.uint64).reshape(-1, 3)
    assert len(kd) == len(keys_o) > 0
    assert (kd == keys_o).all()
    assert (mat_d.cpu().numpy() == mat_o).all()
    assert (tj.decode_meta(kd[:, 2])["count"] == mat_o.sum(axis=1)).all()
    # twice through the same merger (its buffers and counters are reused)
    keys_2, mat_2 = merge_histograms_device(merger, rec, counts)
    assert torch.equal(keys_2, keys_d) and torch.equal(mat_2, mat_d)
    merger.close()
    for c in counters:
        c.close()


def test_c_example_program_on_the_reference_fixture(tmp_path, golden_dir, known_answers):
    # examples/count_tracts.c: a C caller of the drop-in API, compiled with gcc against include/ and the shared library
    import subprocess
    from tests.test_cabi import _build_c_example
    exe = _build_c_example(tmp_path)
    path = os.path.join(golden_dir, known_answers["file"]["path"])
    r = subprocess.run([exe, path, "-k", "10", "-m", "3", "-c", "5", "-b", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    cs = [c for c in known_answers["file"]["cases"] if (c["k"], c["m"], c["min_coverage"], c["remove_biased"]) == (10, 3, 5, 1)][0]
    assert f": {cs['raw']} homopolymer tracts" in r.stdout
    assert f"context histogram: {cs['n_elem']} (context, length) entries, {cs['n_idx']} contexts reach coverage 5, coverage estimate {cs['coverage']}" in r.stdout
    # paired: the same file as R1 and R2
    r = subprocess.run([exe, path, path, "-k", "10", "-m", "3"], capture_output=True, text=True, timeout=120)
    p2 = known_answers["file"]["paired_same_file_twice_k10_m3"]
    assert r.returncode == 0 and f": {p2['raw']} homopolymer tracts" in r.stdout and f"context histogram: {p2['n_elem']} " in r.stdout


# ---- full benchmark size: size-independent properties ---------------------------------------------------------------

def _revcomp_stream(s, L):
    m = s.reshape(-1, L + 1)
    lut = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        lut[a] = b
    r = np.empty_like(m)
    r[:, :L] = lut[m[:, L - 1::-1]] if L > 0 else m[:, :0]
    r[:, L] = 10
    return r.reshape(-1)


def test_full_size_properties_config2():
    """BASELINE config 2 (10 M x 150 bp, k=10 m=3): properties that need no CPU pass at full size."""
    n_reads, L, k, m = 10_000_000, 150, 10, 3
    s = tj.synth_stream(n_reads, L, 5_000_000, n_threads=16)
    c = tj.Counter(k)
    c.scan_host(s, m)
    raw = c.raw_count()
    assert 5.8 < raw / n_reads < 6.15                          # r-bar ~ 5.97
    assert c.finalise(1, 5) == 0
    kept = c.download_kept()
    d = tj.decode_meta(kept["meta"])
    key = np.stack([d["base"].astype(np.uint64), kept["ctx0"], kept["ctx1"], d["length"].astype(np.uint64)], 1)
    # sortedness: strictly decreasing (base, ctx0, ctx1, length)
    a, b = key[:-1], key[1:]
    gt = np.zeros(len(a), bool)
    eq = np.ones(len(a), bool)
    for j in range(4):
        gt |= eq & (a[:, j] > b[:, j])
        eq &= a[:, j] == b[:, j]
    assert gt.all()
    assert (d["canon_flag"] == 3).all() and (d["count"] >= 2).all() and int(d["count"].sum()) <= raw
    gi, gf = c.download_idx()
    assert (gi < gf).all() and (gf[:-1] <= gi[1:]).all() and gf[-1] <= len(kept)
    cov = c.coverage
    # a 1/16 sample agrees with the oracle exactly
    sub = s[: (n_reads // 16) * (L + 1)]
    check_finalise([sub], k, m, 1, 5)
    # linearity: scanning the stream twice doubles every count and keeps the keys
    c.reset()
    c.scan_host(s, m); c.scan_host(s, m)
    assert c.raw_count() == 2 * raw and c.finalise(1, 5) == 0
    k2 = c.download_kept()
    d2 = tj.decode_meta(k2["meta"])
    assert (k2["ctx0"] == kept["ctx0"]).all() and (k2["ctx1"] == kept["ctx1"]).all() and (d2["count"] == 2 * d["count"]).all()
    assert c.coverage == 2 * cov
    # strand symmetry: reverse-complementing every read gives the same histogram
    c.reset()
    c.scan_host(_revcomp_stream(s, L), m)
    assert c.raw_count() == raw and c.finalise(1, 5) == 0
    assert c.download_kept().tobytes() == kept.tobytes() and c.coverage == cov
    # repeatability: the same resident stream scanned 150 times gives the same raw count every time (a race between the
    # waves of a workgroup over the staged-record count once lost some eighty records in one scan out of fifty)
    _repeat_scans(c, s, m, raw, 150)
    c.close()


def _repeat_scans(c, s, m, raw, n):
    """The same resident stream scanned n times: the same raw count AND the same count in every hash bucket each time.  (How
    round 3's lost-record race showed: one scan in fifty came out some eighty records short, spread over sixty buckets.)"""
    import torch
    dev = torch.from_numpy(s).cuda()
    c.reset()
    c.scan_device(dev.data_ptr(), s.size, m)
    assert c.raw_count() == raw
    want = c.bucket_counts()
    assert int(want.sum()) == raw
    bad = []
    for it in range(n):
        c.reset()
        c.scan_device(dev.data_ptr(), s.size, m)
        got = c.bucket_counts()
        if c.raw_count() != raw or (got != want).any():
            bad.append((it, c.raw_count() - raw, int((got != want).sum())))
    del dev
    assert not bad, bad


def _sorted_desc(kept, d):
    key = np.stack([d["base"].astype(np.uint64), kept["ctx0"], kept["ctx1"], d["length"].astype(np.uint64)], 1)
    a, b = key[:-1], key[1:]
    gt = np.zeros(len(a), bool)
    eq = np.ones(len(a), bool)
    for j in range(4):
        gt |= eq & (a[:, j] > b[:, j])
        eq &= a[:, j] == b[:, j]
    return bool(gt.all())


def test_full_size_properties_config3():
    """BASELINE config 3's parameters (150 bp reads, 50 Mb genome, k=15 m=4, strand-bias filter) on a 3 GB stream: the scan
    runs in the library's real 1 GiB pieces (three launches, cuts on read delimiters), two-word records, and the result
    has the properties that need no CPU pass at this size; a 1/16 sample agrees with the oracle exactly."""
    # (TATAJUBA_AMD_C3_READS=100000000 runs it at BASELINE's full size -- 15 GB, fourteen pieces, some 50 GB of host memory and
    # a few minutes: done once per round by hand, DESIGN section 6 says with what outcome)
    n_reads, L, k, m = int(os.environ.get("TATAJUBA_AMD_C3_READS", "20000000")), 150, 15, 4
    s = tj.synth_stream(n_reads, L, 50_000_000, n_threads=16)
    assert s.size > (3 << 30) // 2                              # more than 1.5 GiB: scanned piece by piece
    c = tj.Counter(k)
    c.scan_host(s, m)
    assert c.last_scan_launches() >= max(3, s.size >> 30)
    print(f"config-3 property test: {n_reads} reads, {s.size >> 20} MiB, {c.last_scan_launches()} scan launches")
    raw = c.raw_count()
    assert 1.25 < raw / n_reads < 1.5                           # r-bar ~ 1.37
    assert c.finalise(1, 5) == 0
    kept = c.download_kept()
    d = tj.decode_meta(kept["meta"])
    assert _sorted_desc(kept, d)
    assert (d["canon_flag"] == 3).all() and (d["count"] >= 2).all() and int(d["count"].sum()) <= raw
    gi, gf = c.download_idx()
    assert (gi < gf).all() and (gf[:-1] <= gi[1:]).all() and gf[-1] <= len(kept)
    cov = c.coverage
    check_finalise([s[: (n_reads // 16) * (L + 1)]], k, m, 1, 5)
    # strand symmetry: reverse-complementing every read gives the same histogram
    c.reset()
    c.scan_host(_revcomp_stream(s, L), m)
    assert c.raw_count() == raw and c.finalise(1, 5) == 0
    assert c.download_kept().tobytes() == kept.tobytes() and c.coverage == cov
    # repeatability of the two-word kernel's passes (k > 12: the scan kernel partitions its 1536-record passes itself), pieces included
    _repeat_scans(c, s, m, raw, 30)
    c.close()


def test_full_size_properties_config5():
    """BASELINE config 5's parameters (reads of 2-20 kb, 100 Mb genome, k=25 m=4) on 1.1 GB of ragged long reads: raw
    record density, order, filter and index properties; a 1/16 sample agrees with the oracle exactly; linearity."""
    n_reads, k, m = 100_000, 25, 4
    s = tj.synth_stream(n_reads, 2000, 100_000_000, read_len_max=20000, n_threads=16)
    assert s.size > 1_000_000_000
    c = tj.Counter(k)
    c.scan_host(s, m)
    raw = c.raw_count()
    assert 0.0105 < raw / s.size < 0.0125                       # 0.0117 tracts per base
    st = c.finalise(1, 2)
    assert st == 0
    kept = c.download_kept()
    d = tj.decode_meta(kept["meta"])
    assert _sorted_desc(kept, d)
    assert (d["canon_flag"] == 3).all() and (d["count"] >= 2).all() and int(d["count"].sum()) <= raw
    gi, gf = c.download_idx()
    assert (gi < gf).all() and (gf[:-1] <= gi[1:]).all() and gf[-1] <= len(kept)
    cov = c.coverage
    ends = np.flatnonzero(s[: s.size // 16] == 10)
    check_finalise([s[: ends[-1] + 1]], k, m, 1, 2)
    c.reset()
    c.scan_host(s, m); c.scan_host(s, m)
    assert c.raw_count() == 2 * raw and c.finalise(1, 2) == 0
    k2 = c.download_kept()
    d2 = tj.decode_meta(k2["meta"])
    assert (k2["ctx0"] == kept["ctx0"]).all() and (k2["ctx1"] == kept["ctx1"]).all() and (d2["count"] == 2 * d["count"]).all()
    assert c.coverage == 2 * cov
    _repeat_scans(c, s, m, raw, 30)                             # (long reads, k = 25: the two-word kernel again, tracts walked past the window's end)
    c.close()


# ---- "next" rows: context grouping (N3), tract ids and the in-process exchange (N1), batched window rescans (N4) ------

@pytest.mark.parametrize("k,maxd", [(10, 1), (10, 2), (10, 3), (25, 2)])
def test_group_contexts_matches_the_oracle(k, maxd):
    """tjamd_group_contexts == the oracle's restatement of the reference's greedy grouping (src/context_histogram.c:245-270,
    distance :25-48) on a finalised sample that holds families of contexts 0-3 substitutions apart"""
    rng = random.Random(100 * k + maxd)
    mask = (1 << (2 * k)) - 1
    e = []
    for fam in range(3000):
        c0, c1, base = rng.getrandbits(2 * k) & mask, rng.getrandbits(2 * k) & mask, rng.randrange(2)
        for member in range(rng.choice([1, 1, 2, 3, 5])):
            a, b = c0, c1
            for _ in range(rng.choice([0, 1, 1, 2, 3])):
                if rng.random() < 0.5:
                    a ^= rng.randrange(1, 4) << (2 * rng.randrange(k))
                else:
                    b ^= rng.randrange(1, 4) << (2 * rng.randrange(k))
            for length in rng.sample(range(3, 12), rng.choice([1, 2, 3])):
                e += [(a, b, base, length)] * rng.randrange(2, 6)
    raw = np.zeros(len(e), dtype=tj.ELEM_DTYPE)
    for i, (a, b, base, length) in enumerate(e):
        raw["ctx0"][i], raw["ctx1"][i] = a, b
        raw["meta"][i] = base | (length << 2) | (1 << 12) | (0xffe << 32) | (1 << 49)
    raw["read_offset"] = 0; raw["loc_ref_id"] = raw["loc_pos"] = raw["loc_last"] = -1
    c = tj.Counter(k)
    c.upload_raw(raw)
    assert c.finalise(0, 0) == 0
    kept = c.download_kept()
    gof, grp = c.group_contexts(maxd)
    ogof, first, nel, nctx, integ, mode = orc.group_contexts(kept, maxd)
    assert len(grp) == len(first) and (gof == ogof).all()
    assert (grp["first"] == first).all() and (grp["n_elem"] == nel).all() and (grp["n_context"] == nctx).all()
    assert (grp["integral"] == integ).all() and (grp["mode"] == mode).all()
    if maxd > 1:
        assert len(grp) < len(np.unique(np.stack([kept["ctx0"], kept["ctx1"]], 1), axis=0))     # something was grouped
    c.close()


def _hist_mask(groups, n):
    m = np.zeros(n, bool)
    for f, nl in zip(groups["first"], groups["n_len"]):
        m[f:f + nl] = True
    return m


@pytest.mark.parametrize("k,maxd,lev,free_end", [(10, 1, 2, 0), (10, 1, 3, 0), (10, 2, 3, 0), (10, 0, 2, 0), (10, 2, 0, 0), (25, 2, 4, 0), (32, 1, 3, 0), (4, 1, 2, 0),
                                                  (10, 1, 2, 1), (10, 2, 3, 1), (25, 2, 4, 1)])
def test_context_histograms_with_indel_retry_match_the_oracle(k, maxd, lev, free_end, monkeypatch):
    """tjamd_context_histograms == the oracle's restatement of new_genomic_context_list (src/context_histogram.c:245-270: flank
    distance :25-48, then the retry with the edit distance between the names :19-23,255-261, bookkeeping :181-222, length
    histograms :278-286) on a sample whose families differ by substitutions AND by one-base indels in the right flank (the
    left flank's first bases vary as well: in context order those stay neighbours).  UNPINNED pieces, same on both sides:
    the edit distance (biomcmc_levenshtein_distance is absent from the reference tree: unit-cost global edit distance, or --
    free_end, TATAJUBA_AMD_EDIT_DISTANCE=free_end -- the other reading of its last argument) and the order of equal counts
    in a length histogram."""
    if free_end:
        monkeypatch.setenv("TATAJUBA_AMD_EDIT_DISTANCE", "free_end")
    else:
        monkeypatch.delenv("TATAJUBA_AMD_EDIT_DISTANCE", raising=False)
    orc.set_edit_free_end(free_end)
    rng = random.Random(1000 * k + 10 * maxd + lev)
    mask = (1 << (2 * k)) - 1
    e = []
    for fam in range(2500):
        c0, c1, base = rng.getrandbits(2 * k) & mask, rng.getrandbits(2 * k) & mask, rng.randrange(2)
        for member in range(rng.choice([1, 1, 2, 3, 5, 8])):
            a, b = c0, c1
            for _ in range(rng.choice([0, 1, 1, 2, 3])):
                r = rng.random()
                if r < 0.25:
                    a ^= rng.randrange(1, 4) << (2 * rng.randrange(min(k, 2)))        # first bases of the left flank
                elif r < 0.5:
                    b ^= rng.randrange(1, 4) << (2 * rng.randrange(k))                # substitution in the right flank
                elif r < 0.75:
                    p = rng.randrange(k)                                              # insertion in the right flank at base p
                    lo = b & ((1 << (2 * p)) - 1)
                    b = (lo | (rng.randrange(4) << (2 * p)) | ((b >> (2 * p)) << (2 * p + 2))) & mask
                else:
                    p = rng.randrange(k)                                              # deletion at base p, a new last base
                    lo = b & ((1 << (2 * p)) - 1)
                    b = (lo | ((b >> (2 * p + 2)) << (2 * p)) | (rng.randrange(4) << (2 * k - 2))) & mask
            for length in rng.sample(range(3, 12), rng.choice([1, 2, 3])):
                e += [(a, b, base, length)] * rng.randrange(2, 9)
    raw = np.zeros(len(e), dtype=tj.ELEM_DTYPE)
    for i, (a, b, base, length) in enumerate(e):
        raw["ctx0"][i], raw["ctx1"][i] = a, b
        raw["meta"][i] = base | (length << 2) | (1 << 12) | (0xffe << 32) | (1 << 49)
    raw["read_offset"] = 0; raw["loc_ref_id"] = raw["loc_pos"] = raw["loc_last"] = -1
    c = tj.Counter(k)
    c.upload_raw(raw)
    assert c.finalise(0, 0) == 0
    kept = c.download_kept()
    got = c.context_histograms(maxd, lev)
    want = orc.genomic_context_list(kept, k, maxd, lev, 3)
    if free_end:                                                                      # the second reading takes in more than the first
        orc.set_edit_free_end(0)
        assert (want["join_type"] == 2).sum() > (orc.genomic_context_list(kept, k, maxd, lev, 3)["join_type"] == 2).sum()
    g, w = got["groups"], want["groups"]
    assert len(g) == len(w) and (got["group_of"] == want["group_of"]).all() and (got["join_type"] == want["join_type"]).all()
    for f in ("first", "n_elem", "n_context", "mode", "indel", "n_len", "modal_len", "modal_freq", "integral"):
        assert (g[f] == w[f]).all(), f
    m = _hist_mask(w, len(kept))
    assert (got["hist"]["length"][m] == want["hist_len"][m]).all() and (got["hist"]["freq"][m] == want["hist_freq"][m]).all()
    # what the modal element says about its histogram (reference: src/context_histogram.c:192-196)
    meta = tj.decode_meta(kept["meta"])
    assert (meta["count"][g["mode"]] == w["mode_context_count"]).all() and (meta["length"][g["mode"]] == w["mode_context_length"]).all()
    if lev > 1 and maxd > 0:
        assert (want["join_type"] == 2).sum() > 0 and g["indel"].sum() > 0            # the retry took elements in ...
        ham_only = c.group_contexts(maxd)[1]
        assert len(g) < len(ham_only)                                                 # ... that the flank distance alone leaves out
    if lev == 0:
        gof0, grp0 = c.group_contexts(maxd)                                           # without a retry: the round-2 entry's groups
        assert (gof0 == got["group_of"]).all() and (grp0["n_context"] == g["n_context"]).all() and g["indel"].sum() == 0
    c.close()


@pytest.mark.parametrize("k,maxd,lev", [(10, 1, 2), (8, 2, 3)])
def test_new_genomic_context_list_drop_in(tmp_path, k, maxd, lev):
    """new_genomic_context_list (include/tatajuba_context.h: finalise, then the reference's grouping loop with distances from the
    device) on a FASTQ file of error-laden reads (substitutions and one-base indels, deep enough for the errors to recur) ==
    the oracle's restatement of src/context_histogram.c:224-286, struct field by struct field: contexts in the order they
    were added, modal context, name, indel flag, integral, the length histogram `h`."""
    from tatajuba_amd.capi import Options
    rng = random.Random(5 * k + lev)
    genome = bytearray(rng.choice(b"ACGT") for _ in range(6000))
    for _ in range(150):                                    # plant tracts
        p, ln, b = rng.randrange(50, 5900), rng.randrange(3, 9), rng.choice(b"ACGT")
        genome[p:p + ln] = bytes([b]) * ln
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for _ in range(40000):
        p = rng.randrange(0, len(genome) - 100)
        r = bytearray(genome[p:p + 100])
        for _ in range(rng.choice([0, 0, 1, 1, 2])):        # sequencing errors
            q, what = rng.randrange(len(r)), rng.random()
            if what < 0.6:
                r[q] = rng.choice(b"ACGT")
            elif what < 0.8:
                del r[q]
            else:
                r.insert(q, rng.choice(b"ACGT"))
        r = bytes(r)
        reads.append(r if rng.random() < 0.5 else r.translate(comp)[::-1])
    fq = str(tmp_path / "reads.fq")
    with open(fq, "wb") as fh:
        fh.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads)))
    m = 3
    opt = Options.defaults(k, m, 2, False)
    opt.max_distance_per_flank, opt.levenshtein_distance = maxd, lev
    L = tj.lib()
    h = tj.HopoCounter.new_or_append_from_file(None, fq, opt)
    gl = L.new_genomic_context_list(h._p)
    assert gl, "sample excluded"
    g = gl.contents
    # the oracle on the same reads
    o = orc.Oracle(k)
    o.scan_stream(np.frombuffer(b"".join(r + b"\n" for r in reads), np.uint8), m)
    o.finalise(0, 2)
    assert o.c.status == 0 and h.c.n_elem == o.c.n_elem and g.coverage == o.c.coverage
    e = o.elems()
    w = orc.genomic_context_list(e, k, maxd, lev, m, o.c.coverage)
    wg = w["groups"]
    assert g.n_hist == len(wg) and g.ref_start == 0 and g.name == os.fsencode(fq)
    meta = orc.decode_meta(e["meta"])
    n_indel = 0
    for i in range(g.n_hist):
        ch, x = g.hist[i].contents, wg[i]
        f = int(x["first"])
        assert (ch.n_context, ch.integral, ch.mode_context_count, ch.mode_context_length, ch.mode_context_id) == \
            (x["n_context"], x["integral"], x["mode_context_count"], x["mode_context_length"], x["mode_context_id"]), i
        assert (ch.indel != 0) == bool(x["indel"]) and ch.base == meta["base"][f] and ch.location == -1 and ch.tract_id == -1
        assert (ch.coverage, ch.n_tracts) == (x["coverage"], x["n_tracts"]) == (o.c.coverage, len(wg))   # src/context_histogram.c:302
        n_indel += int(x["indel"])
        assert [ch.context[t] for t in range(2 * ch.n_context)] == w["contexts"][f:f + ch.n_context].reshape(-1).tolist(), i
        mo = int(x["mode"])
        assert ch.name.decode() == orc.name_of(e["ctx0"][mo], e["ctx1"][mo], meta["base"][mo], k), i
        hh = ch.h.contents
        assert hh.n == x["n_len"] and [(hh.i[t].idx, hh.i[t].freq) for t in range(hh.n)] == \
            list(zip(w["hist_len"][f:f + hh.n].tolist(), w["hist_freq"][f:f + hh.n].tolist())), i
        assert hh.min == min(w["hist_len"][f:f + hh.n]) and hh.max == max(w["hist_len"][f:f + hh.n])
    assert n_indel > 0 and g.n_hist < len(e)                 # the retry and the grouping both did something
    L.del_genomic_context_list(gl)
    h.delete()


def test_context_list_refuses_elements_reordered_by_an_aligner_hook(tmp_path, golden_dir):
    """A host program that links find_reference_location_and_sort_hopo_counter (the weak hook at the place of the
    reference's BWA step, src/hopo_counter.c:416) gets hc->elem re-ordered and ref_start > 0: new_genomic_context_list must
    stop with a message, not apply the device's grouping to the re-ordered array (tests/c/hook_guard.c).  With a hook that
    leaves the array alone the list is built."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "tatajuba_amd")
    exe = str(tmp_path / "hook_guard")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "hook_guard.c"),
                           "-L", libdir, "-ltatajuba_amd", "-Wl,-rpath," + libdir, "-o", exe])
    fq = os.path.join(golden_dir, "err1750956.fastq.gz")
    r = subprocess.run([exe, fq, "10", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "re-ordered after finalise_hopo_counter" in r.stderr and "histograms" not in r.stdout
    r = subprocess.run([exe, fq, "10", "3", "keep"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "hook calls 1, histograms " in r.stdout and "histograms -1" not in r.stdout


def test_tract_ids_and_in_process_gather():
    """tjamd_gather_histograms (the exchange for samples that are threads of one process, as in the reference) followed by
    tjamd_merge_samples == the oracle's merge (src/genome_set.c:250-289 restated); tjamd_tract_ids == the oracle's id pass over the union"""
    torch = pytest.importorskip("torch")
    import ctypes as C
    from tatajuba_amd.dist import device_bytes_tensor
    k, counters = 10, []
    for smp in range(3):
        s = tj.synth_stream(30000, 150, 200000, seed_reads=0x7A7A1000 + smp, variant_seed=smp)
        c = tj.Counter(k)
        c.scan_host(s, 3)
        assert c.finalise(1, 0) == 0
        counters.append(c)
    L = tj.lib()
    hs = (C.c_void_p * 3)(*[c._h for c in counters])
    drec, counts = C.c_void_p(), (C.c_long * 3)()
    total = L.tjamd_gather_histograms(counters[0]._h, hs, 3, C.byref(drec), counts)
    assert total == sum(c.n_kept for c in counters) and list(counts) == [c.n_kept for c in counters]
    rec = device_bytes_tensor(drec.value, total * 24, torch.device("cuda", 0)).cpu().numpy()
    parts = [device_bytes_tensor(c.kept_device_ptr, c.n_kept * 24, torch.device("cuda", 0)).cpu().numpy() for c in counters]
    assert rec.tobytes() == b"".join(p.tobytes() for p in parts)
    keys = torch.empty(total * 24, dtype=torch.uint8, device="cuda")
    mat = torch.empty((total, 3), dtype=torch.int32, device="cuda")
    nu = L.tjamd_merge_samples(counters[0]._h, drec, counts, 3, C.c_void_p(keys.data_ptr()), C.c_void_p(mat.data_ptr()), total)
    _, _, keys_o, mat_o = orc.merge_samples(np.frombuffer(rec.tobytes(), dtype=np.uint64).reshape(-1, 3), list(counts))
    assert nu == len(keys_o) and (mat[:nu].cpu().numpy() == mat_o).all()
    assert (np.frombuffer(keys[: nu * 24].cpu().numpy().tobytes(), dtype=np.uint64).reshape(-1, 3) == keys_o).all()
    ids = np.zeros(nu, np.int32)
    nid = L.tjamd_tract_ids(counters[0]._h, C.c_void_p(keys.data_ptr()), nu, None, ids.ctypes.data)
    kd = np.frombuffer(keys[: nu * 24].cpu().numpy().tobytes(), dtype=np.uint64).reshape(-1, 3)
    oids, onid = orc.tract_ids(kd)
    assert nid == onid and (ids == oids).all() and 0 < nid < nu
    for c in counters:
        c.close()


def test_config4_pipeline_eight_samples_end_to_end():
    """BASELINE.json configs[3] as ONE pipeline at reduced size (8 samples of the same genome with per-sample tract-length
    variants, 150 bp reads of both strands, k = 15, min_tract = 4, remove_biased = 1): scan -> finalise on eight counters of
    one GPU -> tjamd_gather_histograms -> tjamd_merge_samples -> tjamd_tract_ids, every stage against the oracle
    (reference: src/genome_set.c:66-94 per-sample loop, :195-229 and :250-289 merge, :207-221 tract ids)."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    from tatajuba_amd.dist import device_bytes_tensor
    k, m, ns = 15, 4, 8
    counters, okept = [], []
    for smp in range(ns):
        s = tj.synth_stream(150000, 150, 1000000, seed_reads=0x7A7A1000 + smp, variant_seed=smp)
        c = tj.Counter(k)
        c.scan_host(s, m)
        o = orc.Oracle(k)
        o.scan_stream(s, m)
        assert c.raw_count() == o.c.n_elem
        st = c.finalise(1, 5)
        o.finalise(1, 5)
        assert st == o.c.status == 0
        assert c.download_kept().tobytes() == o.elems().tobytes()
        gi, gf = c.download_idx()
        ei, ef = o.idx()
        assert (gi == ei).all() and (gf == ef).all() and c.coverage == o.c.coverage
        counters.append(c)
        okept.append(as_records(o.elems()))
        o.close()
    L = tj.lib()
    hs = (C.c_void_p * ns)(*[c._h for c in counters])
    drec, counts = C.c_void_p(), (C.c_long * ns)()
    merger = tj.Counter(k)
    total = L.tjamd_gather_histograms(merger._h, hs, ns, C.byref(drec), counts)
    assert total == sum(len(x) for x in okept) and list(counts) == [len(x) for x in okept]
    rec = device_bytes_tensor(drec.value, total * 24, torch.device("cuda", 0)).cpu().numpy()
    want = np.concatenate(okept)
    assert rec.tobytes() == want.tobytes()
    keys = torch.empty(total * 24, dtype=torch.uint8, device="cuda")
    mat = torch.empty((total, ns), dtype=torch.int32, device="cuda")
    nu = L.tjamd_merge_samples(merger._h, drec, counts, ns, C.c_void_p(keys.data_ptr()), C.c_void_p(mat.data_ptr()), total)
    _, _, keys_o, mat_o = orc.merge_samples(np.frombuffer(want.tobytes(), dtype=np.uint64).reshape(-1, 3), list(counts))
    kd = np.frombuffer(keys[: nu * 24].cpu().numpy().tobytes(), dtype=np.uint64).reshape(-1, 3)
    assert nu == len(keys_o) and (kd == keys_o).all() and (mat[:nu].cpu().numpy() == mat_o).all()
    ids = np.zeros(nu, np.int32)
    nid = L.tjamd_tract_ids(merger._h, C.c_void_p(keys.data_ptr()), nu, None, ids.ctypes.data)
    oids, onid = orc.tract_ids(kd)
    assert nid == onid and (ids == oids).all() and 0 < nid < nu
    # the samples differ (variants) and share most tracts
    share = (mat_o > 0).sum(axis=1)
    assert share.max() == ns and share.min() >= 1 and (share < ns).any()
    merger.close()
    for c in counters:
        c.close()


def test_rccl_allgather_of_one_rank_and_block_size_protocol():
    """tjamd_allgather_histograms (ncclAllGather behind the C ABI) on a communicator of one rank -- all a one-GPU box can run:
    the gathered buffer is the counter's kept records, the first exchange learns the counts first (two collectives), the
    next one sends one block, a sample that outgrows the agreed block size is exchanged again at the right size.  The
    multi-rank path is the same code with world > 1; it has not run on more than one device."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    from tatajuba_amd.dist import device_bytes_tensor
    L = tj.lib()
    k = 10
    small = tj.Counter(k)
    small.scan_host(tj.synth_stream(20000, 150, 100000, seed_reads=0x7A7A1000), 3)
    assert small.finalise(1, 0) == 0
    big = tj.Counter(k)
    big.scan_host(tj.synth_stream(400000, 150, 2000000, seed_reads=0x7A7A1001, variant_seed=1), 3)
    assert big.finalise(0, 0) == 0 and big.n_kept > 2 * small.n_kept + 8192
    ident = C.create_string_buffer(128)
    assert L.tjamd_comm_unique_id(ident) == 0
    comm = L.tjamd_comm_create(small._h, ident, 0, 1)
    assert comm, L.tjamd_last_error()
    assert L.tjamd_comm_rank(comm) == 0 and L.tjamd_comm_world(comm) == 1
    L.tjamd_comm_count.restype = C.c_int; L.tjamd_comm_count.argtypes = [C.c_void_p]
    assert L.tjamd_comm_count(comm) == 1                       # what RCCL itself says (ncclCommCount)
    L.tjamd_comm_last_exchange.restype = C.c_int
    L.tjamd_comm_last_exchange.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    ms, nb, nc = C.c_double(), C.c_long(), C.c_long()
    assert L.tjamd_comm_last_exchange(comm, C.byref(ms), C.byref(nb), C.byref(nc)) != 0      # nothing exchanged yet

    def last():
        assert L.tjamd_comm_last_exchange(comm, C.byref(ms), C.byref(nb), C.byref(nc)) == 0
        return ms.value, nb.value, nc.value

    def exchange(c):
        ptr, cnt = C.c_void_p(), (C.c_long * 1)()
        tot = L.tjamd_allgather_histograms(c._h, comm, C.byref(ptr), cnt)
        assert tot == c.n_kept == cnt[0], L.tjamd_last_error()
        got = device_bytes_tensor(ptr.value, tot * 24, torch.device("cuda", 0)).cpu().numpy()
        assert got.tobytes() == as_records(c.download_kept()).tobytes()

    exchange(small)
    assert L.tjamd_comm_collectives(comm) == 2                 # counts, then the block
    t, b, n = last()
    assert n == 2 and 0.0 < t < 1000.0 and b >= small.n_kept * 24 + 16        # (what bench.py --gpus N reports per exchange)
    exchange(small)
    assert L.tjamd_comm_collectives(comm) == 3                 # the block size is agreed: one collective
    assert last()[2] == 1
    exchange(big)                                              # does not fit the agreed block: one wasted, then counts + block
    assert L.tjamd_comm_collectives(comm) == 6
    assert last()[2] == 3 and last()[1] >= big.n_kept * 24
    exchange(big)
    assert L.tjamd_comm_collectives(comm) == 7
    L.tjamd_comm_destroy(comm)
    small.close(); big.close()


def test_exchange_pack_and_unpack_for_several_ranks_on_one_gpu():
    """What tjamd_allgather_histograms does around its collective, for a communicator of five ranks, on one GPU: every
    rank's kept records packed into a max-padded block (count in the header), the blocks unpacked back to back in rank
    order -- offsets of the ranks r > 0 included, an empty rank, and a rank that holds more than the agreed block (its
    count comes back whole, its records cut at the block: the caller's cue to settle on a larger block and go again)."""
    import ctypes as C
    L = tj.lib()
    rng = np.random.default_rng(17)
    cap = 4096
    ns = [1500, 0, 4096, 6000, 7]
    samples = [rng.integers(0, 1 << 63, size=(n, 3), dtype=np.uint64) for n in ns]
    c = tj.Counter(10)
    ptrs = (C.c_void_p * len(ns))(*[s.ctypes.data for s in samples])
    n_arr = (C.c_long * len(ns))(*ns)
    out = np.zeros((cap * len(ns), 3), np.uint64)
    counts = (C.c_long * len(ns))()
    L.tjamd_debug_exchange_pack_unpack.restype = C.c_long
    L.tjamd_debug_exchange_pack_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_void_p]
    tot = L.tjamd_debug_exchange_pack_unpack(c._h, ptrs, n_arr, len(ns), cap, out.ctypes.data, counts)
    assert tot == sum(min(n, cap) for n in ns), L.tjamd_last_error()
    assert list(counts) == ns                                  # rank 3 says 6000 > cap: the exchange would be repeated with a larger block
    want = np.concatenate([s[:cap] for s in samples])
    assert (out[:tot] == want).all()
    c.close()


def test_merge_samples_c_example(tmp_path):
    """examples/merge_samples.c: two samples, two counters, gather + merge + tract ids from plain C, no Python in the loop"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, libdir = str(tmp_path / "merge_samples"), os.path.join(root, "tatajuba_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "merge_samples.c"),
                           "-L", libdir, "-ltatajuba_amd", "-Wl,-rpath," + libdir, "-o", exe])
    files, kept = [], []
    for smp in range(2):
        s = tj.synth_stream(20000, 150, 100000, seed_reads=0x7A7A1000 + smp, variant_seed=smp)
        reads = bytes(s).split(b"\n")[:-1]
        f = str(tmp_path / f"s{smp}.fq")
        with open(f, "wb") as fh:
            fh.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(reads)))
        files.append(f)
        c = tj.Counter(10)
        c.scan_host(s, 3)
        assert c.finalise(1, 5) == 0
        kept.append(c.download_kept())
        c.close()
    r = subprocess.run([exe, "-k", "10", "-m", "3", "-c", "5"] + files, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    keysets = [set(zip(tj.decode_meta(x["meta"])["base"].tolist(), x["ctx0"].tolist(), x["ctx1"].tolist(), tj.decode_meta(x["meta"])["length"].tolist())) for x in kept]
    union, both = keysets[0] | keysets[1], keysets[0] & keysets[1]
    contexts = {(b, a0, a1) for (b, a0, a1, _) in union}
    assert f"merged: {len(kept[0]) + len(kept[1])} bars gathered, {len(union)} in the union, {len(both)} seen in every sample, {len(contexts)} tract ids" in r.stdout


def test_scan_windows_batch_matches_single_calls_and_is_fast():
    """tjamd_scan_windows: the reference's per-window rescans (src/genome_set.c:525-577: one counter and one
    update_hopo_counter_from_seq per ~100-base window) as one launch; same records as the oracle window by window, m = 2 and
    the all-monomers mode; 10 000 windows well under 50 ms"""
    import ctypes as C
    import time
    rng = random.Random(3)
    k, nw = 8, 10000
    wins = ["".join(rng.choice("ACGT") for _ in range(rng.randrange(60, 140))) for _ in range(nw)]
    wins[5] = "ACGT"                                         # shorter than k: nothing
    wins[6] = ""
    arr = (C.c_char_p * nw)(*[w.encode() for w in wins])
    lens = (C.c_int * nw)(*[len(w) for w in wins])
    L = tj.lib()
    for m in (2, 0):
        cap = sum(len(w) for w in wins) + 16
        out = np.zeros(cap, dtype=tj.ELEM_DTYPE)
        wof = np.zeros(cap, dtype=np.int32)
        n = L.tjamd_scan_windows(k, arr, lens, nw, m, out.ctypes.data, wof.ctypes.data, cap)     # warm-up + result
        t = time.perf_counter()
        n2 = L.tjamd_scan_windows(k, arr, lens, nw, m, out.ctypes.data, wof.ctypes.data, cap)
        dt = time.perf_counter() - t
        assert n == n2 > 0 and dt < 0.05, dt
        exp, expw = [], []
        for i in list(range(0, 40)) + list(range(nw - 40, nw)):
            o = orc.Oracle(k)
            (o.scan_seq(wins[i], m) if m else o.scan_seq_all_monomers(wins[i]))
            exp.append(o.elems().copy()); expw += [i] * len(exp[-1])
            o.close()
        sel = np.isin(wof[:n], np.array(sorted(set(expw)) + [5, 6]))
        got = out[:n][sel]
        assert got.tobytes() == np.concatenate(exp).tobytes() and wof[:n][sel].tolist() == expw
        assert (np.diff(wof[:n]) >= 0).all()


@pytest.mark.gpu
def test_bench_line_keeps_its_contract_on_a_small_sample():
    """bench.py as the driver runs it (one GPU, a child process), on a sample small enough for a test: the one JSON line
    carries the contract's keys, the roofline and the CPU baseline objects, the partition stage of the default sink and the
    fused sink measured beside it -- on the same histogram"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {kk: vv for kk, vv in os.environ.items() if kk not in ("TATAJUBA_AMD_SINK", "TATAJUBA_AMD_FAST")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reads", "400000", "--steps", "3", "--warmup", "1", "--cpu-reads", "50000",
                        "--io-reads", "50000", "--gz-reads", "20000"], capture_output=True, text=True, cwd=root, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "reads/s" and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - 400000 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and abs(roof["achieved"] - roof["algorithmic_bytes"] / (roof["ms"] * 1e-3) / 1e9) < 1e-6 * roof["achieved"]
    # (the roofline object is the step's longest stage: on a sample this small the finalise's latency floor, on the headline's the scan)
    assert (roof["kernel"] == "scan_fast_kernel<1, true>" and roof["algorithmic_bytes"] == 400000 * 151) or roof["kernel"].startswith("finalise")
    assert d["stages"]["scan"]["kernel"] == "scan_fast_kernel<1, true>" and d["stages"]["scan"]["ms"] > 0 and d["stages"]["finalise"]["ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "reads/s"
    st = d["stages"]
    assert st["partition"]["kernel"] == "partition_log_kernel<1>" and st["partition"]["ms"] > 0
    assert st["partition"]["algorithmic_bytes"] == 16.0 * d["config"]["raw_records_per_gpu"]
    fs = st["fused_sink"]
    assert fs["kernel"] == "scan_fast_kernel<1, false>" and fs["reads_per_s"] > 0 and fs["scan_ms"] > 0       # (bench.py itself checks: same kept count)
    assert st["dropin"]["kept_records"] == d["config"]["kept_records"]
    for io in ("from_host", "from_file", "from_gzip", "from_bgzf"):
        assert st[io]["reads_per_s"] > 0, io
