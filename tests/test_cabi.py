"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol the two headers declare, struct
layouts match the reference ABI, the host-only pieces (tables, name strings, reader, generator) behave, and the
device entries fail loudly without a GPU.  No compute call is made here."""
import ctypes as C
import gzip
import os
import random
import re

import numpy as np
import pytest

import tatajuba_amd as tj
from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*\(", txt))
    names |= set(re.findall(r"extern\s+\w+\s+(\w+)\s*\[", txt))
    return {n for n in names if not n.startswith("__") and n not in ("defined", "sizeof")}


def test_library_exports_every_declared_symbol():
    L = tj.lib()
    declared = (_declared("tatajuba_hopo.h") | _declared("tatajuba_amd.h") | _declared("tatajuba_context.h"))
    declared.discard("find_reference_location_and_sort_hopo_counter")     # weak import, provided by the host program
    assert declared == set(tj.EXPORTS), declared ^ set(tj.EXPORTS)
    for name in tj.EXPORTS:
        assert hasattr(L, name) or C.c_char.in_dll(L, name) is not None, name


def test_struct_layouts_match_reference_abi():
    from tatajuba_amd.capi import Options, _HopoCounterStruct
    assert C.sizeof(Options) == 64 and C.sizeof(_HopoCounterStruct) == 136      # SURVEY 8a (probed on the reference)
    assert Options.gff.offset == 24 and Options.kmer_size.offset == 36 and Options.min_coverage.offset == 48
    assert _HopoCounterStruct.opt.offset == 64 and _HopoCounterStruct.ref_counter.offset == 128
    assert tj.ELEM_DTYPE.itemsize == 40 and tj.RECORD_DTYPE.itemsize == 24 and tj.LOCATED_DTYPE.itemsize == 32


def test_tables_match_reference_encoding():
    L = tj.lib()
    tab = np.ctypeslib.as_array((C.c_uint8 * 512).in_dll(L, "dna_in_2_bits")).reshape(256, 2)
    exp = np.full((256, 2), 4, np.uint8)
    for ch, f in zip("ACGTU", [0, 1, 2, 3, 3]):
        for c in (ch, ch.lower()):
            exp[ord(c)] = (f, 3 - f)
    assert (tab == exp).all()
    assert bytes((C.c_char * 4).in_dll(L, "bit_2_dna")) == b"ACGT"


def test_name_and_tract_strings():
    L = tj.lib()
    ctx = (C.c_uint64 * 2)(0x25, 0x32)
    libc = C.CDLL(None)
    p = L.generate_name_from_flanking_contexts(ctx, 0, 3, False)
    assert C.string_at(p) == b"CCG.A.GAT"; libc.free(C.c_void_p(p))
    p = L.generate_name_from_flanking_contexts(ctx, 0, 3, True)
    assert C.string_at(p) == b"ATC.T.CGG"; libc.free(C.c_void_p(p))
    p = L.generate_tract_as_string(ctx, 0, 3, 4, False)
    assert C.string_at(p) == b"CCGAAAAGAT"; libc.free(C.c_void_p(p))
    p = L.generate_tract_as_string(ctx, 0, 3, 4, True)
    assert C.string_at(p) == b"ATCTTTTCGG"; libc.free(C.c_void_p(p))


def test_new_and_del_counter_without_gpu():
    h = tj.HopoCounter.new(10)
    assert h.c.n_alloc == 32 and h.c.n_elem == 0 and h.c.kmer_size == 10 and h.c.ref_counter == 1
    assert not h.c.idx_initial and h.c.name is None
    h.c.opt = tj.Options.defaults(10, 3)
    h.finalise()                      # empty counter: warning + excluded, no device needed (reference :345-349)
    assert h.c.n_elem == 0 and h.c.ref_start == 0
    h.delete()


@pytest.mark.skipif(tj.device_count() > 0, reason="only meaningful on a box without a GPU")
def test_device_entries_fail_loudly_without_gpu():
    with pytest.raises(tj.TatajubaAmdError, match="no HIP device"):
        tj.Counter(10)


def _write(path, text, gz=False):
    data = text.encode("latin-1")
    with (gzip.open(path, "wb") if gz else open(path, "wb")) as fh:
        fh.write(data)


CASES = {
    "fastq4": "@r1 c\nACGTAAAACGT\n+\nIIIIIIIIIII\n@r2\nTTTTGGGGACCA\n+r2\nIIIIIIIIIIII\n",
    "crlf": "@r1\r\nACGTAAAACGT\r\n+\r\nIIIIIIIIIII\r\n@r2\r\nTTTT\r\n+\r\nIIII\r\n",
    "fasta_multiline": ">s1 desc\nACGT\nAAAA\n\nCCCC\n>s2\nGG\n>empty\n>s3\nTTTTT",
    "mixed": ">fa\nACGTACGT\n@fq\nAAAA\nCCCC\n+\nIIII\nIIII\n>fb\nGGGG\n",
    "truncated_quality": "@r1\nACGTACGT\n+\nIIIIIIII\n@r2\nAAAACCCC\n+\nIII\n",
    "missing_quality": "@r1\nACGTACGT\n+\nIIIIIIII\n@r2\nAAAACCCC\n+",
    "qual_mismatch_then_more": "@r1\nACGT\n+\nIIIII\n@r2\nAAAA\n+\nIIII\n",
    "leading_garbage": "garbage line\n\n@r1\nACGT\n+\nIIII\n",
    "quality_starting_with_at": "@r1\nACGTACGT\n+\n@IIIIIII\n@r2\nCCCC\n+\n@@@@\n",
    "no_trailing_newline": "@r1\nACGT\n+\nIIII",
    "lone_cr_line": ">x\n\r\nACGT\n",
    "empty_file": "",
    "only_marker": "@",
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("gz", [False, True])
def test_reader_matches_oracle_parser(tmp_path, name, gz):
    path = str(tmp_path / (name + (".gz" if gz else ".txt")))
    _write(path, CASES[name], gz)
    got, n = tj.read_file_stream(path)
    exp, m = orc.parse_file_to_stream(path)
    assert n == m and got.tobytes() == exp.tobytes()


def test_reader_known_streams(tmp_path):
    p = str(tmp_path / "a.fq")
    _write(p, CASES["fastq4"])
    s, n = tj.read_file_stream(p)
    assert n == 2 and s.tobytes() == b"ACGTAAAACGT\nTTTTGGGGACCA\n"
    _write(p, CASES["fasta_multiline"])
    s, n = tj.read_file_stream(p)
    assert n == 4 and s.tobytes() == b"ACGTAAAACCCC\nGG\n\nTTTTT\n"
    _write(p, CASES["truncated_quality"])
    s, n = tj.read_file_stream(p)
    assert n == 1 and s.tobytes() == b"ACGTACGT\n"       # the bad record ends the file (reference loop `>= 0`)
    _write(p, CASES["crlf"])
    s, n = tj.read_file_stream(p)
    assert s.tobytes() == b"ACGTAAAACGT\nTTTT\n"
    with pytest.raises(FileNotFoundError):
        tj.read_file_stream(str(tmp_path / "missing.fq"))


def test_reader_large_blocks(tmp_path):
    # records straddling the 4 MiB read blocks, and a long multi-line FASTA record
    rng = np.random.default_rng(5)
    reads = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 400)))) for _ in range(40000)]
    txt = "".join(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n" for i, r in enumerate(reads))
    long_seq = "".join(rng.choice(list("ACGT"), size=300000))
    txt += ">long\n" + "\n".join(long_seq[i:i + 70] for i in range(0, len(long_seq), 70)) + "\n"
    p = str(tmp_path / "big.fq.gz")
    _write(p, txt, gz=True)
    s, n = tj.read_file_stream(p)
    assert n == len(reads) + 1 and s.tobytes() == ("\n".join(reads) + "\n" + long_seq + "\n").encode()
    e, m = orc.parse_file_to_stream(p)
    assert m == n and e.tobytes() == s.tobytes()


# ---- the multi-threaded feeder (plain files): byte-identical to the one-reader parse, whatever the cut points ---------

@pytest.mark.parametrize("name", sorted(CASES))
def test_feeder_matches_single_reader_on_small_cases(tmp_path, name):
    from tatajuba_amd.capi import read_file_stream_mt
    path = str(tmp_path / (name + ".txt"))
    _write(path, CASES[name] * 40)                         # repeated so that tiny windows cut inside and between records
    exp, m = tj.read_file_stream(path)
    for threads, window in [(1, 4096), (2, 4096), (3, 5000), (7, 4096), (4, 1 << 20)]:
        got, n = read_file_stream_mt(path, threads, window)
        assert n == m and got.tobytes() == exp.tobytes(), (name, threads, window)


def _adversarial_file(rng, n_records):
    """records of every kind the reader knows, with quality strings that look like headers"""
    out = []
    for i in range(n_records):
        L = int(rng.integers(0, 300))
        seq = "".join(rng.choice(list("ACGTN"), size=L))
        kind = rng.integers(0, 10)
        if kind == 0:                                       # FASTA, multi-line
            w = int(rng.integers(1, 80))
            out.append(f">fa{i}\n" + "\n".join(seq[j:j + w] for j in range(0, L, w)) + "\n")
        elif kind == 1:                                     # multi-line FASTQ
            h = L // 2
            q = "".join(rng.choice(list("@>+I#!"), size=L))
            out.append(f"@ml{i}\n{seq[:h]}\n{seq[h:]}\n+\n{q[:h]}\n{q[h:]}\n")
        elif kind == 2:                                     # quality that starts with '@' / '>' / '+'
            q = rng.choice(list("@>+")) + "".join(rng.choice(list("@>+I"), size=max(L - 1, 0))) if L else ""
            out.append(f"@q{i} x\n{seq}\n+\n{q}\n")
        elif kind == 3:                                     # CRLF
            out.append(f"@cr{i}\r\n{seq}\r\n+\r\n{'I' * L}\r\n")
        elif kind == 4:                                     # blank lines between records
            out.append(f"\n\n@b{i}\n{seq}\n+\n{'I' * L}\n\n")
        else:
            out.append(f"@r{i}\n{seq}\n+\n{'F' * L}\n")
    return "".join(out)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_feeder_matches_single_reader_on_adversarial_files(tmp_path, seed):
    from tatajuba_amd.capi import read_file_stream_mt
    rng = np.random.default_rng(seed)
    txt = _adversarial_file(rng, 6000)
    if seed == 3:                                           # a record with a bad quality string in the middle: the file ends there
        cut = len(txt) // 2
        cut = txt.index("\n@r", cut) + 1
        txt = txt[:cut] + "@bad\nACGTACGT\n+\nIII\n" + txt[cut:]
    path = str(tmp_path / "adv.fq")
    _write(path, txt)
    exp, m = tj.read_file_stream(path)
    ora, mo = orc.parse_file_to_stream(path)
    assert mo == m and ora.tobytes() == exp.tobytes()
    assert m > 1000
    for threads, window in [(2, 8192), (5, 20000), (8, 65536), (16, 1 << 20), (3, 1 << 30)]:
        got, n = read_file_stream_mt(path, threads, window)
        assert n == m and got.tobytes() == exp.tobytes(), (seed, threads, window)


@pytest.mark.parametrize("quals", ["@", "#-27<AFI@>+"])
def test_feeder_really_runs_in_parallel_on_ordinary_fastq(tmp_path, quals):
    import ctypes as C
    from tatajuba_amd.capi import read_file_stream_mt
    rng = np.random.default_rng(4)
    reads = ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(20000)]
    p = str(tmp_path / "plain.fq")
    # worst-case qualities: lines that begin with '@', '>' (Phred 29) or '+' like the lines that structure the file
    q = np.frombuffer(quals.encode(), np.uint8)[rng.integers(0, len(quals), size=(20000, 150))]
    _write(p, "".join(f"@read{i} 1:N:0\n{r}\n+\n{q[i].tobytes().decode()}\n" for i, r in enumerate(reads)))
    got, n = read_file_stream_mt(p, 8, 1 << 20)
    assert n == 20000 and got.tobytes() == ("\n".join(reads) + "\n").encode()
    L = tj.lib()
    L.tjamd_debug_feeder_stats.restype = C.c_long
    fb = C.c_long(-1)
    windows = L.tjamd_debug_feeder_stats(C.byref(fb))
    assert fb.value == 0 and windows >= 5                   # every window came from the parallel readers


def test_feeder_long_records_and_gzip_passthrough(tmp_path):
    from tatajuba_amd.capi import read_file_stream_mt
    rng = np.random.default_rng(11)
    long_seq = "".join(rng.choice(list("ACGT"), size=400000))
    txt = ">chr1\n" + "\n".join(long_seq[i:i + 60] for i in range(0, len(long_seq), 60)) + "\n" + CASES["fastq4"] * 500 + ">chr2\n" + long_seq + "\n"
    p = str(tmp_path / "long.fa")
    _write(p, txt)
    exp, m = tj.read_file_stream(p)
    for threads, window in [(4, 8192), (4, 1 << 18), (2, 1 << 22)]:   # records longer than a range: the one-reader fallback
        got, n = read_file_stream_mt(p, threads, window)
        assert n == m == 1002 and got.tobytes() == exp.tobytes()
    g = str(tmp_path / "a.fq.gz")
    _write(g, CASES["fastq4"] * 100, gz=True)
    got, n = read_file_stream_mt(g, 4, 4096)
    exp, m = tj.read_file_stream(g)
    assert n == m == 200 and got.tobytes() == exp.tobytes()


# ---- gzip input through the feeder: inflate runs ahead of the parse; BGZF members are inflated side by side -------------

def _gzip_member(data, extra=None, level=6):
    """one RFC 1952 member around `data`; extra = callable(total_size_without_extra_known) is not needed: BGZF's BC
    subfield holds the member's size - 1, which is known once the deflate stream is"""
    import struct, zlib
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    tail = struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)
    if extra is None:
        return b"\x1f\x8b\x08\x00" + b"\0" * 4 + b"\x00\xff" + body + tail
    total = 12 + 6 + len(body) + 8                          # header(10) + XLEN(2) + BC subfield(6) + stream + CRC/ISIZE
    return b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, total - 1) + body + tail


def _bgzf(data, block=0xff00, eof=True):
    out = [_gzip_member(data[i:i + block], extra="BC") for i in range(0, len(data), block)]
    if eof:
        out.append(_gzip_member(b"", extra="BC"))
    return b"".join(out)


def _check_gz(path, raw_text_path, combos):
    from tatajuba_amd.capi import read_file_stream_mt
    exp, m = tj.read_file_stream(path)                      # one reader over zlib's gzread, as the reference reads it
    if raw_text_path is not None:
        ora, mo = orc.parse_file_to_stream(raw_text_path)
        assert mo == m and ora.tobytes() == exp.tobytes()
    for threads, window in combos:
        got, n = read_file_stream_mt(path, threads, window)
        assert n == m and got.tobytes() == exp.tobytes(), (path, threads, window)
    return m


@pytest.mark.parametrize("seed", [21, 22])
def test_gzip_feeder_matches_gzread_on_adversarial_files(tmp_path, seed):
    import ctypes as C, gzip
    rng = np.random.default_rng(seed)
    txt = _adversarial_file(rng, 5000).encode()
    raw = str(tmp_path / "adv.fq")
    open(raw, "wb").write(txt)
    combos = [(1, 65536), (2, 65536), (5, 100000), (8, 1 << 20), (3, 1 << 30)]
    L = tj.lib()
    L.tjamd_debug_feeder_bgzf_blocks.restype = C.c_long
    # (a) one member, as gzip(1) writes it
    p = str(tmp_path / "one.fq.gz")
    open(p, "wb").write(gzip.compress(txt, 6))
    m = _check_gz(p, raw, combos)
    assert m > 1000 and L.tjamd_debug_feeder_bgzf_blocks() == 0
    # (b) BGZF, as bgzip writes it (with and without the empty last block)
    for eof in (True, False):
        p = str(tmp_path / f"bgzf{int(eof)}.fq.gz")
        open(p, "wb").write(_bgzf(txt, eof=eof))
        assert _check_gz(p, raw, combos) == m
        assert L.tjamd_debug_feeder_bgzf_blocks() >= len(txt) // 0xff00
    # (c) members glued together: plain ones, then BGZF blocks, then a plain one again; small odd-sized BGZF blocks
    cut1, cut2 = len(txt) // 3, 2 * len(txt) // 3
    p = str(tmp_path / "mixed.fq.gz")
    open(p, "wb").write(gzip.compress(txt[:cut1 // 2]) + _gzip_member(txt[cut1 // 2:cut1]) + _bgzf(txt[cut1:cut2], block=1234, eof=False) + gzip.compress(txt[cut2:]))
    assert _check_gz(p, raw, combos) == m
    assert L.tjamd_debug_feeder_bgzf_blocks() >= (cut2 - cut1) // 1234 - 1
    # (d) bytes that are not gzip after the last member are ignored, as gzread ignores them
    p = str(tmp_path / "trail.fq.gz")
    open(p, "wb").write(gzip.compress(txt) + b"\0\0\0\0garbage")
    assert _check_gz(p, raw, combos[:2]) == m


def _fastq_text(n_reads, seed, L=150, quals=b"FFFF:FF,#"):
    rng = np.random.default_rng(seed)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=(n_reads, L))]
    q = np.frombuffer(quals, np.uint8)[rng.integers(0, len(quals), size=(n_reads, L))]
    return b"".join(b"@r%d/1\n%s\n+\n%s\n" % (i, seq[i].tobytes(), q[i].tobytes()) for i in range(n_reads))


@pytest.mark.parametrize("level", [1, 6, 9])
def test_one_member_gzip_is_inflated_on_several_threads(tmp_path, monkeypatch, level):
    """A .gz file of one member (what gzip(1) writes and the reference's gzopen usually gets, src/hopo_counter.c:142): the
    deflate stream is entered at block starts found by trial, the stretches decoded side by side with the window in front
    unknown and resolved afterwards; a stretch counts only if the decoder in front of it stopped exactly on its first bit
    (csrc/feeder.c: tjz_round).  Same bytes as one reader over zlib's gzread, for every thread count and stretch size."""
    import ctypes as C, gzip
    from tatajuba_amd.capi import read_file_stream_mt
    txt = _fastq_text(60000, 100 + level)
    p = str(tmp_path / "one.fq.gz")
    open(p, "wb").write(gzip.compress(txt, level))
    exp, m = tj.read_file_stream(p)
    assert m == 60000
    L = tj.lib()
    L.tjamd_debug_feeder_gz_stretches.restype = C.c_long
    L.tjamd_debug_feeder_gz_false_starts.restype = C.c_long
    for stretch, threads, window in ((1 << 20, 4, 8 << 20), (65536, 8, 1 << 20), (200000, 3, 65536), (65536, 2, 1 << 30)):
        monkeypatch.setenv("TATAJUBA_AMD_GZ_STRETCH", str(stretch))
        got, n = read_file_stream_mt(p, threads, window)
        assert n == m and got.tobytes() == exp.tobytes(), (stretch, threads, window)
        assert L.tjamd_debug_feeder_gz_stretches() > 2, "the member was not entered in the middle"
    monkeypatch.setenv("TATAJUBA_AMD_GZ_PARALLEL", "0")       # one decoder, as before
    got, n = read_file_stream_mt(p, 4, 1 << 20)
    assert got.tobytes() == exp.tobytes() and L.tjamd_debug_feeder_gz_stretches() == 0


def test_one_member_gzip_on_several_threads_odd_streams(tmp_path, monkeypatch):
    """What the search for block starts must not be fooled by, and what it cannot use: stored blocks (level 0), a member
    of fixed-code blocks, bytes that are no text (no block start is accepted: the single decoder takes over), several
    members one after the other, a file cut short, garbage behind the member."""
    import ctypes as C, gzip, zlib
    from tatajuba_amd.capi import read_file_stream_mt
    monkeypatch.setenv("TATAJUBA_AMD_GZ_STRETCH", "65536")
    L = tj.lib()
    L.tjamd_debug_feeder_gz_stretches.restype = C.c_long
    txt = _fastq_text(20000, 7)

    def member(data, **kw):
        co = zlib.compressobj(wbits=31, **kw)
        return co.compress(data) + co.flush()

    def flushed(data, every):                                   # Z_FULL_FLUSH points: stored empty blocks between the coded ones
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        out = []
        for i in range(0, len(data), every):
            out.append(co.compress(data[i:i + every]))
            out.append(co.flush(zlib.Z_FULL_FLUSH if (i // every) % 2 else zlib.Z_SYNC_FLUSH))
        return b"".join(out) + co.flush()

    cases = {"stored": member(txt, level=0), "fixed": member(txt, level=6, strategy=zlib.Z_FIXED), "huffman_only": member(txt, level=6, strategy=zlib.Z_HUFFMAN_ONLY),
             "rle": member(txt, level=6, strategy=zlib.Z_RLE), "flush_points": flushed(txt, 30011),
             "three_members": gzip.compress(txt[:len(txt) // 3], 9) + gzip.compress(txt[len(txt) // 3: len(txt) // 2], 1) + gzip.compress(txt[len(txt) // 2:], 6),
             "garbage_behind": gzip.compress(txt, 6) + b"\0\0junk that is no gzip header"}
    for name, blob in cases.items():
        p = str(tmp_path / (name + ".fq.gz"))
        open(p, "wb").write(blob)
        exp, m = tj.read_file_stream(p)
        assert m == 20000, name
        for threads, window in ((4, 1 << 20), (7, 65536)):
            got, n = read_file_stream_mt(p, threads, window)
            assert n == m and got.tobytes() == exp.tobytes(), (name, threads, window)
    # a FASTA record whose sequence line is not text to the search (bytes >= 0x80 are legal input to the reader): no block
    # start is accepted anywhere, the single decoder does the file
    rng = np.random.default_rng(5)
    blob = b">x\n" + bytes(rng.integers(128, 256, size=3_000_000, dtype=np.uint8).tolist()) + b"\n>y\nACGT\n"
    p = str(tmp_path / "binary.fa.gz")
    open(p, "wb").write(gzip.compress(blob, 6))
    exp, m = tj.read_file_stream(p)
    got, n = read_file_stream_mt(p, 4, 1 << 20)
    assert m == n == 2 and got.tobytes() == exp.tobytes() and L.tjamd_debug_feeder_gz_stretches() <= 2
    # cut short in the middle of the stream: everything the single reader gets before the break, the same way
    whole = gzip.compress(_fastq_text(30000, 9), 6)
    p = str(tmp_path / "cut.fq.gz")
    open(p, "wb").write(whole[: len(whole) * 2 // 3])
    exp, m = tj.read_file_stream(p)
    got, n = read_file_stream_mt(p, 4, 1 << 20)
    assert n == m and got.tobytes() == exp.tobytes() and 10000 < m < 30000


def test_gzip_feeder_records_longer_than_a_view_and_bad_quality(tmp_path):
    import gzip
    rng = np.random.default_rng(31)
    long_seq = "".join(rng.choice(list("ACGT"), size=300000))
    txt = (">chr1\n" + "\n".join(long_seq[i:i + 60] for i in range(0, len(long_seq), 60)) + "\n" + CASES["fastq4"] * 300
           + "@long\n" + long_seq + "\n+\n" + "I" * len(long_seq) + "\n" + CASES["fastq4"] * 300 + ">chr2\n" + long_seq).encode()   # no newline at the end
    raw = str(tmp_path / "long.fa")
    open(raw, "wb").write(txt)
    for name, blob in (("one", gzip.compress(txt)), ("bgzf", _bgzf(txt))):
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(blob)
        assert _check_gz(p, raw, [(4, 65536), (2, 100000), (4, 1 << 22)]) == 1203
    # a bad quality string in the middle ends the file there for every reader
    bad = (CASES["fastq4"] * 2000 + "@bad\nACGTACGTA\n+\nIII\n" + CASES["fastq4"] * 2000).encode()
    raw = str(tmp_path / "bad.fq")
    open(raw, "wb").write(bad)
    p = str(tmp_path / "bad.fq.gz")
    open(p, "wb").write(_bgzf(bad, block=5000))
    m = _check_gz(p, raw, [(4, 65536), (3, 70000)])
    assert 4000 <= m <= 4001


def test_gzip_feeder_truncated_file_stops_where_the_stream_breaks(tmp_path):
    import gzip
    from tatajuba_amd.capi import read_file_stream_mt
    rng = np.random.default_rng(41)
    reads = ["".join(rng.choice(list("ACGT"), size=100)) for _ in range(20000)]
    txt = "".join(f"@r{i}\n{r}\n+\n{'I' * 100}\n" for i, r in enumerate(reads)).encode()
    whole = "\n".join(reads) + "\n"
    for name, blob in (("one", gzip.compress(txt)), ("bgzf", _bgzf(txt))):
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(blob[: len(blob) * 2 // 3])
        got, n = read_file_stream_mt(p, 4, 65536)
        assert 5000 < n < 20000                            # what could be inflated is read, complete records only ...
        assert whole.startswith(got.tobytes().decode()[: -102] if n else "")   # ... and it is a prefix of the file's reads


def test_inflate_decoder_matches_zlib_on_random_streams():
    # the feeder's own DEFLATE decoder (csrc/tj_inflate.c) against zlib: every block type, every level and strategy,
    # flush points, output taken in pieces of odd sizes (the decoder stops and resumes anywhere), and the exact input length
    import ctypes as C, zlib
    L = tj.lib()
    L.tjamd_debug_inflate.restype = C.c_long
    L.tjamd_debug_inflate.argtypes = [C.c_char_p, C.c_long, C.c_void_p, C.c_long, C.c_long, C.POINTER(C.c_long)]
    L.tjamd_debug_crc32.restype = C.c_uint
    L.tjamd_debug_crc32.argtypes = [C.c_char_p, C.c_long]
    rng = random.Random(5)
    nrng = np.random.default_rng(5)

    def make(kind, n):
        if kind == "random":
            return nrng.integers(0, 256, n, dtype=np.uint8).tobytes()
        if kind == "dna":
            return bytes(nrng.choice(np.frombuffer(b"ACGT", np.uint8), n))
        if kind == "fastq":
            return _adversarial_file(nrng, n // 250 + 1).encode()[:n]
        if kind == "runs":
            return b"".join(bytes([rng.randrange(256)]) * rng.choice([1, 2, 3, 7, 8, 9, 100, 258, 259, 5000]) for _ in range(max(1, n // 200)))[:n]
        return (b"the quick brown fox jumps over the lazy dog " * (n // 44 + 1))[:n]

    for it in range(150):
        kind = rng.choice(["random", "dna", "fastq", "runs", "text"])
        data = make(kind, rng.choice([0, 1, 2, 5, 100, 1000, 40000, 70000, 300000]))
        co = zlib.compressobj(rng.randrange(10), zlib.DEFLATED, -15, 9,
                              rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
        parts, step = [], rng.choice([len(data) + 1, 1000, 65536])
        for i in range(0, len(data), step):
            parts.append(co.compress(data[i:i + step]))
            if rng.random() < 0.3:
                parts.append(co.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])))
        comp = b"".join(parts) + co.flush()
        full = comp + b"0123456789abcdef"                   # (a gzip member goes on after the stream)
        out = np.empty(len(data) + 64, np.uint8)
        for chunk in (rng.choice([1, 7, 333, 4096, 70000]), 1 << 22):
            used = C.c_long(-1)
            got = L.tjamd_debug_inflate(full, len(full), out.ctypes.data, out.size, chunk, C.byref(used))
            assert got == len(data) and out[:got].tobytes() == data and used.value == len(comp), (it, kind, len(data), chunk, got)
        assert L.tjamd_debug_crc32(data, len(data)) == zlib.crc32(data)
    for it in range(200):                                   # anything else is refused, not run off with
        junk = nrng.integers(0, 256, rng.choice([1, 10, 100, 5000]), dtype=np.uint8).tobytes()
        out = np.empty(1 << 18, np.uint8)
        L.tjamd_debug_inflate(junk, len(junk), out.ctypes.data, out.size, 4096, None)
    comp = zlib.compress(make("fastq", 200000), 6)[2:-4]
    for cut in (1, 10, 1000, len(comp) // 2, len(comp) - 1):    # a truncated stream says so
        out = np.empty(300000, np.uint8)
        assert L.tjamd_debug_inflate(comp[:cut], cut, out.ctypes.data, out.size, 1 << 20, None) in (-1, -2)


def test_gzip_feeder_refuses_a_member_that_fails_its_checksum(tmp_path, monkeypatch):
    import gzip
    from tatajuba_amd.capi import read_file_stream_mt
    txt = (CASES["fastq4"] * 3000).encode()
    blob = bytearray(gzip.compress(txt, 6))
    blob[-6] ^= 0x40                                        # the stored CRC-32
    p = str(tmp_path / "badcrc.fq.gz")
    open(p, "wb").write(bytes(blob))
    with pytest.raises(Exception):
        read_file_stream_mt(p, 4, 65536)
    good = str(tmp_path / "good.fq.gz")
    open(good, "wb").write(gzip.compress(txt, 6))
    for mode in ("", "zlib"):                               # both inflaters, same bytes
        monkeypatch.setenv("TATAJUBA_AMD_FEEDER_INFLATE", mode)
        got, n = read_file_stream_mt(good, 4, 65536)
        assert n == 6000


def test_fixture_file_stream(golden_dir, known_answers):
    s, n = tj.read_file_stream(os.path.join(golden_dir, "err1750956.fastq.gz"))
    f = known_answers["file"]
    assert n == f["n_reads"] and s.size == f["n_bases"] + f["n_reads"]


def test_synth_stream_properties():
    a = tj.synth_stream(5000, 150, 200000, n_threads=1)
    b = tj.synth_stream(5000, 150, 200000, n_threads=4)
    assert a.tobytes() == b.tobytes() and a.size == 5000 * 151
    m = a.reshape(5000, 151)
    assert (m[:, 150] == 10).all() and np.isin(m[:, :150], np.frombuffer(b"ACGT", np.uint8)).all()
    # both strands occur and reads of one strand are substrings of the genome implied by the other reads' overlaps:
    # cheap proxy -- the oracle finds tracts on both strands and the strand filter keeps most contexts
    o = orc.Oracle(10)
    o.scan_stream(a, 3)
    raw = o.c.n_elem
    assert 5.0 < raw / 5000 < 7.0                               # r-bar ~ 5.97
    # ragged reads and per-sample variants
    r = tj.synth_stream(300, 2000, 500000, read_len_max=20000, n_threads=3)
    lens = np.diff(np.flatnonzero(np.concatenate(([True], r == 10)))) - 1
    lens[0] += 1
    assert lens.min() >= 2000 and lens.max() <= 20000 and len(lens) == 300
    v1 = tj.synth_stream(2000, 150, 200000, variant_seed=1)
    v2 = tj.synth_stream(2000, 150, 200000, variant_seed=2)
    assert v1.tobytes() != v2.tobytes() and v1.tobytes() != a[:v1.size].tobytes()


# ---- a plain C program against include/tatajuba_hopo.h (examples/count_tracts.c) ---------------------------------------

def _build_c_example(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "count_tracts")
    libdir = os.path.join(root, "tatajuba_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "count_tracts.c"),
                           "-L", libdir, "-ltatajuba_amd", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_c_example_links_and_fails_loudly_without_gpu(tmp_path, golden_dir):
    import subprocess
    tj.lib()                                                # (builds the library if need be)
    exe = _build_c_example(tmp_path)
    if tj.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu test")
    r = subprocess.run([exe, os.path.join(golden_dir, "err1750956.fastq.gz"), "-k", "10", "-m", "3"], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr and "no CPU fallback" in r.stderr


# ---- host helpers of the reference header that need no GPU: distances between packed contexts (src/hopo_counter.c:61-113)

def _u64pair(a, b):
    import ctypes as C
    return (C.c_uint64 * 2)(a, b)


def test_context_distances_match_the_oracle():
    import ctypes as C
    L, O = tj.lib(), orc.lib()
    rng = random.Random(11)
    for it in range(4000):
        k = rng.choice([3, 10, 15, 25, 32])
        mask = (1 << (2 * k)) - 1
        a = [rng.getrandbits(64) & mask, rng.getrandbits(64) & mask]
        b = list(a)
        for _ in range(rng.choice([0, 0, 1, 2, 3, 8])):      # a few substitutions
            f, pos = rng.randrange(2), rng.randrange(k)
            b[f] ^= rng.randrange(1, 4) << (2 * pos)
        if rng.random() < 0.3:                               # or a shifted copy (what the edit-shift distance is for)
            f, sft = rng.randrange(2), 2 * rng.randrange(1, 4)
            b[f] = (a[f] >> sft) if rng.random() < 0.5 else ((a[f] << sft) & mask)
        pa, pb = _u64pair(*a), _u64pair(*b)
        assert L.distance_between_context_kmer_pair(pa, pb) == O.orc_distance_pair(pa, pb)
        for md in (1, 2, 5, 64):
            assert L.distance_between_single_context_kmer(pa, pb, md) == O.orc_distance_single(pa, pb, md)
        s1, s2 = (C.c_int * 4)(9, 9, 9, 9), (C.c_int * 4)(9, 9, 9, 9)
        assert L.distance_between_context_kmer_pair_with_edit_shift(pa, pb, s1) == O.orc_distance_pair_shift(pa, pb, s2)
        assert list(s1) == list(s2)
    # by hand: CCG|GAT vs CCA|GAT differ in one base; a flank shifted by one base costs 1
    ccg, gat, cca = 0x25, 0x32, 0x05
    assert L.distance_between_context_kmer_pair(_u64pair(ccg, gat), _u64pair(cca, gat)) == 1
    sh = (C.c_int * 4)()
    assert L.distance_between_context_kmer_pair_with_edit_shift(_u64pair(ccg, gat), _u64pair(ccg >> 2, gat), sh) == 1 and list(sh) == [1, 0, 0, 0]


def test_oracle_grouping_by_hand():
    """greedy grouping of a finalised array (reference: src/context_histogram.c:245-270): an element joins the group being
    built iff it is closer than max_distance_per_flank to every context in it"""
    e = np.zeros(6, dtype=tj.ELEM_DTYPE)
    ctx = [(0x25, 0x32), (0x25, 0x32), (0x24, 0x32), (0x14, 0x32), (0x14, 0x12), (0x00, 0x00)]     # 1 - 2: 1 apart; 2 - 3: 1; 1 - 3: 2
    cnt = [5, 9, 4, 7, 2, 3]
    for i, ((a, b), c) in enumerate(zip(ctx, cnt)):
        e["ctx0"][i], e["ctx1"][i] = a, b
        e["meta"][i] = (4 << 2) | (c << 12)                # base A, length 4
    gof, first, nel, nctx, integ, mode = orc.group_contexts(e, 1)
    assert gof.tolist() == [0, 0, 1, 2, 3, 4]              # distance must be < 1: identical contexts only
    gof, first, nel, nctx, integ, mode = orc.group_contexts(e, 2)
    assert gof.tolist() == [0, 0, 0, 1, 1, 2]              # the 4th is 2 away from the first context of group 0
    assert nel.tolist() == [3, 2, 1] and nctx.tolist() == [2, 2, 1] and integ.tolist() == [18, 9, 3] and mode.tolist() == [1, 3, 5]
    rec = np.zeros((5, 3), np.uint64)
    rec[:, 0] = [9, 9, 9, 7, 7]; rec[:, 1] = [4, 4, 3, 3, 3]; rec[:, 2] = [1, 1, 1, 1, 0]
    ids, n = orc.tract_ids(rec)
    assert ids.tolist() == [0, 0, 1, 2, 3] and n == 4


def test_oracle_context_histograms_by_hand():
    """new_genomic_context_list restated with the indel retry and the length histograms (oracle/context_oracle.c; reference:
    src/context_histogram.c:19-23,245-286).  By hand, k = 4, names `left.A.right`:
      e0 TTTT|GGGG x5 len 4     opens histogram 0
      e1 TTTT|GGGG x9 len 3     same context: joins (idx_match 0), becomes the modal element
      e2 TTTT|GGGC x4 len 4     one substitution: joins at max_distance 2, a second context
      e3 TTTT|GCGG x7 len 4     2 substitutions from e2's context: fails the flank test; edit distance to the modal name
                                TTTT.A.GGGG is 1 -> joins through the retry when levenshtein_distance >= 2, third context
      e4 TTTT|CGGG x2 len 5     2 from e3's context -> fails; edit distance to TTTT.A.GGGG = 1 -> retry
      e5 AAAA|AAAA x3 len 4     far from everything: opens histogram 1
    UNPINNED: the edit distance itself (biomcmc-lib is absent) -- only its use is the reference's."""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    pack = lambda s: sum(code[ch] << (2 * i) for i, ch in enumerate(s))
    rows = [("TTTT", "GGGG", 5, 4), ("TTTT", "GGGG", 9, 3), ("TTTT", "GGGC", 4, 4), ("TTTT", "GCGG", 7, 4), ("TTTT", "CGGG", 2, 5), ("AAAA", "AAAA", 3, 4)]
    e = np.zeros(len(rows), dtype=tj.ELEM_DTYPE)
    for i, (l, r, c, ln) in enumerate(rows):
        e["ctx0"][i], e["ctx1"][i] = pack(l), pack(r)
        e["meta"][i] = (ln << 2) | (c << 12)
    e["read_offset"] = -1
    assert orc.levenshtein("kitten", "sitting") == 3 and orc.levenshtein("TTTT.A.GCGG", "TTTT.A.GGGG") == 1 and orc.levenshtein("ACGT", "CGTA") == 2
    r = orc.genomic_context_list(e, 4, 2, 2, 3)
    assert r["group_of"].tolist() == [0, 0, 0, 0, 0, 1] and r["join_type"].tolist() == [0, 1, 1, 2, 2, 0]
    g = r["groups"]
    assert g["n_elem"].tolist() == [5, 1] and g["n_context"].tolist() == [4, 1] and g["indel"].tolist() == [1, 0]
    assert g["mode"].tolist() == [1, 5] and g["mode_context_id"].tolist() == [0, 0] and g["mode_context_length"].tolist() == [3, 4]
    assert g["integral"].tolist() == [27, 3]
    # lengths of histogram 0: 4 -> 5 + 4 + 7 = 16, 3 -> 9, 5 -> 2; highest count first
    assert g["n_len"].tolist() == [3, 1] and r["hist_len"][:3].tolist() == [4, 3, 5] and r["hist_freq"][:3].tolist() == [16, 9, 2]
    assert g["modal_len"].tolist() == [4, 4] and g["modal_freq"].tolist() == [16, 3]
    assert r["contexts"][:4].tolist() == [[pack("TTTT"), pack("GGGG")], [pack("TTTT"), pack("GGGC")], [pack("TTTT"), pack("GCGG")], [pack("TTTT"), pack("CGGG")]]
    # the retry off (levenshtein_distance 0 can never be undercut): e3 opens a histogram, e4 is 2 away from e3 -> another one
    r0 = orc.genomic_context_list(e, 4, 2, 0, 3)
    assert r0["group_of"].tolist() == [0, 0, 0, 1, 2, 3] and r0["groups"]["indel"].sum() == 0
    gof, first, nel, nctx, integ, mode = orc.group_contexts(e, 2)      # round 2's restatement of the flank test alone agrees
    assert gof.tolist() == r0["group_of"].tolist() and nctx.tolist() == r0["groups"]["n_context"].tolist()


def test_oracle_merge_order_by_hand():
    """the cross-sample merge order restated (oracle/context_oracle.c; reference: src/genome_set.c:250-289, on a tie the
    new genome's entry goes first :278-281), context-keyed: descending (base, ctx0, ctx1, signed length)"""
    def rec(c0, c1, base, ln, cnt, flag=3):
        return [c0, c1, base | ((ln & 0x3ff) << 2) | (cnt << 12) | (0xffe << 32) | (flag << 49)]
    s0 = [rec(9, 4, 1, 5, 10), rec(9, 4, 1, 3, 2), rec(7, 1, 0, 4, 6)]
    s1 = [rec(9, 4, 1, 5, 7, flag=1), rec(8, 8, 1, 4, 3), rec(7, 1, 0, 4, 1)]
    s2 = [rec(9, 9, 1, 2, 5), rec(7, 1, 0, 4, 8)]
    allr = np.array(s0 + s1 + s2, dtype=np.uint64)
    cs, ci, keys, mat = orc.merge_samples(allr, [3, 3, 2])
    # concatenated list: (sample, index), equal keys with the later sample first
    assert list(zip(cs.tolist(), ci.tolist())) == [(2, 0), (1, 0), (0, 0), (0, 1), (1, 1), (2, 1), (1, 2), (0, 2)]
    assert keys[:, 0].tolist() == [9, 9, 9, 8, 7] and keys[:, 1].tolist() == [9, 4, 4, 8, 1]
    assert mat.tolist() == [[0, 0, 5], [10, 7, 0], [2, 0, 0], [0, 3, 0], [6, 1, 8]]
    d = orc.decode_meta(keys[:, 2])
    assert d["count"].tolist() == [5, 17, 2, 3, 15] and d["canon_flag"].tolist() == [3, 3, 3, 3, 3] and d["length"].tolist() == [2, 5, 3, 4, 4]


def test_rccl_exchange_entry_points_check_their_arguments_without_a_gpu():
    """the process-per-GPU exchange lives in the C library (ncclAllGather behind tjamd_allgather_histograms): the library
    exports the entry points, refuses bad arguments before touching a device, and looks RCCL up when a communicator call
    first needs it -- it is not a load-time dependency of the one-GPU drop-in, and a missing RCCL is said loudly"""
    import subprocess, sys
    L = tj.lib()
    out = subprocess.run(["readelf", "-d", tj.library_path()], capture_output=True, text=True).stdout
    assert "librccl" not in out and "libamdhip64" in out
    code = ("import ctypes as C, tatajuba_amd.capi as tj; L = tj.lib(); b = C.create_string_buffer(128); "
            "rc = L.tjamd_comm_unique_id(b); print(rc, L.tjamd_last_error().decode())")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, TATAJUBA_AMD_RCCL="/nonexistent/librccl.so"))
    assert r.returncode == 0, r.stderr
    assert r.stdout.split()[0] != "0" and "RCCL not found (tried /nonexistent/librccl.so)" in r.stdout
    assert L.tjamd_comm_unique_id(None) != 0 and b"null buffer" in L.tjamd_last_error()
    ident = C.create_string_buffer(128)
    assert L.tjamd_comm_create(None, ident, 0, 1) is None and b"bad arguments" in L.tjamd_last_error()
    cnt = (C.c_long * 2)()
    ptr = C.c_void_p()
    assert L.tjamd_allgather_histograms(None, None, C.byref(ptr), cnt) < 0 and b"bad arguments" in L.tjamd_last_error()
    assert L.tjamd_comm_rank(None) == -1 and L.tjamd_comm_world(None) == -1 and L.tjamd_comm_collectives(None) == -1
    L.tjamd_comm_destroy(None)
    buf = C.create_string_buffer(64)
    assert L.tjamd_peer_access_report(buf, 64) == 0 and buf.value == b""


def test_edit_distance_readings_differ_on_a_shifted_name():
    """UNPINNED piece, made explicit: biomcmc_levenshtein_distance (.., 1, 1, true) is absent from the reference tree.  The
    restatement (oracle, device, host alike) is the global unit-cost distance -- reading (a) of oracle/context_oracle.c's
    header: the flag strips common borders, which changes nothing.  Under reading (b), free end gaps, a name whose right
    flank is shifted by one base would cost 1 instead of 2; this pair is where the two readings part, for whoever holds
    biomcmc-lib to settle with one call."""
    a, b = "ACGTACGTAC.A.TTGACCATGG", "ACGTACGTAC.A.TGACCATGGA"       # right flank shifted left by one base
    assert orc.levenshtein(a, b) == 2                                 # global: one deletion + one insertion

    def free_end(x, y):                                               # reading (b) by hand: one string may end early, the rest of the other is free
        d = [[0] * (len(y) + 1) for _ in range(len(x) + 1)]
        for i in range(len(x) + 1):
            for j in range(len(y) + 1):
                if i == 0 or j == 0:
                    d[i][j] = i + j
                else:
                    d[i][j] = min(d[i - 1][j - 1] + (x[i - 1] != y[j - 1]), d[i - 1][j] + 1, d[i][j - 1] + 1)
        return min(min(d[len(x)]), min(row[len(y)] for row in d))
    assert free_end(a, b) == 1 == orc.levenshtein(a, b, free_end=True)
    rng = random.Random(77)
    for _ in range(300):                                              # the oracle's second reading against the by-hand table
        x = "".join(rng.choice("ACGT.") for _ in range(rng.randrange(0, 14)))
        y = "".join(rng.choice("ACGT.") for _ in range(rng.randrange(0, 14)))
        assert orc.levenshtein(x, y, free_end=True) == free_end(x, y) <= orc.levenshtein(x, y)
    # the host function a caller links (include/tatajuba_context.h) follows the same reading as the oracle
    import ctypes as C
    L = tj.lib()
    from tatajuba_amd.capi import ContextHistogramStruct
    ch = ContextHistogramStruct()
    ch.name = a.encode()
    L.indel_distance_between_context_histogram_and_hopo_context.restype = C.c_int
    L.indel_distance_between_context_histogram_and_hopo_context.argtypes = [C.c_void_p, C.c_char_p]
    assert L.indel_distance_between_context_histogram_and_hopo_context(C.byref(ch), b.encode()) == 2
    # ... and the other reading when asked for (TATAJUBA_AMD_EDIT_DISTANCE=free_end, read once per process: a child)
    import subprocess, sys
    code = ("import ctypes as C, tatajuba_amd as tj; from tatajuba_amd.capi import ContextHistogramStruct; L = tj.lib(); ch = ContextHistogramStruct(); "
            f"ch.name = {a.encode()!r}; L.indel_distance_between_context_histogram_and_hopo_context.restype = C.c_int; "
            "L.indel_distance_between_context_histogram_and_hopo_context.argtypes = [C.c_void_p, C.c_char_p]; "
            f"print(L.indel_distance_between_context_histogram_and_hopo_context(C.byref(ch), {b.encode()!r}))")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, TATAJUBA_AMD_EDIT_DISTANCE="free_end"))
    assert r.returncode == 0 and r.stdout.split()[-1] == "1", r.stderr
