"""ctypes binding of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing under
tatajuba_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# numpy view of hopo_element (include/tatajuba_hopo.h; reference src/hopo_counter.h:34-49)
ELEM_DTYPE = np.dtype([("ctx0", "<u8"), ("ctx1", "<u8"), ("meta", "<u8"), ("read_offset", "<i4"),
                       ("loc_ref_id", "<i4"), ("loc_pos", "<i4"), ("loc_last", "<i4")])
assert ELEM_DTYPE.itemsize == 40


def _sx(v, bits):
    """sign-extend the low `bits` bits of an int64 array"""
    v = v.astype(np.int64) & ((1 << bits) - 1)
    return np.where(v >= (1 << (bits - 1)), v - (1 << bits), v)


def decode_meta(meta):
    """bitfield word -> dict of signed fields (LSB-first layout, see tatajuba_hopo.h)"""
    m = np.asarray(meta, dtype=np.uint64)
    sh = lambda s: (m >> np.uint64(s))
    return {"base": _sx(sh(0), 2), "length": _sx(sh(2), 10), "count": _sx(sh(12), 20),
            "mismatches": _sx(sh(32), 12), "multi": _sx(sh(44), 3), "neg_strand": _sx(sh(47), 2),
            "canon_flag": _sx(sh(49), 3)}


class _OrcCounter(C.Structure):
    _fields_ = [("elem", C.c_void_p), ("n_elem", C.c_int), ("n_alloc", C.c_int), ("kmer_size", C.c_int),
                ("coverage", C.c_int), ("ref_start", C.c_int), ("idx_initial", C.POINTER(C.c_int)),
                ("idx_final", C.POINTER(C.c_int)), ("n_idx", C.c_int), ("n_reads", C.c_long),
                ("n_undefined", C.c_long), ("status", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("hopo_oracle.c", "context_oracle.c", "hopo_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_new.restype = C.POINTER(_OrcCounter); L.orc_new.argtypes = [C.c_int]
        L.orc_free.argtypes = [C.POINTER(_OrcCounter)]
        L.orc_scan_seq.argtypes = [C.POINTER(_OrcCounter), C.c_char_p, C.c_int, C.c_int]
        L.orc_scan_seq_all_monomers.argtypes = [C.POINTER(_OrcCounter), C.c_char_p, C.c_int]
        L.orc_scan_file.restype = C.c_long; L.orc_scan_file.argtypes = [C.POINTER(_OrcCounter), C.c_char_p, C.c_int]
        L.orc_scan_stream.restype = C.c_long
        L.orc_scan_stream.argtypes = [C.POINTER(_OrcCounter), C.c_void_p, C.c_size_t, C.c_int]
        L.orc_finalise.argtypes = [C.POINTER(_OrcCounter), C.c_int, C.c_int]
        U64P = C.POINTER(C.c_uint64)
        L.orc_distance_single.argtypes = [U64P, U64P, C.c_int]
        L.orc_distance_pair.argtypes = [U64P, U64P]
        L.orc_distance_pair_shift.argtypes = [U64P, U64P, C.POINTER(C.c_int)]
        L.orc_group_contexts.restype = C.c_long
        L.orc_group_contexts.argtypes = [C.c_void_p, C.c_long, C.c_int] + [C.c_void_p] * 6
        L.orc_tract_ids.restype = C.c_long; L.orc_tract_ids.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        L.orc_name_from_contexts.restype = C.c_void_p; L.orc_name_from_contexts.argtypes = [U64P, C.c_int, C.c_int, C.c_int]
        L.orc_levenshtein.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.orc_levenshtein_mode.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_set_edit_free_end.argtypes = [C.c_int]
        L.orc_set_edit_free_end.restype = None
        L.orc_genomic_context_list.restype = C.c_long
        L.orc_genomic_context_list.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
        L.orc_merge_samples.restype = C.c_long
        L.orc_merge_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_parse_file_to_stream.restype = C.c_void_p
        L.orc_parse_file_to_stream.argtypes = [C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_long)]
        _LIB = L
    return _LIB


class Oracle:
    """One sample's counter on the CPU oracle."""

    def __init__(self, kmer_size):
        self.k = kmer_size
        self._p = lib().orc_new(kmer_size)

    def close(self):
        if self._p:
            lib().orc_free(self._p)
            self._p = None

    __del__ = close

    def scan_seq(self, seq, m):
        if isinstance(seq, str):
            seq = seq.encode("latin-1")
        lib().orc_scan_seq(self._p, seq, len(seq), m)

    def scan_seq_all_monomers(self, seq):
        if isinstance(seq, str):
            seq = seq.encode("latin-1")
        lib().orc_scan_seq_all_monomers(self._p, seq, len(seq))

    def scan_file(self, path, m):
        n = lib().orc_scan_file(self._p, os.fsencode(path), m)
        if n < 0:
            raise FileNotFoundError(path)
        return n

    def scan_stream(self, buf, m):
        """buf: bytes / numpy uint8 array of '\\n'-terminated reads"""
        a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
        a = np.ascontiguousarray(a)
        return lib().orc_scan_stream(self._p, a.ctypes.data, a.size, m)

    def finalise(self, remove_biased, min_coverage):
        lib().orc_finalise(self._p, int(bool(remove_biased)), int(min_coverage))

    # ---- views -------------------------------------------------------------------------------------------
    @property
    def c(self):
        return self._p.contents

    def elems(self):
        n = self.c.n_elem
        if n == 0:
            return np.zeros(0, dtype=ELEM_DTYPE)
        buf = (C.c_char * (n * 40)).from_address(self.c.elem)
        return np.frombuffer(buf, dtype=ELEM_DTYPE).copy()

    def idx(self):
        n = self.c.n_idx
        if n == 0 or not self.c.idx_initial:
            return np.zeros(0, np.int32), np.zeros(0, np.int32)
        return (np.ctypeslib.as_array(self.c.idx_initial, (n,)).copy(),
                np.ctypeslib.as_array(self.c.idx_final, (n,)).copy())


def parse_file_to_stream(path):
    n = C.c_size_t(0)
    r = C.c_long(0)
    p = lib().orc_parse_file_to_stream(os.fsencode(path), C.byref(n), C.byref(r))
    if not p:
        raise FileNotFoundError(path)
    out = np.frombuffer((C.c_char * n.value).from_address(p), dtype=np.uint8).copy() if n.value else np.zeros(0, np.uint8)
    C.CDLL(None).free(C.c_void_p(p))
    return out, r.value


def name_of(ctx0, ctx1, base, k):
    """left.base.right string (reference: generate_name_from_flanking_contexts, src/hopo_counter.c:471-493)"""
    dna = "ACGT"
    left = "".join(dna[(int(ctx0) >> (2 * i)) & 3] for i in range(k))
    right = "".join(dna[(int(ctx1) >> (2 * i)) & 3] for i in range(k))
    return f"{left}.{dna[int(base)]}.{right}"


def group_contexts(elems, max_distance_per_flank):
    """oracle restatement of the reference's greedy grouping on a finalised element array (see hopo_oracle.c)"""
    e = np.ascontiguousarray(elems)
    n = len(e)
    gof = np.zeros(max(n, 1), np.int32)
    first, nel, nctx, mode = (np.zeros(max(n, 1), np.int32) for _ in range(4))
    integ = np.zeros(max(n, 1), np.int64)
    ng = lib().orc_group_contexts(e.ctypes.data, n, max_distance_per_flank, gof.ctypes.data, first.ctypes.data, nel.ctypes.data,
                                  nctx.ctypes.data, integ.ctypes.data, mode.ctypes.data)
    return gof[:n], first[:ng], nel[:ng], nctx[:ng], integ[:ng], mode[:ng]


def tract_ids(rec3):
    r = np.ascontiguousarray(rec3)
    n = len(r)
    out = np.zeros(max(n, 1), np.int32)
    nid = lib().orc_tract_ids(r.ctypes.data, n, out.ctypes.data)
    return out[:n], nid


ORC_GROUP_DTYPE = np.dtype([("first", "<i4"), ("n_elem", "<i4"), ("n_context", "<i4"), ("mode", "<i4"), ("indel", "<i4"), ("n_len", "<i4"),
                            ("modal_len", "<i4"), ("modal_freq", "<i4"), ("mode_context_id", "<i4"), ("mode_context_count", "<i4"),
                            ("mode_context_length", "<i4"), ("coverage", "<i4"), ("n_tracts", "<i4"), ("_pad", "<i4"), ("integral", "<i8")])
assert ORC_GROUP_DTYPE.itemsize == 64


def levenshtein(a, b, cost_sub=1, cost_indel=1, free_end=False):
    """UNPINNED restatement of biomcmc_levenshtein_distance (see context_oracle.c); free_end: its second reading"""
    a, b = (x.encode("latin-1") if isinstance(x, str) else x for x in (a, b))
    return lib().orc_levenshtein_mode(a, len(a), b, len(b), cost_sub, cost_indel, int(bool(free_end)))


def set_edit_free_end(on):
    """which reading genomic_context_list's retry uses (default: the global distance)"""
    lib().orc_set_edit_free_end(int(bool(on)))


def genomic_context_list(elems, kmer_size, max_distance_per_flank, levenshtein_distance, min_tract_size, coverage=0):
    """oracle restatement of new_genomic_context_list (grouping with the indel retry + length histograms, see
    context_oracle.c).  Returns dict: group_of, join_type (per element), groups (ORC_GROUP_DTYPE), hist_len, hist_freq
    (per-element arrays: group g's histogram at [first, first + n_len)), contexts (uint64 [n, 2]: g's list at first ...)"""
    e = np.ascontiguousarray(elems)
    n = len(e)
    m = max(n, 1)
    gof, jt, hl, hf = (np.zeros(m, np.int32) for _ in range(4))
    g = np.zeros(m, ORC_GROUP_DTYPE)
    ctx = np.zeros((m, 2), np.uint64)
    ng = lib().orc_genomic_context_list(e.ctypes.data, n, kmer_size, max_distance_per_flank, levenshtein_distance, min_tract_size, int(coverage),
                                        gof.ctypes.data, jt.ctypes.data, g.ctypes.data, hl.ctypes.data, hf.ctypes.data, ctx.ctypes.data)
    return {"group_of": gof[:n], "join_type": jt[:n], "groups": g[:ng], "hist_len": hl[:n], "hist_freq": hf[:n], "contexts": ctx[:n]}


def merge_samples(rec3, counts):
    """oracle restatement of the cross-sample merge order (src/genome_set.c:250-289) on context keys.  rec3: uint64 [n, 3]
    (the samples' kept tjamd_records back to back), counts: records per sample.  Returns (cat_sample, cat_index,
    keys uint64 [n_union, 3], per-sample counts int32 [n_union, n_samples])."""
    r = np.ascontiguousarray(rec3, dtype=np.uint64).reshape(-1, 3)
    n, ns = len(r), len(counts)
    cnt = (C.c_long * ns)(*[int(x) for x in counts])
    cs, ci = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    keys = np.zeros((max(n, 1), 3), np.uint64)
    mat = np.zeros((max(n, 1), ns), np.int32)
    nu = lib().orc_merge_samples(r.ctypes.data, cnt, ns, cs.ctypes.data, ci.ctypes.data, keys.ctypes.data, mat.ctypes.data)
    return cs[:n], ci[:n], keys[:nu], mat[:nu]
