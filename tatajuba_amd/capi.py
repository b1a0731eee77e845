"""ctypes mirror of the C boundary (include/tatajuba_hopo.h and include/tatajuba_amd.h).

Two views of the same library:
  * HopoCounter / Options -- the drop-in API with the reference's names (new_or_append_hopo_counter_from_file,
    update_hopo_counter_from_seq, finalise_hopo_counter, del_hopo_counter; reference: src/hopo_counter.h:68-80), so that
    parity tests read like calls into tatajuba itself;
  * Counter -- the tjamd_* extension for device-resident read streams (benchmark, multi-GPU).
Nothing here computes: every result comes from the HIP kernels through the C ABI, and a missing library or GPU raises.
"""
import ctypes as C
import os

import numpy as np

from .build import build_library, library_path

ELEM_DTYPE = np.dtype([("ctx0", "<u8"), ("ctx1", "<u8"), ("meta", "<u8"), ("read_offset", "<i4"),
                       ("loc_ref_id", "<i4"), ("loc_pos", "<i4"), ("loc_last", "<i4")])
RECORD_DTYPE = np.dtype([("ctx0", "<u8"), ("ctx1", "<u8"), ("meta", "<u8")])
GROUP_DTYPE = np.dtype([("first", "<i4"), ("n_elem", "<i4"), ("n_context", "<i4"), ("mode", "<i4"), ("integral", "<i8")])
# tjamd_context_group / tjamd_length_freq (include/tatajuba_amd.h)
CONTEXT_GROUP_DTYPE = np.dtype([("first", "<i4"), ("n_elem", "<i4"), ("n_context", "<i4"), ("mode", "<i4"), ("indel", "<i4"), ("n_len", "<i4"),
                                ("modal_len", "<i4"), ("modal_freq", "<i4"), ("integral", "<i8")])
LENGTH_FREQ_DTYPE = np.dtype([("length", "<i4"), ("freq", "<i4")])
LOCATED_DTYPE = np.dtype([("ctx0", "<u8"), ("ctx1", "<u8"), ("meta", "<u8"), ("pos", "<u8")])


class TatajubaAmdError(RuntimeError):
    pass


def _sx(v, bits):
    v = v.astype(np.int64) & ((1 << bits) - 1)
    return np.where(v >= (1 << (bits - 1)), v - (1 << bits), v)


def decode_meta(meta):
    """hopo_element bitfield word -> signed fields (layout: include/tatajuba_hopo.h)"""
    m = np.asarray(meta, dtype=np.uint64)
    sh = lambda s: (m >> np.uint64(s))
    return {"base": _sx(sh(0), 2), "length": _sx(sh(2), 10), "count": _sx(sh(12), 20),
            "mismatches": _sx(sh(32), 12), "multi": _sx(sh(44), 3), "neg_strand": _sx(sh(47), 2),
            "canon_flag": _sx(sh(49), 3)}


class Options(C.Structure):
    """tatajuba_options_t (reference: src/hopo_counter.h:20-32), passed by value"""
    _fields_ = [("reference_fasta_filename", C.c_char_p), ("outdir", C.c_char_p),
                ("paired_end", C.c_bool), ("remove_biased", C.c_bool), ("save_vcf", C.c_bool),
                ("gff", C.c_void_p),
                ("max_distance_per_flank", C.c_int), ("kmer_size", C.c_int), ("min_tract_size", C.c_int),
                ("levenshtein_distance", C.c_int), ("min_coverage", C.c_int), ("n_samples", C.c_int),
                ("n_threads", C.c_int)]

    @classmethod
    def defaults(cls, kmer_size=25, min_tract_size=4, min_coverage=5, remove_biased=True, paired_end=False):
        # defaults and clamps of the reference CLI: src/main.c:56-60,184-192
        o = cls()
        o.kmer_size, o.min_tract_size, o.min_coverage = kmer_size, min_tract_size, min_coverage
        o.remove_biased, o.paired_end = remove_biased, paired_end
        o.max_distance_per_flank, o.levenshtein_distance, o.n_samples, o.n_threads = 1, 2, 1, 1
        return o


class _HopoCounterStruct(C.Structure):
    """struct hopo_counter_struct (reference: src/hopo_counter.h:51-59)"""
    _fields_ = [("elem", C.c_void_p), ("name", C.c_char_p),
                ("ref_start", C.c_int), ("n_elem", C.c_int), ("n_alloc", C.c_int), ("kmer_size", C.c_int),
                ("coverage", C.c_int),
                ("idx_initial", C.POINTER(C.c_int)), ("idx_final", C.POINTER(C.c_int)), ("n_idx", C.c_int),
                ("opt", Options), ("ref_counter", C.c_int)]


assert C.sizeof(Options) == 64 and C.sizeof(_HopoCounterStruct) == 136

_LIB = None

# every symbol the two headers declare (checked by tests/test_cabi.py against the .so and the header text)
EXPORTS = [
    "new_hopo_counter", "del_hopo_counter", "new_or_append_hopo_counter_from_file", "update_hopo_counter_from_seq",
    "update_hopo_counter_from_seq_all_monomers",
    "finalise_hopo_counter", "compare_hopo_element_decreasing", "compare_hopo_context",
    "generate_name_from_flanking_contexts", "generate_tract_as_string", "print_tatajuba_options",
    "distance_between_single_context_kmer", "distance_between_context_kmer_pair", "distance_between_context_kmer_pair_with_edit_shift",
    "leftmost_hopo_name_and_length_from_string", "hopo_counter_histogram_integral",
    "dna_in_2_bits", "bit_2_dna",
    "tjamd_device_count", "tjamd_source_hash", "tjamd_last_error", "tjamd_version", "tjamd_counter_create", "tjamd_counter_destroy",
    "tjamd_counter_reset", "tjamd_counter_set_stream", "tjamd_counter_set_order_stream", "tjamd_counter_device", "tjamd_scan_device", "tjamd_scan_host",
    "tjamd_scan_host_located", "tjamd_read_file_stream_mt", "tjamd_host_alloc", "tjamd_host_free", "tjamd_device_alloc", "tjamd_device_free", "tjamd_device_download", "tjamd_sync", "tjamd_mark", "tjamd_wait_mark", "tjamd_reserve", "tjamd_raw_count",
    "tjamd_download_raw", "tjamd_undefined_runs", "tjamd_upload_raw", "tjamd_finalise", "tjamd_finalise_begin", "tjamd_finalise_end", "tjamd_kept_count",
    "tjamd_n_idx", "tjamd_coverage", "tjamd_download_kept", "tjamd_download_idx", "tjamd_kept_device_ptr",
    "tjamd_merge_samples", "tjamd_gather_histograms", "tjamd_peer_access_report", "tjamd_comm_unique_id", "tjamd_comm_create", "tjamd_comm_destroy",
    "tjamd_comm_set_stream", "tjamd_comm_rank", "tjamd_comm_world", "tjamd_comm_collectives", "tjamd_comm_count", "tjamd_comm_last_exchange", "tjamd_last_merge_ms", "tjamd_allgather_histograms", "tjamd_tract_ids", "tjamd_group_contexts", "tjamd_context_histograms", "tjamd_scan_windows", "tjamd_thread_cleanup", "tjamd_last_scan_ms", "tjamd_last_partition_ms", "tjamd_counter_uses_log", "tjamd_last_finalise_ms", "tjamd_last_scan_launches", "tjamd_plan_mismatches",
    "tjamd_synth_stream", "tjamd_read_file_stream",
    # include/tatajuba_context.h
    "new_genomic_context_list", "del_genomic_context_list", "del_context_histogram",
    "distance_between_context_histogram_and_hopo_context", "indel_distance_between_context_histogram_and_hopo_context",
]


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch's wheel bundles its own libamdhip64.so (soname libamdhip64.so.7, the same
    soname our library needs); if ours pulled in /opt/rocm's copy first, a later `import torch` would load a second
    runtime and see no GPU.  So when torch is installed, map its copy first (without importing torch) and let the
    dynamic loader bind libtatajuba_amd.so to it by soname.  TATAJUBA_AMD_HIP_RUNTIME=system skips this."""
    if os.environ.get("TATAJUBA_AMD_HIP_RUNTIME", "") == "system":
        return
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        build_library()
    _share_hip_runtime_with_torch()
    L = C.CDLL(path)
    P = C.POINTER(_HopoCounterStruct)
    L.new_hopo_counter.restype = P; L.new_hopo_counter.argtypes = [C.c_int]
    L.del_hopo_counter.restype = None; L.del_hopo_counter.argtypes = [P]
    L.new_or_append_hopo_counter_from_file.restype = P
    L.new_or_append_hopo_counter_from_file.argtypes = [P, C.c_char_p, Options]
    L.update_hopo_counter_from_seq.restype = None
    L.update_hopo_counter_from_seq.argtypes = [P, C.c_char_p, C.c_int, C.c_int]
    L.update_hopo_counter_from_seq_all_monomers.restype = None
    L.update_hopo_counter_from_seq_all_monomers.argtypes = [P, C.c_char_p, C.c_int]
    L.finalise_hopo_counter.restype = None; L.finalise_hopo_counter.argtypes = [P]
    L.generate_name_from_flanking_contexts.restype = C.c_void_p
    L.generate_name_from_flanking_contexts.argtypes = [C.POINTER(C.c_uint64), C.c_int8, C.c_int, C.c_bool]
    L.generate_tract_as_string.restype = C.c_void_p
    L.generate_tract_as_string.argtypes = [C.POINTER(C.c_uint64), C.c_int8, C.c_int, C.c_int, C.c_bool]
    L.compare_hopo_element_decreasing.restype = C.c_int
    L.compare_hopo_element_decreasing.argtypes = [C.c_void_p, C.c_void_p]

    L.tjamd_device_count.restype = C.c_int
    L.tjamd_last_error.restype = C.c_char_p
    L.tjamd_version.restype = C.c_char_p
    L.tjamd_counter_create.restype = C.c_void_p; L.tjamd_counter_create.argtypes = [C.c_int, C.c_int]
    L.tjamd_counter_destroy.restype = None; L.tjamd_counter_destroy.argtypes = [C.c_void_p]
    L.tjamd_counter_reset.argtypes = [C.c_void_p]
    L.tjamd_counter_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.tjamd_counter_device.argtypes = [C.c_void_p]
    L.tjamd_scan_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.tjamd_scan_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.tjamd_scan_host_located.restype = C.c_long
    L.tjamd_scan_host_located.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_long]
    L.tjamd_host_alloc.restype = C.c_void_p; L.tjamd_host_alloc.argtypes = [C.c_size_t]
    L.tjamd_host_free.restype = None; L.tjamd_host_free.argtypes = [C.c_void_p]
    L.tjamd_sync.argtypes = [C.c_void_p]
    L.tjamd_raw_count.restype = C.c_long; L.tjamd_raw_count.argtypes = [C.c_void_p]
    L.tjamd_download_raw.restype = C.c_long; L.tjamd_download_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    L.tjamd_undefined_runs.restype = C.c_long; L.tjamd_undefined_runs.argtypes = [C.c_void_p]
    L.tjamd_upload_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    L.tjamd_finalise.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.tjamd_finalise_begin.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.tjamd_finalise_end.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.tjamd_kept_count.restype = C.c_long; L.tjamd_kept_count.argtypes = [C.c_void_p]
    L.tjamd_n_idx.argtypes = [C.c_void_p]
    L.tjamd_coverage.argtypes = [C.c_void_p]
    L.tjamd_download_kept.restype = C.c_long; L.tjamd_download_kept.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    L.tjamd_download_idx.restype = C.c_long
    L.tjamd_download_idx.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    L.tjamd_kept_device_ptr.restype = C.c_void_p; L.tjamd_kept_device_ptr.argtypes = [C.c_void_p]
    L.tjamd_merge_samples.restype = C.c_long
    L.tjamd_merge_samples.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_long), C.c_int, C.c_void_p, C.c_void_p, C.c_long]
    L.tjamd_gather_histograms.restype = C.c_long
    L.tjamd_gather_histograms.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_long)]
    L.tjamd_tract_ids.restype = C.c_long; L.tjamd_tract_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
    L.tjamd_group_contexts.restype = C.c_long
    L.tjamd_group_contexts.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
    L.new_genomic_context_list.restype = C.POINTER(GenomicContextListStruct); L.new_genomic_context_list.argtypes = [P]
    L.del_genomic_context_list.restype = None; L.del_genomic_context_list.argtypes = [C.POINTER(GenomicContextListStruct)]
    L.tjamd_comm_unique_id.argtypes = [C.c_void_p]
    L.tjamd_comm_create.restype = C.c_void_p; L.tjamd_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.tjamd_comm_destroy.restype = None; L.tjamd_comm_destroy.argtypes = [C.c_void_p]
    L.tjamd_comm_rank.argtypes = [C.c_void_p]; L.tjamd_comm_world.argtypes = [C.c_void_p]
    L.tjamd_comm_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.tjamd_comm_collectives.restype = C.c_long; L.tjamd_comm_collectives.argtypes = [C.c_void_p]
    L.tjamd_allgather_histograms.restype = C.c_long
    L.tjamd_allgather_histograms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_long)]
    L.tjamd_peer_access_report.argtypes = [C.c_char_p, C.c_int]
    L.tjamd_context_histograms.restype = C.c_long
    L.tjamd_context_histograms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    # (without these ctypes truncates the 64-bit device pointer to a C int)
    L.tjamd_device_alloc.restype = C.c_void_p; L.tjamd_device_alloc.argtypes = [C.c_void_p, C.c_size_t]
    L.tjamd_device_free.restype = None; L.tjamd_device_free.argtypes = [C.c_void_p, C.c_void_p]
    L.tjamd_device_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.tjamd_scan_windows.restype = C.c_long
    L.tjamd_scan_windows.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
    U64P = C.POINTER(C.c_uint64)
    L.distance_between_single_context_kmer.argtypes = [U64P, U64P, C.c_int]
    L.distance_between_context_kmer_pair.argtypes = [U64P, U64P]
    L.distance_between_context_kmer_pair_with_edit_shift.argtypes = [U64P, U64P, C.POINTER(C.c_int)]
    L.leftmost_hopo_name_and_length_from_string.restype = C.c_void_p
    L.leftmost_hopo_name_and_length_from_string.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.hopo_counter_histogram_integral.argtypes = [P, C.c_int]
    L.tjamd_last_scan_ms.restype = C.c_double; L.tjamd_last_scan_ms.argtypes = [C.c_void_p]
    L.tjamd_last_finalise_ms.restype = C.c_double; L.tjamd_last_finalise_ms.argtypes = [C.c_void_p]
    L.tjamd_last_partition_ms.restype = C.c_double; L.tjamd_last_partition_ms.argtypes = [C.c_void_p]
    L.tjamd_last_scan_launches.restype = C.c_long; L.tjamd_last_scan_launches.argtypes = [C.c_void_p]
    L.tjamd_synth_stream.restype = C.c_long
    L.tjamd_synth_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_long, C.c_long, C.c_int, C.c_int,
                                     C.c_void_p, C.c_long, C.c_int]
    L.tjamd_read_file_stream.restype = C.c_long
    L.tjamd_read_file_stream.argtypes = [C.c_char_p, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
    _LIB = L
    return L


def _err():
    return (lib().tjamd_last_error() or b"").decode(errors="replace")


def device_count():
    return lib().tjamd_device_count()


def synth_stream(n_reads, read_len, genome_len, seed_genome=0x7A7A0001, seed_reads=0x7A7A1000, variant_seed=0,
                 read_len_max=0, n_threads=8, out=None):
    """Synthetic '\\n'-delimited read stream as a numpy uint8 array (host memory)."""
    need = -lib().tjamd_synth_stream(seed_genome, seed_reads, variant_seed, genome_len, n_reads, read_len, read_len_max,
                                     None, 0, n_threads)
    if need <= 0:
        raise TatajubaAmdError("bad synthetic-stream parameters")
    if out is None:
        out = np.empty(need, dtype=np.uint8)
    assert out.dtype == np.uint8 and out.size >= need and out.flags["C_CONTIGUOUS"]
    got = lib().tjamd_synth_stream(seed_genome, seed_reads, variant_seed, genome_len, n_reads, read_len, read_len_max,
                                   out.ctypes.data, out.size, n_threads)
    if got != need:
        raise TatajubaAmdError("synthetic stream generation failed")
    return out[:need]


def read_file_stream_mt(path, n_threads=4, window_bytes=0):
    """read_file_stream through the multi-threaded feeder (plain files; host-only)."""
    L = lib()
    L.tjamd_read_file_stream_mt.restype = C.c_long
    L.tjamd_read_file_stream_mt.argtypes = [C.c_char_p, C.c_void_p, C.c_long, C.POINTER(C.c_long), C.c_int, C.c_long]
    n = C.c_long(0)
    need = L.tjamd_read_file_stream_mt(os.fsencode(path), None, 0, C.byref(n), n_threads, window_bytes)
    if need < 0:
        raise FileNotFoundError(path)
    out = np.empty(max(need, 1), dtype=np.uint8)
    got = L.tjamd_read_file_stream_mt(os.fsencode(path), out.ctypes.data, need, C.byref(n), n_threads, window_bytes)
    assert got == need
    return out[:need], n.value


def read_file_stream(path):
    """Parse a FASTA/FASTQ(.gz) file with the product's host reader -> (uint8 stream, n_reads).  Host-only."""
    n = C.c_long(0)
    need = lib().tjamd_read_file_stream(os.fsencode(path), None, 0, C.byref(n))
    if need < 0:
        raise FileNotFoundError(path)
    out = np.empty(max(need, 1), dtype=np.uint8)
    got = lib().tjamd_read_file_stream(os.fsencode(path), out.ctypes.data, need, C.byref(n))
    assert got == need
    return out[:need], n.value


class EmpfreqElement(C.Structure):
    _fields_ = [("freq", C.c_int), ("idx", C.c_int)]


class EmpfreqStruct(C.Structure):
    _fields_ = [("i", C.POINTER(EmpfreqElement)), ("n", C.c_int), ("min", C.c_int), ("max", C.c_int)]


class ContextHistogramStruct(C.Structure):
    """struct context_histogram_struct (include/tatajuba_context.h; reference src/context_histogram.h:18-43 without gffeature)"""
    _fields_ = [("context", C.POINTER(C.c_uint64)),
                ("base", C.c_int32, 2), ("multi", C.c_int32, 3), ("indel", C.c_int32, 2), ("neg_strand", C.c_int32, 1), ("mismatches", C.c_int32, 12),
                ("name", C.c_char_p), ("n_context", C.c_int), ("integral", C.c_int), ("location", C.c_int), ("loc2d", C.c_int * 3),
                ("coverage", C.c_int), ("n_tracts", C.c_int), ("mode_context_count", C.c_int), ("mode_context_length", C.c_int),
                ("mode_context_id", C.c_int), ("tmp_count", C.POINTER(C.c_int)), ("tmp_length", C.POINTER(C.c_int)), ("index", C.c_int),
                ("h", C.POINTER(EmpfreqStruct)), ("tract_id", C.c_int), ("ref_counter", C.c_int)]


class GenomicContextListStruct(C.Structure):
    _fields_ = [("hist", C.POINTER(C.POINTER(ContextHistogramStruct))), ("name", C.c_char_p), ("opt", Options),
                ("n_hist", C.c_int), ("coverage", C.c_int), ("ref_start", C.c_int)]


class Comm:
    """tjamd_comm: the process-per-GPU exchange of the C library (ncclAllGather over RCCL behind tjamd_allgather_histograms)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        """rank 0: the bytes every rank passes to Comm(...) (hand them over with whatever the job has: MPI, a file, a store)"""
        buf = C.create_string_buffer(Comm.ID_BYTES)
        if lib().tjamd_comm_unique_id(buf) != 0:
            raise TatajubaAmdError(_err())
        return buf.raw

    def __init__(self, counter, ident, rank, world):
        self.world = world
        self._id = C.create_string_buffer(bytes(ident), Comm.ID_BYTES)
        self._h = lib().tjamd_comm_create(counter._h, self._id, rank, world)
        if not self._h:
            raise TatajubaAmdError(_err())

    def set_stream(self, hip_stream):
        if lib().tjamd_comm_set_stream(self._h, C.c_void_p(hip_stream)) != 0:
            raise TatajubaAmdError(_err())

    def allgather(self, counter):
        """(device pointer of all ranks' kept records back to back, ctypes long array of records per rank, total)"""
        ptr, cnt = C.c_void_p(), (C.c_long * self.world)()
        tot = lib().tjamd_allgather_histograms(counter._h, self._h, C.byref(ptr), cnt)
        if tot < 0:
            raise TatajubaAmdError(_err())
        return ptr, cnt, tot

    @property
    def collectives(self):
        return lib().tjamd_comm_collectives(self._h)

    @property
    def count(self):
        """ranks RCCL reports for the communicator (ncclCommCount)"""
        lib().tjamd_comm_count.restype = C.c_int
        lib().tjamd_comm_count.argtypes = [C.c_void_p]
        return int(lib().tjamd_comm_count(self._h))

    def last_exchange(self):
        """(device ms, bytes delivered to this rank, collectives) of the last allgather; None before the first"""
        f = lib().tjamd_comm_last_exchange
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_long)]
        ms, nb, nc = C.c_double(), C.c_long(), C.c_long()
        if f(self._h, C.byref(ms), C.byref(nb), C.byref(nc)) != 0:
            return None
        return ms.value, nb.value, nc.value

    def close(self):
        if getattr(self, "_h", None):
            lib().tjamd_comm_destroy(self._h)
            self._h = None

    __del__ = close


class Counter:
    """tjamd_counter: device-side per-sample accumulator (include/tatajuba_amd.h)."""

    def __init__(self, kmer_size, device=0):
        self.k = kmer_size
        self._h = lib().tjamd_counter_create(device, kmer_size)
        if not self._h:
            raise TatajubaAmdError(_err())

    def close(self):
        if getattr(self, "_h", None):
            lib().tjamd_counter_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc):
        if rc != 0:
            raise TatajubaAmdError(_err())

    def _chkn(self, n):
        if n < 0:
            raise TatajubaAmdError(_err())
        return n

    def reset(self):
        self._chk(lib().tjamd_counter_reset(self._h))

    def set_stream(self, hip_stream):
        self._chk(lib().tjamd_counter_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_order_stream(self, hip_stream):
        """second stream for the ordering step of finalise_begin / finalise_end (tjamd_counter_set_order_stream); None: none"""
        lib().tjamd_counter_set_order_stream.argtypes = [C.c_void_p, C.c_void_p]
        self._chk(lib().tjamd_counter_set_order_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def sync(self):
        self._chk(lib().tjamd_sync(self._h))

    def scan_device(self, dptr, n_bytes, m):
        self._chk(lib().tjamd_scan_device(self._h, C.c_void_p(dptr), n_bytes, m))

    def scan_host(self, buf, m):
        a = np.ascontiguousarray(np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf)
        self._chk(lib().tjamd_scan_host(self._h, a.ctypes.data, a.size, m))
        self.sync()           # `a` may be a temporary

    def scan_host_located(self, buf, m):
        a = np.ascontiguousarray(np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf)
        cap = a.size // 2 + 2
        out = np.zeros(cap, dtype=LOCATED_DTYPE)
        n = self._chkn(lib().tjamd_scan_host_located(self._h, a.ctypes.data, a.size, m, out.ctypes.data, cap))
        return out[:n]

    def raw_count(self):
        return self._chkn(lib().tjamd_raw_count(self._h))

    def undefined_runs(self):
        return self._chkn(lib().tjamd_undefined_runs(self._h))

    def download_raw(self):
        n = self.raw_count()
        out = np.zeros(n, dtype=RECORD_DTYPE)
        self._chkn(lib().tjamd_download_raw(self._h, out.ctypes.data, n))
        return out

    def upload_raw(self, elems):
        e = np.ascontiguousarray(elems, dtype=ELEM_DTYPE)
        self._chk(lib().tjamd_upload_raw(self._h, e.ctypes.data, e.size))

    def finalise(self, remove_biased, min_coverage):
        st = C.c_int(-1)
        self._chk(lib().tjamd_finalise(self._h, int(bool(remove_biased)), int(min_coverage), C.byref(st)))
        return st.value

    def finalise_begin(self, remove_biased, min_coverage):
        """queue the device finalise and return; finalise_end() fetches the outcome (tjamd_finalise_begin / _end)"""
        self._chk(lib().tjamd_finalise_begin(self._h, int(bool(remove_biased)), int(min_coverage)))

    def finalise_end(self):
        st = C.c_int(-1)
        self._chk(lib().tjamd_finalise_end(self._h, C.byref(st)))
        return st.value

    @property
    def n_kept(self):
        return lib().tjamd_kept_count(self._h)

    @property
    def n_idx(self):
        return lib().tjamd_n_idx(self._h)

    @property
    def coverage(self):
        return lib().tjamd_coverage(self._h)

    @property
    def kept_device_ptr(self):
        return lib().tjamd_kept_device_ptr(self._h)

    def download_kept(self):
        n = self.n_kept
        out = np.zeros(n, dtype=ELEM_DTYPE)
        self._chkn(lib().tjamd_download_kept(self._h, out.ctypes.data, n))
        return out

    def download_idx(self):
        n = self.n_idx
        a, b = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self._chkn(lib().tjamd_download_idx(self._h, a.ctypes.data, b.ctypes.data, n))
        return a, b

    def group_contexts(self, max_distance_per_flank):
        """(group_of int32[n_kept], groups structured array) -- tjamd_group_contexts"""
        n = self.n_kept
        gof = np.zeros(max(n, 1), dtype=np.int32)
        grp = np.zeros(max(n, 1), dtype=GROUP_DTYPE)
        ng = self._chkn(lib().tjamd_group_contexts(self._h, max_distance_per_flank, gof.ctypes.data, grp.ctypes.data, n))
        return gof[:n], grp[:ng]

    def context_histograms(self, max_distance_per_flank, levenshtein_distance):
        """tjamd_context_histograms: dict of group_of, join_type (int32[n_kept]), groups (CONTEXT_GROUP_DTYPE), hist
        (LENGTH_FREQ_DTYPE[n_kept]: histogram g's entries at [first, first + n_len))"""
        n = self.n_kept
        gof, jt = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        grp = np.zeros(max(n, 1), dtype=CONTEXT_GROUP_DTYPE)
        hist = np.zeros(max(n, 1), dtype=LENGTH_FREQ_DTYPE)
        ng = self._chkn(lib().tjamd_context_histograms(self._h, max_distance_per_flank, levenshtein_distance, gof.ctypes.data, jt.ctypes.data,
                                                       grp.ctypes.data, hist.ctypes.data, n))
        return {"group_of": gof[:n], "join_type": jt[:n], "groups": grp[:ng], "hist": hist[:n]}

    def last_scan_launches(self):
        return int(lib().tjamd_last_scan_launches(self._h))

    def plan_mismatches(self):
        lib().tjamd_plan_mismatches.restype = C.c_long
        lib().tjamd_plan_mismatches.argtypes = [C.c_void_p]
        return int(lib().tjamd_plan_mismatches(self._h))

    def last_scan_ms(self):
        return lib().tjamd_last_scan_ms(self._h)

    def last_partition_ms(self):
        return lib().tjamd_last_partition_ms(self._h)

    def bucket_counts(self):
        """raw records per hash bucket (diagnostic: tjamd_debug_bucket_counts; synchronises)"""
        import numpy as np
        f = lib().tjamd_debug_bucket_counts
        f.restype = C.c_long
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        b = np.zeros(256, np.uint32)
        if f(self._h, b.ctypes.data, 256) < 0:
            raise TatajubaAmdError(_err())
        return b

    def last_merge_ms(self):
        lib().tjamd_last_merge_ms.restype = C.c_double
        lib().tjamd_last_merge_ms.argtypes = [C.c_void_p]
        return lib().tjamd_last_merge_ms(self._h)

    def uses_log(self):
        lib().tjamd_counter_uses_log.restype = C.c_int
        lib().tjamd_counter_uses_log.argtypes = [C.c_void_p]
        return bool(lib().tjamd_counter_uses_log(self._h))

    def last_finalise_ms(self):
        return lib().tjamd_last_finalise_ms(self._h)


class HopoCounter:
    """The reference's hopo_counter through the drop-in C functions (include/tatajuba_hopo.h)."""

    def __init__(self, ptr):
        self._p = ptr

    # reference: new_hopo_counter (src/hopo_counter.c:159)
    @classmethod
    def new(cls, kmer_size):
        return cls(lib().new_hopo_counter(kmer_size))

    # reference: new_or_append_hopo_counter_from_file (src/hopo_counter.c:135); hc=None creates
    @classmethod
    def new_or_append_from_file(cls, hc, filename, opt):
        p = lib().new_or_append_hopo_counter_from_file(hc._p if hc is not None else None, os.fsencode(filename), opt)
        if hc is not None:
            return hc
        return cls(p)

    # reference: update_hopo_counter_from_seq (src/hopo_counter.c:219)
    def update_from_seq(self, seq, min_tract_size):
        if isinstance(seq, str):
            seq = seq.encode("latin-1")
        lib().update_hopo_counter_from_seq(self._p, seq, len(seq), min_tract_size)

    # reference: update_hopo_counter_from_seq_all_monomers (src/hopo_counter.c:260)
    def update_from_seq_all_monomers(self, seq):
        if isinstance(seq, str):
            seq = seq.encode("latin-1")
        lib().update_hopo_counter_from_seq_all_monomers(self._p, seq, len(seq))

    # reference: finalise_hopo_counter (src/hopo_counter.c:339)
    def finalise(self):
        lib().finalise_hopo_counter(self._p)

    # reference: del_hopo_counter (src/hopo_counter.c:175)
    def delete(self):
        if self._p:
            lib().del_hopo_counter(self._p)
            self._p = None

    __del__ = delete

    @property
    def c(self):
        return self._p.contents

    def elems(self, n=None):
        n = self.c.n_elem if n is None else n
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
