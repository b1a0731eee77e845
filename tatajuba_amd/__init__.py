"""tatajuba_amd -- MI355X-native homopolymer-tract counting engine (tatajuba's hopo_counter hot path).

The product is libtatajuba_amd.so: a C-ABI shared library (include/tatajuba_hopo.h, include/tatajuba_amd.h) whose host
side is C and whose compute is hand-written HIP for gfx950.  This Python package only loads it through ctypes for the
tests and the benchmark; it holds no compute of its own and never touches oracle/.
"""
from .build import build_library, library_path  # noqa: F401
from .capi import (Counter, Comm, HopoCounter, Options, lib, device_count, synth_stream, ELEM_DTYPE, RECORD_DTYPE,  # noqa: F401
                   LOCATED_DTYPE, GROUP_DTYPE, decode_meta, TatajubaAmdError, read_file_stream, EXPORTS)
