"""Builds libtatajuba_amd.so in-tree (gcc for the C host, hipcc --offload-arch=gfx950 for the kernels)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.path.join(_HERE, os.environ.get("TJ_DIAG_LIB", "libtatajuba_amd.so"))     # (TJ_DIAG_LIB: an experimental build, tools/ only)
_SOURCES = ["hopo_device.hip", "hopo_host.c", "fastq_reader.c", "fastq_reader.h", "feeder.c", "feeder.h", "tj_inflate.c", "tj_inflate.h", "synth.c", "exports.map", "Makefile",
            os.path.join("..", "..", "include", "tatajuba_amd.h"), os.path.join("..", "..", "include", "tatajuba_hopo.h")]


def library_path():
    return _SO


def _stale():
    if not os.path.exists(_SO):
        return True
    t = os.path.getmtime(_SO)
    return any(os.path.getmtime(os.path.join(_CSRC, s)) > t for s in _SOURCES)


def build_library(force=False, verbose=False):
    """Compile if sources are newer than the .so (or force).  Needs hipcc; cross-compiles without a GPU."""
    if "TJ_DIAG_LIB" in os.environ:
        return _SO
    if force or _stale():
        cmd = ["make", "-C", _CSRC] + (["-B"] if force else [])
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if verbose or out.returncode:
            print(out.stdout)
        if out.returncode:
            raise RuntimeError("building libtatajuba_amd.so failed")
    return _SO
