"""Builds libtatajuba_amd.so in-tree (gcc for the C host, hipcc --offload-arch=gfx950 for the kernels).

The library carries a hash of the sources it was built from (tjamd_source_hash (), computed by the Makefile); the same
hash is computed here over the tree, so "is this .so the tree's?" is a comparison of two strings, not of time stamps:
a prebuilt library that travelled with the tree is reused only if it matches, and a mismatch after a build is an error."""
import ctypes
import glob
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_INC = os.path.join(_HERE, "..", "include")
_SO = os.path.join(_HERE, os.environ.get("TJ_DIAG_LIB", "libtatajuba_amd.so"))     # (TJ_DIAG_LIB: an experimental build, tools/ only)

last_action = None      # "compiled" / "reused", set by build_library


def library_path():
    return _SO


def source_hash():
    """the Makefile's SRC_HASH: sha256 over (basename, NUL, contents) of csrc/*.{c,h,hip}, exports.map, Makefile, include/*.h"""
    files = sorted(os.path.basename(f) for pat in ("*.c", "*.h", "*.hip") for f in glob.glob(os.path.join(_CSRC, pat)))
    paths = [os.path.join(_CSRC, f) for f in files] + [os.path.join(_CSRC, "exports.map"), os.path.join(_CSRC, "Makefile")]
    paths += [os.path.join(_INC, f) for f in sorted(os.path.basename(f) for f in glob.glob(os.path.join(_INC, "*.h")))]
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def library_hash(path=None):
    """the hash inside a built library (None if there is none or it predates the hash); read from the file, not by loading it"""
    path = path or _SO
    try:
        data = open(path, "rb").read()
    except OSError:
        return None
    tag = b"tatajuba_amd 0.3 (gfx950) src "
    i = data.find(tag)
    return data[i + len(tag): i + len(tag) + 16].decode("ascii", "replace") if i >= 0 else None


def build_library(force=False, verbose=False):
    """Compile unless the library in the tree was built from exactly these sources (or force).  Needs hipcc; cross-compiles
    without a GPU.  Raises if what comes out does not carry the tree's hash."""
    global last_action
    if "TJ_DIAG_LIB" in os.environ:
        last_action = "reused"
        return _SO
    want = source_hash()
    if not force and library_hash() == want:
        last_action = "reused"
        return _SO
    cmd = ["make", "-C", _CSRC] + (["-B"] if force else [])
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode:
        print(out.stdout)
    if out.returncode:
        raise RuntimeError("building libtatajuba_amd.so failed")
    got = library_hash()
    if got != want:
        raise RuntimeError(f"libtatajuba_amd.so carries source hash {got}, the tree's is {want}")
    last_action = "compiled"
    return _SO


def check_loaded(lib):
    """the loaded library must be the tree's (a stale prebuilt .so on the library path would pass every symbol check)"""
    lib.tjamd_source_hash.restype = ctypes.c_char_p
    got, want = lib.tjamd_source_hash().decode(), source_hash()
    if "TJ_DIAG_LIB" not in os.environ and got != want:
        raise RuntimeError(f"the loaded libtatajuba_amd.so was built from sources {got}, the tree's hash is {want}: rebuild (make -C tatajuba_amd/csrc)")
    return got
