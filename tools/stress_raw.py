"""Diagnostic: scan one resident 10 M-read stream TJ_RUNS times (TJ_K, TJ_M: parameters) and compare the raw record count
with the first run's -- a lost append or a dropped tile shows up as a different count; a deviating run is printed as
(run, records, buckets that differ, smallest and largest per-bucket difference).  TJ_DIAG_LIB picks the build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tatajuba_amd.build as B
B._SO = os.path.join(ROOT, "tatajuba_amd", os.environ.get("TJ_DIAG_LIB", "libtatajuba_amd.so"))
import tatajuba_amd.capi as capi
capi.library_path = lambda: B._SO
import numpy as np
import tatajuba_amd as tj
import torch
n = int(os.environ.get("TJ_RUNS", "60"))
s = tj.synth_stream(10_000_000, 150, 5_000_000, n_threads=16)
d = torch.from_numpy(s).cuda()
K, M = int(os.environ.get("TJ_K", "10")), int(os.environ.get("TJ_M", "3"))
c = tj.Counter(K)
want, bad = None, []
import ctypes as C
L = tj.lib()
L.tjamd_debug_bucket_counts.restype = C.c_long
L.tjamd_debug_bucket_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ref_b = None
for it in range(n):
    c.reset(); c.scan_device(d.data_ptr(), s.size, M)
    r = c.raw_count()
    b = np.zeros(256, np.uint32)
    L.tjamd_debug_bucket_counts(c._h, b.ctypes.data, 256)
    if want is None: want, ref_b = r, b.copy()
    elif r != want:
        diff = b.astype(np.int64) - ref_b.astype(np.int64)
        bad.append((it, r - want, int((diff != 0).sum()), int(diff.min()), int(diff.max())))
print(os.environ.get("TJ_DIAG_LIB", "libtatajuba_amd.so"), "k", K, "m", M, "runs", n, "reference", want, "deviations", bad)
c.close()
