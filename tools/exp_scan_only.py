"""Diagnostic: run only the scan kernel of an experimental library build (TJ_DIAG_LIB) a few times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tatajuba_amd.build as B
B._SO = os.path.join(ROOT, "tatajuba_amd", os.environ.get("TJ_DIAG_LIB", "libtatajuba_amd.so"))
import tatajuba_amd.capi as capi
capi.library_path = lambda: B._SO
import tatajuba_amd as tj
import torch
s = tj.synth_stream(10_000_000, 150, 5_000_000, n_threads=16)
d = torch.from_numpy(s).cuda()
c = tj.Counter(int(os.environ.get("TJ_K", "10")))
for it in range(3):
    c.reset(); c.scan_device(d.data_ptr(), s.size, int(os.environ.get("TJ_M", "3"))); c.sync()
print(os.environ.get("TJ_DIAG_LIB"), "scan ms", c.last_scan_ms())
