"""Diagnostic: scan + finalise of the bench sample with an experimental library build (TJ_DIAG_LIB); prints stage times."""
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
for it in range(4):
    c.reset(); c.scan_device(d.data_ptr(), s.size, int(os.environ.get("TJ_M", "3")))
    try:
        c.finalise(True, 5)
    except Exception as e:
        pass
print(os.environ.get("TJ_DIAG_LIB"), "scan ms %.3f finalise ms %.3f" % (c.last_scan_ms(), c.last_finalise_ms()))
