"""Diagnostic only: per-section cycle shares of aggregate1_kernel (waves' s_memtime stamps; -DTJ_STAMPS=1 build)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tatajuba_amd.build as B
B._SO = os.path.join(ROOT, "tatajuba_amd", os.environ.get("TJ_DIAG_LIB", "libtatajuba_amd_diag.so"))
import tatajuba_amd.capi as capi
capi.library_path = lambda: B._SO
import tatajuba_amd as tj
import torch
L = tj.lib()
s = tj.synth_stream(10_000_000, 150, 5_000_000, n_threads=16)
d = torch.from_numpy(s).cuda()
c = tj.Counter(int(os.environ.get("TJ_K", "10")))
out = (C.c_ulonglong * 32)()
for it in range(3):
    c.reset(); c.scan_device(d.data_ptr(), s.size, 3); c.sync()
    L.tjamd_debug_stamps(out, 1)
    c.finalise(True, 5)
    L.tjamd_debug_stamps(out, 1)
v = np.array(list(out), dtype=np.float64)[16:24]
names = ["(loop top)", "clear+first fetch", "wait loads", "issue next fetch", "barrier", "insert", "loop-end sync", "emit"]
print("finalise ms", c.last_finalise_ms(), " waves*cycles total %.3g  (100 MHz s_memtime ticks)" % v.sum())
for n, x in zip(names, v):
    print(f"{n:20s} {x / v.sum() * 100:6.2f} %   {x / (256 * 16):10.0f} ticks per wave")
