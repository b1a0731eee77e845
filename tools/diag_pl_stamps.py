"""Diagnostic only: where a workgroup of partition_log_kernel spends its time (s_memtime stamps of thread 0).
Needs tatajuba_amd/libtatajuba_amd_pldiag.so (hopo_device.hip built with -DTJ_STAMPS=2: tools/build_exp_many.sh "pldiag:-DTJ_STAMPS=2")."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tatajuba_amd.build as B
B._SO = os.path.join(ROOT, "tatajuba_amd", os.environ.get("TJ_DIAG_LIB", "libtatajuba_amd_pldiag.so"))
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
v = np.array(list(out), dtype=np.float64)
names = ["0 loop/count", "1 sweep 1: loads + counts", "2 barrier", "3 owners: prefix, reservation, chunk", "4 barrier", "5 sweep 2: loads + LDS sort", "6 wait + barrier",
         "7 copy-out", "8 barrier"]
print("partition ms", c.last_partition_ms(), "ticks total", v[:16].sum())
for n, x in zip(names, v[:9]):
    print(f"{n:40s} {x / v[:16].sum() * 100:6.2f} %")
c.close()
