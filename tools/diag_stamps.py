"""Diagnostic only: where a workgroup of the scan kernel spends its time, from in-kernel s_memtime stamps of thread 0.
Needs tatajuba_amd/libtatajuba_amd_diag.so (hopo_device.hip built with -DTJ_STAMPS=1; see tools/build_diag.sh).
The stamped build's run time is not a benchmark number; read the shares."""
import ctypes as C
import os
import sys

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
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
s = tj.synth_stream(n_reads, 150, int(os.environ.get("TJ_GENOME", "5000000")), n_threads=16)
d = torch.from_numpy(s).cuda()
c = tj.Counter(int(os.environ.get("TJ_K", "10")))
out = (C.c_ulonglong * 32)()
for it in range(3):
    c.reset()
    c.scan_device(d.data_ptr(), s.size, int(os.environ.get("TJ_M", "3")))
    c.sync()
    L.tjamd_debug_stamps(out, 1)
v = np.array(list(out), dtype=np.float64)
names = ["0 loop-top", "1 wait-dma", "2 phase1", "3 prefetch-issue", "4 phase2", "5 barrierA", "6 phase3-compute", "7 put (incl. partition)",
         "8 barrier-end", "9 part:entry-barrier", "10 part:hist+rank", "11 part:scan+reserve", "12 part:permute", "13 part:owners", "14 part:barrier",
         "15 part:copy-out"]
print("stamped scan ms", c.last_scan_ms())
tot = v[:9].sum()
for n, x in zip(names, v[:16]):
    print(f"{n:28s} {x / tot * 100:6.2f} %")
