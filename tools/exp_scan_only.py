"""Diagnostic: run only the scan of a library build (TJ_DIAG_LIB) a few times; prints the HIP-event time per scan."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tatajuba_amd.build as B
B._SO = os.path.join(ROOT, "tatajuba_amd", os.environ.get("TJ_DIAG_LIB", "libtatajuba_amd.so"))
import tatajuba_amd.capi as capi
capi.library_path = lambda: B._SO
import tatajuba_amd as tj
import torch
s = tj.synth_stream(int(os.environ.get("TJ_READS", "10000000")), 150, int(os.environ.get("TJ_GENOME", "5000000")), n_threads=16)
d = torch.from_numpy(s).cuda()
c = tj.Counter(int(os.environ.get("TJ_K", "10")))
ms = []
for it in range(int(os.environ.get("TJ_REPS", "6"))):
    try:
        c.reset(); c.scan_device(d.data_ptr(), s.size, int(os.environ.get("TJ_M", "3"))); c.sync()
    except tj.TatajubaAmdError as e:           # (ablated builds may trip the capacity checks: the time still counts)
        print("error:", str(e)[:60])
    ms.append(c.last_scan_ms())
    pm = c.last_partition_ms() if hasattr(c, "last_partition_ms") else 0.0
print(os.environ.get("TJ_DIAG_LIB"), "fast", os.environ.get("TATAJUBA_AMD_FAST", "1"), "scan ms", " ".join("%.3f" % x for x in ms), "raw", (c.raw_count() if not os.environ.get("TJ_NORAW") else -1),
      "GB/s %.0f" % (s.size / min(ms[1:]) / 1e6), "partition ms %.3f" % pm)
c.close()      # (before the interpreter starts taking modules apart: the counter owns a stream and device buffers)
