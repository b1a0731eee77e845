"""End-to-end rates around the hot path, for DESIGN.md (never bench.py's `value`): host-resident stream through
tjamd_scan_host (PCIe-inclusive), and FASTQ files (plain / gzip) through new_or_append_hopo_counter_from_file.
Run on the GPU box:  python tools/e2e_rates.py > gpurun_out/e2e.json"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tatajuba_amd as tj  # noqa: E402
from tatajuba_amd import capi  # noqa: E402

out = {}
n_reads, L = 10_000_000, 150
s = tj.synth_stream(n_reads, L, 5_000_000)
L_ = capi.lib()
pin = L_.tjamd_host_alloc(s.size)
C.memmove(pin, s.ctypes.data, s.size)
c = tj.Counter(10)
B = 64 << 20
for rep in range(3):
    c.reset()
    t = time.perf_counter()
    for o in range(0, s.size, B):            # batches cut anywhere are fine only on read boundaries: cut at '\n'
        e = min(o + B, s.size)
        L_.tjamd_scan_host(c._h, C.c_void_p(pin + o), e - o, 3)
    c.sync()
    dt = time.perf_counter() - t
# (cuts inside reads lose the tracts that straddle them; rate measurement only)
out["pinned_host_stream"] = {"reads_per_s": n_reads / dt, "GBps": s.size / dt / 1e9, "batch_MiB": 64}
t = time.perf_counter(); c.finalise(True, 5); out["finalise_s"] = time.perf_counter() - t
c.close()
L_.tjamd_host_free(C.c_void_p(pin))

tmp = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
nf = 2_000_000
reads = bytes(s[: nf * (L + 1)]).split(b"\n")[:-1]
q = b"I" * L
fq = os.path.join(tmp, "a.fq")
with open(fq, "wb") as f:
    for i in range(0, nf, 100000):
        f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (j, reads[j], q) for j in range(i, min(nf, i + 100000))))
subprocess.check_call(["gzip", "-1", "-k", "-f", fq])
opt = tj.Options.defaults(10, 3, 5, True)
for name, path in (("fastq_plain", fq), ("fastq_gz", fq + ".gz")):
    best = 1e9
    for rep in range(2):
        t = time.perf_counter()
        h = tj.HopoCounter.new_or_append_from_file(None, path, opt)
        h.finalise()
        best = min(best, time.perf_counter() - t)
        n = h.c.n_elem
        h.delete()
    out[name] = {"reads_per_s": nf / best, "file_MB": os.path.getsize(path) / 1e6, "kept": n, "seconds": best}
print(json.dumps(out))
