"""Diagnostic: where the drop-in stage's host time goes once the device finalise is done (bench.py's stages.dropin):
tjamd_download_kept into a fresh array / into a warm one, tjamd_download_idx likewise."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tatajuba_amd as tj
import torch
L = tj.lib()
s = tj.synth_stream(10_000_000, 150, 5_000_000, n_threads=16)
d = torch.from_numpy(s).cuda()
c = tj.Counter(10)
def once(fresh):
    c.reset(); c.scan_device(d.data_ptr(), s.size, 3)
    assert c.finalise(1, 5) == 0
    n, ni = c.n_kept, c.n_idx
    t0 = time.perf_counter()
    out = np.zeros(n, dtype=tj.ELEM_DTYPE) if fresh else once.out
    t1 = time.perf_counter()
    assert L.tjamd_download_kept(c._h, out.ctypes.data, n) == n
    t2 = time.perf_counter()
    a, b = (np.zeros(ni, np.int32), np.zeros(ni, np.int32)) if fresh else once.idx
    t3 = time.perf_counter()
    assert L.tjamd_download_idx(c._h, a.ctypes.data, b.ctypes.data, ni) == ni
    t4 = time.perf_counter()
    once.out, once.idx = out, (a, b)
    return n, ni, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3
once(True)
for fresh in (True, True, False, False, True, False):
    n, ni, ta, tk, tb, ti = once(fresh)
    print("fresh" if fresh else "warm ", "kept %d idx %d: alloc %.3f ms, download_kept %.3f ms, alloc idx %.3f, download_idx %.3f ms" % (n, ni, ta, tk, tb, ti))
c.close()
