"""Experiment: scan time when a fraction of the reads holds an N (tiles with an N go to the general kernel).
usage: exp_n_reads.py [fraction ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tatajuba_amd as tj
s0 = tj.synth_stream(10_000_000, 150, 5_000_000, n_threads=16)
rng = np.random.default_rng(3)
fracs = [float(x) for x in sys.argv[1:]] or [0.0, 0.001, 0.005, 0.02, 0.1]
for frac in fracs:
    s = s0.copy()
    n_bad = int(10_000_000 * frac)
    if n_bad:
        reads = rng.choice(10_000_000, n_bad, replace=False)
        s[reads * 151 + rng.integers(0, 150, n_bad)] = ord("N")
    d = torch.from_numpy(s).cuda()
    c = tj.Counter(10)
    best = 1e9
    for it in range(5):
        c.reset(); c.scan_device(d.data_ptr(), s.size, 3); c.sync()
        best = min(best, c.last_scan_ms())
    L = tj.lib(); L.tjamd_debug_slow_tiles.restype = __import__("ctypes").c_long; L.tjamd_debug_slow_tiles.argtypes = [__import__("ctypes").c_void_p]
    print(f"reads with an N: {frac * 100:5.2f} %   scan {best:.3f} ms   raw {c.raw_count()}   tiles left to the generic kernel: {L.tjamd_debug_slow_tiles(c._h)} of {(s.size + 16287) // 16288}")
    c.close()
