import ctypes as C, time, random, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import tatajuba_amd as tj
rng = random.Random(3); k, nw = 8, 10000
wins = ["".join(rng.choice("ACGT") for _ in range(100)) for _ in range(nw)]
arr = (C.c_char_p * nw)(*[w.encode() for w in wins]); lens = (C.c_int * nw)(*[len(w) for w in wins])
L = tj.lib(); cap = 1100000
out = np.zeros(cap, dtype=tj.ELEM_DTYPE); wof = np.zeros(cap, dtype=np.int32)
L.tjamd_scan_windows(k, arr, lens, nw, 2, out.ctypes.data, wof.ctypes.data, cap)
t = time.perf_counter(); n = L.tjamd_scan_windows(k, arr, lens, nw, 2, out.ctypes.data, wof.ctypes.data, cap); dt = time.perf_counter() - t
print("scan_windows: %d windows of 100 bases, %d records, %.2f ms" % (nw, n, dt * 1e3))
h = tj.HopoCounter.new(k)
t = time.perf_counter()
for w in wins[:1000]:
    hh = tj.HopoCounter.new(k); hh.update_from_seq(w, 2); hh.delete()
print("1000 x (new_hopo_counter + update_hopo_counter_from_seq + del): %.2f ms" % ((time.perf_counter() - t) * 1e3))
