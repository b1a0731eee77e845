"""Diagnostic: how evenly does the bucket hash spread the raw records?  usage: bucket_balance.py k m reads read_len read_len_max genome"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tatajuba_amd as tj
k, m, reads, rl, rlmax, genome = [int(x) for x in sys.argv[1:7]]
s = tj.synth_stream(reads, rl, genome, read_len_max=rlmax, n_threads=16)
c = tj.Counter(k)
c.scan_host(s, m)
out = (C.c_uint * 256)()
tj.lib().tjamd_debug_bucket_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
tj.lib().tjamd_debug_bucket_counts(c._h, out, 256)
v = np.array(list(out), dtype=np.float64)
print(f"k={k} records {int(v.sum())} buckets: min {int(v.min())} median {int(np.median(v))} max {int(v.max())} max/mean {v.max() / v.mean():.2f} empty {(v == 0).sum()}")
