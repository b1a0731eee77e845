"""Diagnostic: does the cross-sample merge (second stream, helper thread) overlap the next sample's scan on one GPU?"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tatajuba_amd as tj
from tatajuba_amd.dist import device_bytes_tensor, merge_histograms_device
N = 8
dev = torch.device("cuda", 0)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
s = tj.synth_stream(10_000_000, 150, 5_000_000, n_threads=16)
d = torch.from_numpy(s).cuda()
c = tj.Counter(10); c.set_stream(main.cuda_stream)
merger = tj.Counter(10); merger.set_stream(side.cuda_stream)
c.reset(); c.scan_device(d.data_ptr(), s.size, 3); c.finalise(1, 5)
n = c.n_kept
one = device_bytes_tensor(c.kept_device_ptr, n * 24, dev).clone()
rec = torch.cat([one] * N); cnts = [n] * N

def merge():
    torch.cuda.set_device(0)
    with torch.cuda.stream(side):
        merge_histograms_device(merger, rec, cnts)

def run(mode, steps=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        c.reset(); c.scan_device(d.data_ptr(), s.size, 3)
        th = None
        if mode == "overlap":
            th = threading.Thread(target=merge); th.start()
        c.finalise(1, 5)
        if mode == "serial":
            merge()
        if th: th.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3
for mode in ("none", "serial", "overlap", "none", "serial", "overlap"):
    print(mode, "%.3f ms/step" % run(mode))
