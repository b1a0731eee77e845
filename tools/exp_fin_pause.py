"""Diagnostic: does the finalise time depend on what the GPU did just before it (clock management)?  The same sample is
scanned, then the host waits `pause` ms before it queues the finalise; and the scan is run 1-3 times back to back first."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tatajuba_amd as tj
import torch
s = tj.synth_stream(10_000_000, 150, 5_000_000, n_threads=16)
d = torch.from_numpy(s).cuda()
c = tj.Counter(10)
c2 = tj.Counter(10)
for heat in (0, 1, 3):
    for pause in (0.0, 0.002, 0.02):
        fin, scan = [], []
        for it in range(8):
            for _ in range(heat):                       # extra scans on another counter right before: a hotter chip
                c2.reset(); c2.scan_device(d.data_ptr(), s.size, 3)
            if heat: c2.sync()
            c.reset(); c.scan_device(d.data_ptr(), s.size, 3); c.sync()
            if pause: time.sleep(pause)
            assert c.finalise(1, 5) == 0
            fin.append(c.last_finalise_ms()); scan.append(c.last_scan_ms())
        print("extra scans %d pause %4.0f ms: scan %.3f finalise %.3f (min %.3f)" % (heat, pause * 1e3, np.median(scan), np.median(fin), min(fin)))
c.close(); c2.close()
