import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tatajuba_amd as tj, torch
s = tj.synth_stream(30_000_000, 150, 20_000_000, n_threads=16)
d = torch.from_numpy(s).cuda()
c = tj.Counter(10)
for it in range(2):
    c.reset(); c.scan_device(d.data_ptr(), s.size, 3); c.sync()
    print("launches", c.last_scan_launches(), "scan ms %.3f" % c.last_scan_ms(), "partition ms %.3f" % c.last_partition_ms(), "raw", c.raw_count(), "GB", s.size / 1e9,
          "frac %.3f" % (s.size / c.last_scan_ms() / 1e6 / 8000))
c.close()
