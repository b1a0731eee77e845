import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, tatajuba_amd as tj
s = tj.synth_stream(100_000_000, 150, 50_000_000, n_threads=16)
d = torch.from_numpy(s).cuda()
free0, total = torch.cuda.mem_get_info()
c = tj.Counter(15)
for it in range(2):
    c.reset(); c.scan_device(d.data_ptr(), s.size, 4); st = c.finalise(1, 5)
free1, _ = torch.cuda.mem_get_info()
print("stream %.1f GB; library buffers %.1f GB; launches %d; scan %.2f ms fin %.2f ms kept %d" % (s.size/1e9, (free0-free1)/1e9, c.last_scan_launches(), c.last_scan_ms(), c.last_finalise_ms(), c.n_kept))
